// ldpc_aot.hpp -- the ahead-of-time instances of the code-specialised bodies (ldpc_spec.hpp) for the shipped example code
//  at the liftings the BASELINE configurations use.  Each instance is DEFINED in one of the small translation
// units under aot/ (so that the library builds in parallel and a change to the host code does not recompile 13 unrolled
// kernels) and DECLARED here for ldpc_hip.hip's launch table.  Everything else is compiled at ldpc_hip_open() with hiprtc from the
// same header (ldpc_jit.hpp).
#pragma once

#include <hip/hip_runtime.h>

#include "ldpc_spec.hpp"
#include "code_appendix_c_m64.hpp"

#define LDPC_AOT_DECLARE(name, threads, waves_per_simd) \
    __global__ void __launch_bounds__(threads, waves_per_simd) name(const ldpc_spec::SpecArgs a);
#define LDPC_AOT_KERNEL(name, body, Code, threads, waves_per_simd)                                   \
    __global__ void __launch_bounds__(threads, waves_per_simd) name(const ldpc_spec::SpecArgs a) {   \
        ldpc_spec::body<ldpc_spec::Code>(a);                                                         \
    }

LDPC_AOT_DECLARE(ms_spec_appendix_c_m64_kernel, 64, 2)
LDPC_AOT_DECLARE(ms_spec_appendix_c_m126_kernel, 128, 2)
LDPC_AOT_DECLARE(ms_chunk_appendix_c_m126_kernel, 64, 1)
LDPC_AOT_DECLARE(ms_spec_appendix_c_m512_kernel, 512, 2)
LDPC_AOT_DECLARE(ims_spec_appendix_c_m64_kernel, 64, 3)
LDPC_AOT_DECLARE(ims_spec_appendix_c_m126_kernel, 128, 2)
LDPC_AOT_DECLARE(lms_spec_appendix_c_m64_kernel, 64, 2)
LDPC_AOT_DECLARE(lms_spec_appendix_c_m512_kernel, 512, 2)
LDPC_AOT_DECLARE(sp_spec_appendix_c_m64_kernel, 512, 4)
LDPC_AOT_DECLARE(bp_spec_appendix_c_m64_kernel, 512, 4)
LDPC_AOT_DECLARE(asp_spec_appendix_c_m64_kernel, 512, 4)
LDPC_AOT_DECLARE(tasp_spec_appendix_c_m64_kernel, 128, 2)
LDPC_AOT_DECLARE(tasp_spec_appendix_c_m126_kernel, 256, 2)
