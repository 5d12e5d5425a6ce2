// aot/tasp.hip -- ahead-of-time instances (ldpc_aot.hpp), one translation unit of the parallel build
#include "../ldpc_aot.hpp"

// two lanes per check: 2 M threads per frame, two waves per SIMD

LDPC_AOT_KERNEL(tasp_spec_appendix_c_m64_kernel, tasp_body, CodeAppendixCM64, 128, 2)
LDPC_AOT_KERNEL(tasp_spec_appendix_c_m126_kernel, tasp_body, CodeAppendixCM126, 256, 2)
