// aot/ims.hip -- ahead-of-time instances (ldpc_aot.hpp), one translation unit of the parallel build
#include "../ldpc_aot.hpp"

LDPC_AOT_KERNEL(ims_spec_appendix_c_m64_kernel, ims_body, CodeAppendixCM64, 64, 3)
LDPC_AOT_KERNEL(ims_spec_appendix_c_m126_kernel, ims_body, CodeAppendixCM126, 128, 2)
