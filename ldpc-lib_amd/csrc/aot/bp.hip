// aot/bp.hip -- ahead-of-time instances (ldpc_aot.hpp), one translation unit of the parallel build
#include "../ldpc_aot.hpp"

// asp / bp: two frames per CU (<= 128 VGPRs, some spills) beats one frame with 243 VGPRs
LDPC_AOT_KERNEL(bp_spec_appendix_c_m64_kernel, bp_body, CodeAppendixCM64, 512, 4)
