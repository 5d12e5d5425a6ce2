// aot/sp.hip -- ahead-of-time instances (ldpc_aot.hpp), one translation unit of the parallel build
#include "../ldpc_aot.hpp"

// eight waves per frame, two frames per CU (ldpc_spec::kSpBodyWaves)
LDPC_AOT_KERNEL(sp_spec_appendix_c_m64_kernel, sp_body, CodeAppendixCM64, 512, 4)
