// aot/ms_m64.hip -- ahead-of-time instances (ldpc_aot.hpp), one translation unit of the parallel build
#include "../ldpc_aot.hpp"

LDPC_AOT_KERNEL(ms_spec_appendix_c_m64_kernel, ms_m64_body, CodeAppendixCM64, 64, 2)
