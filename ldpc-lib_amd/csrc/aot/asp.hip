// aot/asp.hip -- ahead-of-time instances (ldpc_aot.hpp), one translation unit of the parallel build
#include "../ldpc_aot.hpp"

LDPC_AOT_KERNEL(asp_spec_appendix_c_m64_kernel, asp_body, CodeAppendixCM64, 512, 4)
