// aot/lms.hip -- ahead-of-time instances (ldpc_aot.hpp), one translation unit of the parallel build
#include "../ldpc_aot.hpp"

LDPC_AOT_KERNEL(lms_spec_appendix_c_m64_kernel, lms_body, CodeAppendixCM64, 64, 2)
LDPC_AOT_KERNEL(lms_spec_appendix_c_m512_kernel, lms_body, CodeAppendixCM512, 512, 2)
