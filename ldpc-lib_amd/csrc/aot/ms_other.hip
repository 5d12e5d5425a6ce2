// aot/ms_other.hip -- ahead-of-time instances (ldpc_aot.hpp), one translation unit of the parallel build
#include "../ldpc_aot.hpp"

LDPC_AOT_KERNEL(ms_spec_appendix_c_m126_kernel, ms_body, CodeAppendixCM126, 128, 2)
LDPC_AOT_KERNEL(ms_chunk_appendix_c_m126_kernel, ms_chunk_body, CodeAppendixCM126, 64, 1)
LDPC_AOT_KERNEL(ms_spec_appendix_c_m512_kernel, ms_body, CodeAppendixCM512, 512, 2)
