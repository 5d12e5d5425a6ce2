// ubench_mtgen.hip -- where does the time of the exact-replay generator's word kernel go?  Variants of the block-parallel recurrence
// (csrc/ldpc_mt.hpp: mt_generate_kernel) on 653 streams of 2^20 words (a 2^27-sample round): with / without the stores to HBM, with
// the streams' memory staggered, different workgroup sizes.  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mtgen.hip -o /tmp/ubench_mtgen
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../ldpc-lib_amd/csrc/ldpc_spec.hpp"
#include "../ldpc-lib_amd/csrc/ldpc_mt.hpp"

template <int T>
__global__ void __launch_bounds__(T) lib_gen(const ldpc_mt::GenArgs a) { ldpc_mt::mt_generate_body<T>(a); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using ldpc_mt::MTN;
using ldpc_mt::mt_twist;

__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// T threads per stream; STORE: write the words; pitch: words between the starts of consecutive streams in memory
template <int T, bool STORE>
__global__ void __launch_bounds__(T) gen(const uint32_t *states, uint32_t *xraw, uint32_t nwords, size_t pitch, uint32_t *sink) {
    __shared__ uint32_t xb[2][MTN + 16];
    __shared__ uint32_t tv[MTN + 16];
    const int tid = threadIdx.x, j = blockIdx.x;
    for (int i = tid; i < MTN; i += T) xb[0][i] = states[(size_t)j * MTN + i];
    __syncthreads();
    uint32_t *out = xraw + (size_t)j * pitch + MTN;
    int cur = 0;
    uint32_t keep = 0;
    for (uint32_t w0 = 0; w0 < nwords; w0 += MTN) {
        const uint32_t *x = xb[cur];
        uint32_t *nx = xb[cur ^ 1];
        for (int i = tid; i < MTN - 1; i += T) tv[i] = mt_twist(x[i], x[i + 1]);
        lds_barrier();
        const uint32_t remaining = nwords - w0;
        for (int i = tid; i < MTN; i += T) {
            uint32_t v;
            if (i < 227) v = x[i + 397] ^ tv[i];
            else if (i < 454) v = x[i + 170] ^ tv[i - 227] ^ tv[i];
            else if (i < MTN - 1) v = x[i - 57] ^ tv[i - 454] ^ tv[i - 227] ^ tv[i];
            else v = x[i - 57] ^ tv[i - 454] ^ tv[i - 227] ^ mt_twist(x[MTN - 1], x[397] ^ tv[0]);
            nx[i] = v;
            if (STORE) { if ((uint32_t)i < remaining) out[w0 + i] = v; }
            else keep ^= v;
        }
        lds_barrier();
        cur ^= 1;
    }
    if (!STORE && keep == 0x12345u) sink[0] = keep;
}

// plain streaming store of the same volume, for the write bandwidth of the part
__global__ void __launch_bounds__(256) fill(uint32_t *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (uint32_t)i;
}

template <class F>
float timeit(F f, int reps = 3) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const int S = 653;
    const uint32_t nwords = 1u << 20;
    const size_t pad = 2368;
    std::vector<uint32_t> h((size_t)S * MTN);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)rand() * 2654435761u + (uint32_t)i;
    uint32_t *st, *x, *sink;
    CK(hipMalloc(&st, h.size() * 4));
    CK(hipMemcpy(st, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const size_t words = (size_t)MTN + (size_t)S * (nwords + pad) + 64;
    CK(hipMalloc(&x, words * 4));
    CK(hipMalloc(&sink, 64));
    const double gb = (double)S * nwords * 4 / 1e9;
    float t;
    t = timeit([&] { hipLaunchKernelGGL(fill, dim3(256 * 16), dim3(256), 0, 0, x, (size_t)S * nwords); });
    printf("streaming store of %.2f GB            : %8.3f ms  %6.2f TB/s\n", gb, t, gb / t);
    t = timeit([&] { hipLaunchKernelGGL((gen<320, true>), dim3(S), dim3(320), 0, 0, st, x, nwords, (size_t)nwords, sink); });
    printf("gen<320> store, pitch 2^20             : %8.3f ms  %6.2f TB/s\n", t, gb / t);
    t = timeit([&] { hipLaunchKernelGGL((gen<320, true>), dim3(S), dim3(320), 0, 0, st, x, nwords, (size_t)nwords + pad, sink); });
    printf("gen<320> store, pitch 2^20 + %zu      : %8.3f ms  %6.2f TB/s\n", pad, t, gb / t);
    t = timeit([&] { hipLaunchKernelGGL((gen<320, false>), dim3(S), dim3(320), 0, 0, st, x, nwords, (size_t)nwords, sink); });
    printf("gen<320> no store                      : %8.3f ms\n", t);
    t = timeit([&] { hipLaunchKernelGGL((gen<640, true>), dim3(S), dim3(640), 0, 0, st, x, nwords, (size_t)nwords, sink); });
    printf("gen<640> store, pitch 2^20             : %8.3f ms  %6.2f TB/s\n", t, gb / t);
    t = timeit([&] { hipLaunchKernelGGL((gen<640, false>), dim3(S), dim3(640), 0, 0, st, x, nwords, (size_t)nwords, sink); });
    printf("gen<640> no store                      : %8.3f ms\n", t);
    t = timeit([&] { hipLaunchKernelGGL((gen<256, true>), dim3(S), dim3(256), 0, 0, st, x, nwords, (size_t)nwords, sink); });
    printf("gen<256> store, pitch 2^20             : %8.3f ms  %6.2f TB/s\n", t, gb / t);
    t = timeit([&] { hipLaunchKernelGGL((gen<256, false>), dim3(S), dim3(256), 0, 0, st, x, nwords, (size_t)nwords, sink); });
    printf("gen<256> no store                      : %8.3f ms\n", t);
    t = timeit([&] { hipLaunchKernelGGL((gen<128, false>), dim3(S), dim3(128), 0, 0, st, x, nwords, (size_t)nwords, sink); });
    printf("gen<128> no store                      : %8.3f ms\n", t);
    t = timeit([&] { hipLaunchKernelGGL((gen<128, true>), dim3(S), dim3(128), 0, 0, st, x, nwords, (size_t)nwords + pad, sink); });
    printf("gen<128> store, pitch 2^20 + %zu      : %8.3f ms  %6.2f TB/s\n", pad, t, gb / t);
    {
        ldpc_mt::GenArgs ga{st, x, S, 0, 20, (long long)S * nwords};
        t = timeit([&] { hipLaunchKernelGGL((lib_gen<320>), dim3(S), dim3(320), 0, 0, ga); });
        printf("library body <320> (csrc/ldpc_mt.hpp)    : %8.3f ms  %6.2f TB/s\n", t, gb / t);
        t = timeit([&] { hipLaunchKernelGGL((lib_gen<640>), dim3(S), dim3(640), 0, 0, ga); });
        printf("library body <640>                       : %8.3f ms  %6.2f TB/s\n", t, gb / t);
        t = timeit([&] { hipLaunchKernelGGL((lib_gen<256>), dim3(S), dim3(256), 0, 0, ga); });
        printf("library body <256>                       : %8.3f ms  %6.2f TB/s\n", t, gb / t);
        t = timeit([&] { hipLaunchKernelGGL((lib_gen<192>), dim3(S), dim3(192), 0, 0, ga); });
        printf("library body <192>                       : %8.3f ms  %6.2f TB/s\n", t, gb / t);
        t = timeit([&] { hipLaunchKernelGGL((lib_gen<128>), dim3(S), dim3(128), 0, 0, ga); });
        printf("library body <128>                       : %8.3f ms  %6.2f TB/s\n", t, gb / t);
    }
    return 0;
}
