// ubench_valu.hip -- VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD for the
// instruction classes the min-sum kernel is made of, at 1/2/4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ void __launch_bounds__(64) k(unsigned *out, int iters, unsigned long long *cyc) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {  // 8 independent int adds
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(threadIdx.x));
            } else if (KIND == 1) {  // 8 independent f64 adds
                asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                             "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(1.0));
            } else if (KIND == 2) {  // 8 independent f64 min
                asm volatile("v_min_f64 %0, %0, %8\n v_min_f64 %1, %1, %8\n v_min_f64 %2, %2, %8\n v_min_f64 %3, %3, %8\n"
                             "v_min_f64 %4, %4, %8\n v_min_f64 %5, %5, %8\n v_min_f64 %6, %6, %8\n v_min_f64 %7, %7, %8"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(1e9));
            } else if (KIND == 3) {  // 8 independent cndmask (VOP3 with SGPR mask)
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(threadIdx.x) : "vcc");
            } else if (KIND == 4) {  // dependent f64 chain (latency)
                asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                             "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1"
                             : "+v"(d0) : "v"(1.0));
            } else if (KIND == 5) {  // dependent int chain
                asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                             "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1"
                             : "+v"(a0) : "v"(threadIdx.x));
            } else if (KIND == 6) {  // 8 independent v_bfi_b32 (VOP3, 3 operands)
                asm volatile("v_bfi_b32 %0, %8, %0, %1\n v_bfi_b32 %1, %8, %1, %2\n v_bfi_b32 %2, %8, %2, %3\n v_bfi_b32 %3, %8, %3, %4\n"
                             "v_bfi_b32 %4, %8, %4, %5\n v_bfi_b32 %5, %8, %5, %6\n v_bfi_b32 %6, %8, %6, %7\n v_bfi_b32 %7, %8, %7, %0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(threadIdx.x));
            } else if (KIND == 7) {  // 8 independent v_cmp_lt_f64 -> vcc
                asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %2\n v_cmp_lt_f64 vcc, %2, %3\n v_cmp_lt_f64 vcc, %3, %4\n"
                             "v_cmp_lt_f64 vcc, %4, %5\n v_cmp_lt_f64 vcc, %5, %6\n v_cmp_lt_f64 vcc, %6, %7\n v_cmp_lt_f64 vcc, %7, %0"
                             :: "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7) : "vcc");
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
int run(const char *name) {
    const int iters = 2000;
    unsigned *out; unsigned long long *cyc;
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * 4 * wps;  // 64-thread blocks: wps waves per SIMD on every CU
        CHK(hipMalloc(&out, blocks * 64 * 4)); CHK(hipMalloc(&cyc, blocks * 8));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, 10, cyc);
        CHK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc);
        CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks);
        CHK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        const double ninst = (double)iters * 64;  // per wave
        // s_memtime counts at 100 MHz?  report both: wall-based cycles at 2.4 GHz and memtime ticks
        printf("%-22s waves/SIMD %d: %.2f ms  | per-wave memtime ticks/inst %.3f | SIMD cycles/inst @2.4GHz (wall) %.3f\n", name, wps, ms,
               avg / ninst, ms * 1e-3 * 2.4e9 / (ninst * wps));
        CHK(hipFree(out)); CHK(hipFree(cyc));
    }
    return 0;
}

int main() {
    run<0>("v_add_u32 indep x8"); run<1>("v_add_f64 indep x8"); run<2>("v_min_f64 indep x8"); run<3>("v_cndmask indep x8");
    run<6>("v_bfi_b32 indep x8"); run<7>("v_cmp_lt_f64 indep x8"); run<4>("v_add_f64 dependent"); run<5>("v_add_u32 dependent");
    return 0;
}
