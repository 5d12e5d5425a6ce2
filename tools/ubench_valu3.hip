// ubench_valu3.hip -- select patterns: where does the VCC form of v_cndmask hurt? (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int KIND>
__global__ void __launch_bounds__(64) k(unsigned *out, int iters) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x55555555\n s_mov_b32 vcc_lo, 0x33333333\n s_mov_b32 vcc_hi, 0x33333333" ::: "s20", "s21", "vcc");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            // A: compare -> vcc, two e32 selects (x4 groups = 12 instructions)
            if (KIND == 0) asm volatile(
                "v_cmp_eq_u32 vcc, %0, %8\n v_cndmask_b32_e32 %1, %1, %2, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc\n"
                "v_cmp_eq_u32 vcc, %5, %8\n v_cndmask_b32_e32 %6, %6, %7, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n"
                "v_cmp_eq_u32 vcc, %0, %8\n v_cndmask_b32_e32 %1, %1, %2, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc\n"
                "v_cmp_eq_u32 vcc, %5, %8\n v_cndmask_b32_e32 %6, %6, %7, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc" : OPS : "v"(threadIdx.x) : "vcc");
            // B: compare -> sgpr pair, two e64 selects
            if (KIND == 1) asm volatile(
                "v_cmp_eq_u32_e64 s[20:21], %0, %8\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n"
                "v_cmp_eq_u32_e64 s[22:23], %5, %8\n v_cndmask_b32_e64 %6, %6, %7, s[22:23]\n v_cndmask_b32_e64 %2, %2, %4, s[22:23]\n"
                "v_cmp_eq_u32_e64 s[20:21], %0, %8\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n"
                "v_cmp_eq_u32_e64 s[22:23], %5, %8\n v_cndmask_b32_e64 %6, %6, %7, s[22:23]\n v_cndmask_b32_e64 %2, %2, %4, s[22:23]" : OPS : "v"(threadIdx.x) : "s20", "s21", "s22", "s23");
            // C: e64 select with vcc as the mask operand (12 selects)
            if (KIND == 2) asm volatile(
                "v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n"
                "v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc\n"
                "v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc" : OPS : "v"(threadIdx.x) : "vcc");
            // D: e32 selects whose two data sources differ from the destination (12 selects)
            if (KIND == 3) asm volatile(
                "v_cndmask_b32_e32 %0, %1, %2, vcc\n v_cndmask_b32_e32 %3, %4, %5, vcc\n v_cndmask_b32_e32 %6, %7, %1, vcc\n v_cndmask_b32_e32 %2, %4, %5, vcc\n"
                "v_cndmask_b32_e32 %0, %1, %2, vcc\n v_cndmask_b32_e32 %3, %4, %5, vcc\n v_cndmask_b32_e32 %6, %7, %1, vcc\n v_cndmask_b32_e32 %2, %4, %5, vcc\n"
                "v_cndmask_b32_e32 %0, %1, %2, vcc\n v_cndmask_b32_e32 %3, %4, %5, vcc\n v_cndmask_b32_e32 %6, %7, %1, vcc\n v_cndmask_b32_e32 %2, %4, %5, vcc" : OPS :: "vcc");
            // E: 12 x v_lshlrev_b32 with a REGISTER shift amount;  F: v_mov_b32;  G: v_add_u32 with SGPR operand; H: v_lshl_or_b32; I: v_sub_u32
            if (KIND == 4) asm volatile("v_lshlrev_b32 %0, %8, %0\n v_lshlrev_b32 %1, %8, %1\n v_lshlrev_b32 %2, %8, %2\n v_lshlrev_b32 %3, %8, %3\n v_lshlrev_b32 %4, %8, %4\n v_lshlrev_b32 %5, %8, %5\n v_lshlrev_b32 %6, %8, %6\n v_lshlrev_b32 %7, %8, %7\n v_lshlrev_b32 %0, %8, %0\n v_lshlrev_b32 %1, %8, %1\n v_lshlrev_b32 %2, %8, %2\n v_lshlrev_b32 %3, %8, %3" : OPS : "v"(threadIdx.x & 1));
            if (KIND == 5) asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4" : OPS);
            if (KIND == 6) asm volatile("v_add_u32 %0, s20, %0\n v_add_u32 %1, s20, %1\n v_add_u32 %2, s20, %2\n v_add_u32 %3, s20, %3\n v_add_u32 %4, s20, %4\n v_add_u32 %5, s20, %5\n v_add_u32 %6, s20, %6\n v_add_u32 %7, s20, %7\n v_add_u32 %0, s20, %0\n v_add_u32 %1, s20, %1\n v_add_u32 %2, s20, %2\n v_add_u32 %3, s20, %3" : OPS :: "s20");
            if (KIND == 7) asm volatile("v_and_b32 %0, 0x80000000, %0\n v_and_b32 %1, 0x80000000, %1\n v_and_b32 %2, 0x80000000, %2\n v_and_b32 %3, 0x80000000, %3\n v_and_b32 %4, 0x80000000, %4\n v_and_b32 %5, 0x80000000, %5\n v_and_b32 %6, 0x80000000, %6\n v_and_b32 %7, 0x80000000, %7\n v_and_b32 %0, 0x80000000, %0\n v_and_b32 %1, 0x80000000, %1\n v_and_b32 %2, 0x80000000, %2\n v_and_b32 %3, 0x80000000, %3" : OPS);
            if (KIND == 8) asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3\n v_add_u32 %4, %4, %4\n v_add_u32 %5, %5, %5\n v_add_u32 %6, %6, %6\n v_add_u32 %7, %7, %7\n v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3" : OPS);
            if (KIND == 9) asm volatile("v_cmp_gt_i32 vcc, 0, %0\n v_cmp_gt_i32 vcc, 0, %1\n v_cmp_gt_i32 vcc, 0, %2\n v_cmp_gt_i32 vcc, 0, %3\n v_cmp_gt_i32 vcc, 0, %4\n v_cmp_gt_i32 vcc, 0, %5\n v_cmp_gt_i32 vcc, 0, %6\n v_cmp_gt_i32 vcc, 0, %7\n v_cmp_gt_i32 vcc, 0, %0\n v_cmp_gt_i32 vcc, 0, %1\n v_cmp_gt_i32 vcc, 0, %2\n v_cmp_gt_i32 vcc, 0, %3" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int KIND>
int run(const char *name) {
    const int iters = 1000; unsigned *out;
    for (int wps : {2, 4}) {
        const int blocks = 256 * 4 * wps;
        CHK(hipMalloc(&out, blocks * 64 * 4));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, 10); CHK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1)); CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters);
        CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s waves/SIMD %d: %.3f ms  SIMD cycles/inst @2.4GHz %.2f\n", name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 96 * wps));
        CHK(hipFree(out));
    }
    return 0;
}
int main() {
    run<0>("cmp->vcc + 2x cndmask_e32(vcc)"); run<1>("cmp->sgpr + 2x cndmask_e64(sgpr)"); run<2>("cndmask_e64 with vcc mask");
    run<3>("cndmask_e32(vcc), dst != srcs"); run<4>("v_lshlrev_b32 reg shift"); run<5>("v_mov_b32"); run<6>("v_add_u32 sgpr operand");
    run<7>("v_and_b32 literal"); run<8>("v_add_u32 x+x (shift left 1)"); run<9>("v_cmp_gt_i32 0 -> vcc");
    return 0;
}
