// ubench_valu2.hip -- second round: issue cost of the exact instruction forms the min-sum kernel uses (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define R8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
template <int KIND>
__global__ void __launch_bounds__(64) k(unsigned *out, int iters) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    extern __shared__ double lds[];
    lds[threadIdx.x] = 1.0; lds[threadIdx.x + 64] = 2.0;
    const unsigned addr = threadIdx.x * 8;
    asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x55555555\n s_mov_b32 vcc_lo, 0x33333333\n s_mov_b32 vcc_hi, 0x33333333" ::: "s20", "s21", "vcc");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define DOPS "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
            if (KIND == 0) asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc" : OPS : "v"(threadIdx.x) : "vcc");
            if (KIND == 1) asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]" : OPS : "v"(threadIdx.x) : "s20", "s21");
            if (KIND == 2) asm volatile("v_and_or_b32 %0, %0, %8, %1\n v_and_or_b32 %1, %1, %8, %2\n v_and_or_b32 %2, %2, %8, %3\n v_and_or_b32 %3, %3, %8, %4\n v_and_or_b32 %4, %4, %8, %5\n v_and_or_b32 %5, %5, %8, %6\n v_and_or_b32 %6, %6, %8, %7\n v_and_or_b32 %7, %7, %8, %0" : OPS : "v"(threadIdx.x));
            if (KIND == 3) asm volatile("v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %1, %1, %2, 31\n v_alignbit_b32 %2, %2, %3, 31\n v_alignbit_b32 %3, %3, %4, 31\n v_alignbit_b32 %4, %4, %5, 31\n v_alignbit_b32 %5, %5, %6, 31\n v_alignbit_b32 %6, %6, %7, 31\n v_alignbit_b32 %7, %7, %0, 31" : OPS);
            if (KIND == 4) asm volatile("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n v_lshlrev_b32 %4, 3, %4\n v_lshlrev_b32 %5, 3, %5\n v_lshlrev_b32 %6, 3, %6\n v_lshlrev_b32 %7, 3, %7" : OPS);
            if (KIND == 5) asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_cmp_eq_u32 vcc, %1, %2\n v_cmp_eq_u32 vcc, %2, %3\n v_cmp_eq_u32 vcc, %3, %4\n v_cmp_eq_u32 vcc, %4, %5\n v_cmp_eq_u32 vcc, %5, %6\n v_cmp_eq_u32 vcc, %6, %7\n v_cmp_eq_u32 vcc, %7, %0" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");
            if (KIND == 6) asm volatile("v_cmp_eq_u32_e64 s[20:21], %0, %1\n v_cmp_eq_u32_e64 s[20:21], %1, %2\n v_cmp_eq_u32_e64 s[20:21], %2, %3\n v_cmp_eq_u32_e64 s[20:21], %3, %4\n v_cmp_eq_u32_e64 s[20:21], %4, %5\n v_cmp_eq_u32_e64 s[20:21], %5, %6\n v_cmp_eq_u32_e64 s[20:21], %6, %7\n v_cmp_eq_u32_e64 s[20:21], %7, %0" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "s20", "s21");
            if (KIND == 7) asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : DOPS : "v"(1.0000001));
            if (KIND == 8) asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" : OPS : "v"(threadIdx.x));
            if (KIND == 9) asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8\n ds_read_b64 %3, %8 offset:512\n ds_read_b64 %4, %8\n ds_read_b64 %5, %8 offset:512\n ds_read_b64 %6, %8\n ds_read_b64 %7, %8 offset:512\n s_waitcnt lgkmcnt(0)" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7) : "v"(addr) : "memory");
            if (KIND == 10) asm volatile("ds_add_f64 %8, %0\n ds_add_f64 %8, %1 offset:512\n ds_add_f64 %8, %2\n ds_add_f64 %8, %3 offset:512\n ds_add_f64 %8, %4\n ds_add_f64 %8, %5 offset:512\n ds_add_f64 %8, %6\n ds_add_f64 %8, %7 offset:512\n s_waitcnt lgkmcnt(0)" :: "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7), "v"(addr) : "memory");
            if (KIND == 11) asm volatile("ds_write_b64 %8, %0\n ds_write_b64 %8, %1 offset:512\n ds_write_b64 %8, %2\n ds_write_b64 %8, %3 offset:512\n ds_write_b64 %8, %4\n ds_write_b64 %8, %5 offset:512\n ds_write_b64 %8, %6\n ds_write_b64 %8, %7 offset:512\n s_waitcnt lgkmcnt(0)" :: "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7), "v"(addr) : "memory");
            if (KIND == 12) asm volatile("v_cmp_eq_u32_sdwa vcc, %0, %1 src0_sel:WORD_1 src1_sel:DWORD\n v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:WORD_1 src1_sel:DWORD\n v_cmp_eq_u32_sdwa vcc, %2, %3 src0_sel:WORD_1 src1_sel:DWORD\n v_cmp_eq_u32_sdwa vcc, %3, %4 src0_sel:WORD_1 src1_sel:DWORD\n v_cmp_eq_u32_sdwa vcc, %4, %5 src0_sel:WORD_1 src1_sel:DWORD\n v_cmp_eq_u32_sdwa vcc, %5, %6 src0_sel:WORD_1 src1_sel:DWORD\n v_cmp_eq_u32_sdwa vcc, %6, %7 src0_sel:WORD_1 src1_sel:DWORD\n v_cmp_eq_u32_sdwa vcc, %7, %0 src0_sel:WORD_1 src1_sel:DWORD" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");
            if (KIND == 13) asm volatile("v_bfi_b32 %0, s20, %0, %1\n v_bfi_b32 %1, s20, %1, %2\n v_bfi_b32 %2, s20, %2, %3\n v_bfi_b32 %3, s20, %3, %4\n v_bfi_b32 %4, s20, %4, %5\n v_bfi_b32 %5, s20, %5, %6\n v_bfi_b32 %6, s20, %6, %7\n v_bfi_b32 %7, s20, %7, %0" : OPS :: "s20");
            if (KIND == 14) asm volatile("v_and_b32 %0, %0, %8\n v_or_b32 %0, %0, %1\n v_and_b32 %2, %2, %8\n v_or_b32 %2, %2, %3\n v_and_b32 %4, %4, %8\n v_or_b32 %4, %4, %5\n v_and_b32 %6, %6, %8\n v_or_b32 %6, %6, %7" : OPS : "v"(threadIdx.x));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (unsigned)lds[threadIdx.x];
}

template <int KIND>
int run(const char *name) {
    const int iters = 1000;
    unsigned *out;
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * 4 * wps;
        CHK(hipMalloc(&out, blocks * 64 * 4));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 1024, 0, out, 10);
        CHK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 1024, 0, out, iters);
        CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        const double ninst = (double)iters * 64;
        printf("%-28s waves/SIMD %d: %.3f ms  SIMD cycles/inst @2.4GHz %.2f\n", name, wps, ms, ms * 1e-3 * 2.4e9 / (ninst * wps));
        CHK(hipFree(out));
    }
    return 0;
}

int main() {
    run<0>("v_cndmask_e32 vcc"); run<1>("v_cndmask_e64 sgpr"); run<2>("v_and_or_b32"); run<3>("v_alignbit_b32"); run<4>("v_lshlrev_b32");
    run<5>("v_cmp_eq_u32 ->vcc"); run<6>("v_cmp_eq_u32_e64 ->sgpr"); run<12>("v_cmp_eq_u32_sdwa"); run<7>("v_mul_f64"); run<8>("v_xor_b32");
    run<13>("v_bfi_b32 (sgpr mask)"); run<14>("v_and+v_or (2 VOP2)");
    run<9>("ds_read_b64 (per CU: /4)"); run<10>("ds_add_f64"); run<11>("ds_write_b64");
    return 0;
}
