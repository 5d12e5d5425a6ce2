// ubench_valu4.hip -- issue cost of the integer / packed-16 / byte-permute instructions an int8/int16 min-sum would use (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
// 12 independent instructions: two-source form "op d, d, x" over 8 registers (+4 repeats)
#define R2(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" \
               op " %6, %6, %8\n" op " %7, %7, %8\n" op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n"
#define R3(op) op " %0, %0, %8, %1\n" op " %1, %1, %8, %2\n" op " %2, %2, %8, %3\n" op " %3, %3, %8, %4\n" op " %4, %4, %8, %5\n" op " %5, %5, %8, %6\n" \
               op " %6, %6, %8, %7\n" op " %7, %7, %8, %0\n" op " %0, %0, %8, %1\n" op " %1, %1, %8, %2\n" op " %2, %2, %8, %3\n" op " %3, %3, %8, %4\n"
template <int KIND>
__global__ void __launch_bounds__(64) k(unsigned *out, int iters) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned x = threadIdx.x * 3 + 1;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) asm volatile(R2("v_pk_add_i16") : OPS : "v"(x));
            if (KIND == 1) asm volatile(R2("v_pk_min_i16") : OPS : "v"(x));
            if (KIND == 2) asm volatile(R2("v_pk_max_u16") : OPS : "v"(x));
            if (KIND == 3) asm volatile(R2("v_pk_sub_i16") : OPS : "v"(x));
            if (KIND == 4) asm volatile(R3("v_pk_mad_u16") : OPS : "v"(x));
            if (KIND == 5) asm volatile(R2("v_pk_lshlrev_b16") : OPS : "v"(x));
            if (KIND == 6) asm volatile(R3("v_med3_i32") : OPS : "v"(x));
            if (KIND == 7) asm volatile(R3("v_perm_b32") : OPS : "v"(x));
            if (KIND == 8) asm volatile(R3("v_bfe_i32") : OPS : "v"(x));
            if (KIND == 9) asm volatile(R3("v_lshl_or_b32") : OPS : "v"(x));
            if (KIND == 10) asm volatile(R2("v_mul_u32_u24") : OPS : "v"(x));
            if (KIND == 11) asm volatile(R3("v_sad_u32") : OPS : "v"(x));
            if (KIND == 12) asm volatile(R2("v_min_i32") : OPS : "v"(x));
            if (KIND == 13) asm volatile(R2("v_max_i32") : OPS : "v"(x));
            if (KIND == 14) asm volatile(R2("v_xor_b32") : OPS : "v"(x));
            if (KIND == 15) asm volatile(R2("v_sub_u32") : OPS : "v"(x));
            if (KIND == 16) asm volatile(R2("v_ashrrev_i32") : OPS : "v"(x));
            if (KIND == 17) asm volatile(R3("v_bfi_b32") : OPS : "v"(x));
            if (KIND == 18) asm volatile(R3("v_add3_u32") : OPS : "v"(x));
            if (KIND == 19) asm volatile(R2("v_mul_lo_u32") : OPS : "v"(x));
            if (KIND == 20) asm volatile(R3("v_min3_i32") : OPS : "v"(x));
            if (KIND == 21) asm volatile(R2("v_pk_ashrrev_i16") : OPS : "v"(x));
            if (KIND == 22) asm volatile(R3("v_xad_u32") : OPS : "v"(x));
            if (KIND == 23) asm volatile(R3("v_and_or_b32") : OPS : "v"(x));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
// LDS: 12 byte / dword reads or writes per group, consecutive lanes -> consecutive elements
template <int KIND>
__global__ void __launch_bounds__(64) kl(unsigned *out, int iters) {
    __shared__ unsigned lds[4096];
    unsigned acc = 0;
    const unsigned b1 = threadIdx.x, b4 = threadIdx.x * 4;
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            unsigned v;
            if (KIND == 0) { asm volatile("ds_read_i8 %0, %1 offset:%2" : "=v"(v) : "v"(b1), "i"(u * 64)); acc += v; }
            if (KIND == 1) { asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(b4), "i"(u * 256)); acc += v; }
            if (KIND == 2) asm volatile("ds_write_b8 %0, %1 offset:%2" :: "v"(b1), "v"(acc), "i"(u * 64));
            if (KIND == 3) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(b4), "v"(acc), "i"(u * 256));
            if (KIND == 4) { asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(v) : "v"(b1 * 2), "i"(u * 128)); acc += v; }
            if (KIND == 5) asm volatile("ds_write_b16 %0, %1 offset:%2" :: "v"(b1 * 2), "v"(acc), "i"(u * 128));
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc + lds[threadIdx.x];
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
        printf("%-28s waves/SIMD %d: %.3f ms  SIMD cycles/inst @2.4GHz %.2f\n", name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 96 * wps));
        CHK(hipFree(out));
    }
    return 0;
}
template <int KIND>
int runl(const char *name) {
    const int iters = 2000; unsigned *out;
    for (int wps : {2, 4}) {
        const int blocks = 256 * 4 * wps;   // wps waves per SIMD = 4*wps waves per CU
        CHK(hipMalloc(&out, blocks * 64 * 4));
        hipLaunchKernelGGL(kl<KIND>, dim3(blocks), dim3(64), 0, 0, out, 10); CHK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1)); CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(kl<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters);
        CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s waves/CU %d: %.3f ms  CU cycles per wave-instruction @2.4GHz %.2f\n", name, 4 * wps, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 12 * 4 * wps));
        CHK(hipFree(out));
    }
    return 0;
}
int main() {
    run<0>("v_pk_add_i16"); run<1>("v_pk_min_i16"); run<2>("v_pk_max_u16"); run<3>("v_pk_sub_i16"); run<4>("v_pk_mad_u16"); run<5>("v_pk_lshlrev_b16");
    run<6>("v_med3_i32"); run<7>("v_perm_b32"); run<8>("v_bfe_i32"); run<9>("v_lshl_or_b32"); run<10>("v_mul_u32_u24"); run<11>("v_sad_u32");
    run<12>("v_min_i32"); run<13>("v_max_i32"); run<14>("v_xor_b32"); run<15>("v_sub_u32"); run<16>("v_ashrrev_i32"); run<17>("v_bfi_b32");
    run<18>("v_add3_u32"); run<19>("v_mul_lo_u32"); run<20>("v_min3_i32"); run<21>("v_pk_ashrrev_i16"); run<22>("v_xad_u32"); run<23>("v_and_or_b32");
    runl<0>("ds_read_i8"); runl<1>("ds_read_b32"); runl<2>("ds_write_b8"); runl<3>("ds_write_b32"); runl<4>("ds_read_u16"); runl<5>("ds_write_b16");
    return 0;
}
