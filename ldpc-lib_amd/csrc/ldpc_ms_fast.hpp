// ldpc_ms_fast.hpp -- the flagship kernel: flooding min-sum for lifting M = 64 (one frame == one wavefront).
//
// Same algorithm and bit-exact results as ms_flood_kernel (ldpc_kernels.hpp; upstream decoders.cpp:4554-4767), but
// specialised so that the inner loops are straight-line code with everything about an edge known at compile time
// except its shift and block column:
//   * the code table travels BY VALUE in the kernel-argument segment -> the descriptors of a block row arrive with
//     one s_load_dwordx8 into SGPRs (no per-edge vector or scalar memory latency in the loops);
//   * block rows and the first 8 circulants of each row are statically unrolled: record registers, the
//     edge's bit position in the row word and the "is this the min1 edge" test are immediates;
//   * a row's LDS reads are all issued before the first dependent use (8 loads in flight per wave);
//   * STATE1 scatter-adds with LDS fp64 atomics (ds_add_f64, IEEE round-to-nearest like v_add_f64; LDS operations of
//     one wave execute in order, so the per-variable sum still runs in ascending block-row order from the first
//     term, exactly as the reference's `0.0 + c2v_0 + c2v_1 ...`); the first edge of every block column stores
//     instead of adding, which removes the zeroing pass (0.0 + x == x for every x but -0.0, see ldpc_kernels.hpp);
//   * |v2c| clamping to MAX_VAL (decoders.cpp:4730) is folded into the min1/min2 initial value: min(MAX, v...) is
//     the same number as min over clamped v, and `clamp(v) < min1` equals `v < min1` because min1 <= MAX always.
// LDS: 16 KiB per wave (soft/acc fp64), 8 waves per CU.  VGPR budget <= 256 (2 waves per SIMD).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ldpc_kernels.hpp"

namespace ldpc {

constexpr int kFastRows = 16;   // block rows
constexpr int kFastCols = 32;   // block columns
constexpr int kFastSlots = 8;   // circulants per block row (all unrolled)

// 16-bit edge descriptors, two per dword: [15] valid, [11] first edge of its block column, [10:6] block column,
// [5:0] shift.  64 dwords in all: they are loaded once and stay in SGPRs for the whole kernel.
struct FastTab {
    uint32_t pk[kFastRows][kFastSlots / 2];
};
__host__ __device__ inline uint32_t fast_desc(uint32_t first, uint32_t k, uint32_t shift) {
    return 0x8000u | (first << 11) | (k << 6) | shift;
}

__device__ __forceinline__ double lds_ld(const char *b, uint32_t off) { return *reinterpret_cast<const double *>(b + off); }
__device__ __forceinline__ void lds_st(char *b, uint32_t off, double v) { *reinterpret_cast<double *>(b + off) = v; }
__device__ __forceinline__ void lds_add(char *b, uint32_t off, double v) {
    __hip_atomic_fetch_add(reinterpret_cast<double *>(b + off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// magnitude bits of `mag_hi` (>= 0) combined with the sign bit of `sign_src`
__device__ __forceinline__ uint32_t with_sign(uint32_t mag_hi, uint32_t sign_src) {
    return (mag_hi & 0x7fffffffu) | (sign_src & 0x80000000u);
}
__device__ __forceinline__ uint32_t edge_addr(uint32_t n8, uint32_t d) {
    return ((n8 + ((d & 63u) << 3)) & 511u) | (((d >> 6) & 31u) << 9);
}
__device__ __forceinline__ bool edge_valid(uint32_t d) { return (d & 0x8000u) != 0; }
__device__ __forceinline__ bool edge_first(uint32_t d) { return (d & 0x0800u) != 0; }

template <bool ATOMIC>
__global__ void __launch_bounds__(64, 2) ms_flood_m64_kernel(const DecArgs a, const FastTab t) {
    extern __shared__ double lds[];  // [2048] soft / acc
    char *const ldsb = reinterpret_cast<char *>(lds);
    const int lane = threadIdx.x;
    const uint32_t n8 = (uint32_t)lane * 8u;
    const int rh = a.rh, nh = a.nh, N = a.N;
    const double alpha = a.alpha;
    const long long fr = blockIdx.x;  // grid == B: one workgroup (one wave) per frame

    uint32_t tab[kFastRows][kFastSlots / 2];  // uniform: SGPRs
#pragma unroll
    for (int j = 0; j < kFastRows; ++j)
#pragma unroll
        for (int q = 0; q < kFastSlots / 2; ++q) tab[j][q] = t.pk[j][q];

    double y[kFastCols];
#pragma unroll
    for (int k = 0; k < kFastCols; ++k) y[k] = (k < nh) ? a.llr[fr * N + k * 64 + lane] + 0.0 : 0.0;

    double m1[kFastRows], m2[kFastRows];
    uint32_t meta[kFastRows];  // [15:0] v2c sign bit per slot, [23:16] slot of the min1 edge
#pragma unroll
    for (int j = 0; j < kFastRows; ++j) { m1[j] = 0.0; m2[j] = 0.0; meta[j] = 0u; }

    int res = -a.maxiter;
    for (int iter = 0; iter < a.maxiter; ++iter) {
        // ---------------- STATE1 (:4633-4667): acc[v] = sum of c2v in ascending block row
#pragma unroll
        for (int j = 0; j < kFastRows; ++j) {
            if (j < rh) {
                const uint32_t mt = meta[j];
                const uint32_t pos = mt >> 16;
                // bit s of W = sign of the c2v on slot s = (own v2c sign) xor (row sign)
                const uint32_t W = (mt & 0xffffu) ^ (0u - (__popc(mt & 0xffffu) & 1u));
                uint32_t tw[kFastSlots / 2];
#pragma unroll
                for (int q = 0; q < kFastSlots / 2; ++q) {
                    tw[q] = tab[j][q];
                    asm volatile("" : "+s"(tw[q]));  // opaque: keeps the per-edge scalar decode inside the loop (no LICM blow-up)
                }
#pragma unroll
                for (int s = 0; s < kFastSlots; ++s) {
                    const uint32_t d = tw[s >> 1] >> ((s & 1) * 16);
                    if (edge_valid(d)) {
                        const double aa = (pos == (uint32_t)s) ? m2[j] : m1[j];
                        const double cv = mkdouble(with_sign(hi32(aa), W << (31 - s)), lo32(aa));
                        const uint32_t ad = edge_addr(n8, d);
                        if (edge_first(d)) lds_st(ldsb, ad, cv);
                        else if (ATOMIC) lds_add(ldsb, ad, cv);
                        else lds_st(ldsb, ad, lds_ld(ldsb, ad) + cv);
                    }
                }
            }
        }
        // ---------------- STATE2 (:4670-4685): soft = y + acc*alpha, two roundings
#pragma unroll
        for (int k = 0; k < kFastCols; ++k) {
            if (k < nh) {
                const double p = lds_ld(ldsb, n8 + k * 512) * alpha;
                lds_st(ldsb, n8 + k * 512, y[k] + p);
            }
        }
        // ---------------- STATE3 (:4690-4755)
        uint32_t failw = 0;
#pragma unroll
        for (int j = 0; j < kFastRows; ++j) {
            if (j < rh) {
                const uint32_t mt = meta[j];
                const uint32_t pos = mt >> 16;
                const uint32_t W = (mt & 0xffffu) ^ (0u - (__popc(mt & 0xffffu) & 1u));
                double a1 = m1[j] * alpha, a2 = m2[j] * alpha;
                asm volatile("" : "+v"(a1), "+v"(a2));  // keep the two products per ROW (not one per edge)
                double nm1 = kMaxVal, nm2 = kMaxVal;    // also performs the MAX_VAL clamp (see header)
                uint32_t npos = 0, nS = 0, sy = 0;
                uint32_t tw[kFastSlots / 2];
#pragma unroll
                for (int q = 0; q < kFastSlots / 2; ++q) {
                    tw[q] = tab[j][q];
                    asm volatile("" : "+s"(tw[q]));
                }
                double r[kFastSlots];
#pragma unroll
                for (int s = 0; s < kFastSlots; ++s) {
                    const uint32_t d = tw[s >> 1] >> ((s & 1) * 16);
                    r[s] = 0.0;
                    if (edge_valid(d)) r[s] = lds_ld(ldsb, edge_addr(n8, d));
                }
#pragma unroll
                for (int s = 0; s < kFastSlots; ++s) {
                    const uint32_t d = tw[s >> 1] >> ((s & 1) * 16);
                    if (edge_valid(d)) {
                        sy ^= hi32(r[s]);
                        const double aa = (pos == (uint32_t)s) ? a2 : a1;
                        const double x = mkdouble(with_sign(hi32(aa), W << (31 - s)), lo32(aa));
                        const double tt = r[s] - x;                 // v2c
                        nS |= (hi32(tt) >> 31) << s;
                        const double v = fabs(tt);
                        const bool c1 = v < nm1;                    // strict: the first minimum keeps the position
                        nm2 = fmin(fmax(v, nm1), nm2);
                        npos = c1 ? (uint32_t)s : npos;
                        nm1 = fmin(v, nm1);
                    }
                }
                failw |= sy;
                m1[j] = nm1; m2[j] = nm2; meta[j] = nS | (npos << 16);
            }
        }
        if (__ballot((failw >> 31) != 0) == 0ull) { res = iter + 1; break; }  // :4761-4766
    }

    // ---------------- outputs
    if (lane == 0 && a.iters) a.iters[fr] = res;
    if (a.hard) {
        unsigned long long mine = 0ull;
#pragma unroll
        for (int k = 0; k < kFastCols; ++k) {
            if (k < nh) {
                const unsigned long long b = __ballot((hi32(lds_ld(ldsb, n8 + k * 512)) >> 31) != 0);
                if (lane == k) mine = b;
            }
        }
        // block column k = variables 64k..64k+63 = hard words 2k, 2k+1: lanes 0..nh-1 write one 8-byte pair each
        if (lane < nh) reinterpret_cast<unsigned long long *>(a.hard + fr * a.hard_words)[lane] = mine;
    }
    if (a.soft_out) {
#pragma unroll
        for (int k = 0; k < kFastCols; ++k)
            if (k < nh) a.soft_out[fr * N + k * 64 + lane] = lds_ld(ldsb, n8 + k * 512);
    }
}

}  // namespace ldpc
