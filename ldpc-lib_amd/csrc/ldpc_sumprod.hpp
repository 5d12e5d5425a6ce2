// ldpc_sumprod.hpp -- flooding sum-product in the likelihood-ratio domain for gfx950.
//
// Restates sum_prod_decod_qc_lm (decoders.cpp:1923-2185; semantics SURVEY Appendix A.2) as four lane-parallel
// phases per iteration.  The CPU code walks block columns serially; the only order that matters numerically is
// the order of the floating-point products, and that is preserved exactly:
//   A (variable lanes)  for every edge q of the variable's column: AA = yd * prod_{q' != q, rows ascending} ZZ[q'],
//                       ZZ[q] <- (AA-1)/(AA+1)            (:2017-2060; old ZZ of the column is read before any write)
//   B (check lanes)     s = 1.0 * prod_{edges of the row, columns ascending} ZZ[e][(n+c) mod M]      (:2047-2050)
//   C (variable lanes)  soft = yd; for q rows ascending: A = s[(t-c) mod M] / ZZ[q]; A = (1+A)/(1-A);
//                       clamp to [-5.2e-9 (sic), 1.9e8]; ZZ[q] <- A; soft *= A                        (:2103-2127)
//   D (check lanes)     syndrome = xor over the row's edges of (soft < 1.0)                           (:2129-2149)
// One frame per 256-thread workgroup.  LDS holds the per-edge messages ZZ[ne][M] (fp64), the check products
// s[R] and one hard-decision byte per variable; yd and soft of a thread's <= 8 variables stay in VGPRs.
// exp() is evaluated with glibc's algorithm (ldpc_spec::exp_glibc), so soft values equal the CPU reference's bitwise.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ldpc_kernels.hpp"

namespace ldpc {

constexpr int kSpThreads = 256;
constexpr int kSpNVM = 8;   // variables (and checks) per thread held in registers: N <= 8*256

__host__ __device__ inline size_t sp_lds_bytes(int ne, int M, int R, int N) {
    return sizeof(double) * ((size_t)ne * M + R) + (((size_t)N + 15) & ~(size_t)15) + 16;
}

__device__ __forceinline__ double sp_mind(double a, double b) { return a < b ? a : b; }  // decoders.cpp:104
__device__ __forceinline__ double sp_maxd(double a, double b) { return a < b ? b : a; }  // decoders.cpp:105

__global__ void __launch_bounds__(kSpThreads) sp_flood_kernel(const DecArgs a) {
    extern __shared__ double lds[];
    const int M = a.M, N = a.N, R = a.rh * a.M, T = kSpThreads;
    const int ne = a.col_start[a.nh];
    double *ZZ = lds;                                  // [ne][M]
    double *S = lds + (size_t)ne * M;                  // [R]
    unsigned char *hb = (unsigned char *)(S + R);      // [N] soft < 1.0
    int *sh_flag = (int *)(hb + (((size_t)N + 15) & ~(size_t)15));
    const int tid = threadIdx.x;
    const long long fr = blockIdx.x;
    if (fr >= a.B) return;  // uniform per workgroup

    double yd[kSpNVM], sf[kSpNVM];
#pragma unroll
    for (int q = 0; q < kSpNVM; ++q) {
        const int v = tid + q * T;
        yd[q] = 1.0; sf[q] = 1.0;
        if (v < N) {
            const double yl = sp_maxd(sp_mind(a.llr[fr * N + v], 20.0), -20.0);  // :1949 INPUT_LIMIT
            yd[q] = sf[q] = ldpc_spec::exp_glibc(yl);   // the reference's exp(), bit for bit (ldpc_spec.hpp)
            hb[v] = sf[q] < 1.0;
        }
    }
    for (int i = tid; i < ne * M; i += T) ZZ[i] = 1.0;  // :1957-1959
    __syncthreads();

    auto syndrome_fail = [&]() -> bool {               // phase D
        bool f = false;
#pragma unroll
        for (int q = 0; q < kSpNVM; ++q) {
            const int r = tid + q * T;
            if (r < R) {
                const int j = r / M, n = r - j * M;
                unsigned sy = 0;
                for (int e = a.row_start[j]; e < a.row_start[j + 1]; ++e) {
                    const uint32_t d = a.edges[e];
                    sy ^= hb[(d >> 16) * M + rot_idx(n, d & 0xffffu, M)];
                }
                f |= sy != 0;
            }
        }
        return f;
    };

    int res = -a.maxiter;
    bool conv = !frame_vote<true>(syndrome_fail(), 1, 0, 0ull, sh_flag);  // :1964-2002
    if (conv) res = 0;
    for (int iter = 0; !conv && iter < a.maxiter; ++iter) {
        // ---- phase A
#pragma unroll
        for (int q = 0; q < kSpNVM; ++q) {
            const int v = tid + q * T;
            if (v < N) {
                const int k = v / M, t = v - k * M;
                const int c0 = a.col_start[k], c1 = a.col_start[k + 1];
                // AA_u = yd * zz[0] * .. * zz[u-1] * zz[u+1] * .. in that order (:2027-2041).  The prefix is carried
                // in a register, the tail is read from the not-yet-overwritten entries, so ZZ is updated in place.
                double prefix = yd[q];
                for (int u = c0; u < c1; ++u) {
                    const int zi = a.col_slot[u] * M + t;
                    const double orig = ZZ[zi];
                    double AA = prefix;
                    for (int w = u + 1; w < c1; ++w) AA *= ZZ[a.col_slot[w] * M + t];
                    ZZ[zi] = (AA - 1) / (AA + 1);                    // :2044
                    prefix *= orig;
                }
            }
        }
        __syncthreads();
        // ---- phase B
#pragma unroll
        for (int q = 0; q < kSpNVM; ++q) {
            const int r = tid + q * T;
            if (r < R) {
                const int j = r / M, n = r - j * M;
                double s = 1.0;                                      // :2010
                for (int e = a.row_start[j]; e < a.row_start[j + 1]; ++e) {
                    const uint32_t d = a.edges[e];
                    s *= ZZ[e * M + rot_idx(n, d & 0xffffu, M)];     // :2047-2050
                }
                S[r] = s;
            }
        }
        __syncthreads();
        // ---- phase C
#pragma unroll
        for (int q = 0; q < kSpNVM; ++q) {
            const int v = tid + q * T;
            if (v < N) {
                const int k = v / M, t = v - k * M;
                double soft = yd[q];                                 // :2011
                for (int u = a.col_start[k]; u < a.col_start[k + 1]; ++u) {
                    const uint32_t d = a.col_edges[u];
                    const int j = d >> 16, c = d & 0xffffu;
                    int nn = t - c; if (nn < 0) nn += M;             // rotate by M-circ (:2113)
                    const int zi = a.col_slot[u] * M + t;
                    double A = S[j * M + nn] / ZZ[zi];
                    A = (1 + A) / (1 - A);
                    A = sp_maxd(sp_mind(A, 1.9e+8), -5.2e-9);        // :2120 (negative lower clamp is upstream's)
                    ZZ[zi] = A;
                    soft *= A;
                }
                sf[q] = soft;
                hb[v] = soft < 1.0;
            }
        }
        __syncthreads();
        // ---- phase D
        if (!frame_vote<true>(syndrome_fail(), 1, 0, 0ull, sh_flag)) { conv = true; res = iter + 1; }  // :2151-2166
    }

    if (tid == 0 && a.iters) a.iters[fr] = res;
    if (a.hard) {
        for (int w = tid; w < a.hard_words; w += T) {
            uint32_t bits = 0;
            for (int b = 0; b < 32; ++b) {
                const int v = 32 * w + b;
                if (v < N) bits |= (uint32_t)hb[v] << b;
            }
            a.hard[fr * a.hard_words + w] = bits;
        }
    }
    if (a.soft_out) {
#pragma unroll
        for (int q = 0; q < kSpNVM; ++q) {
            const int v = tid + q * T;
            if (v < N) a.soft_out[fr * N + v] = sf[q];
        }
    }
}

}  // namespace ldpc
