// ldpc_global.hpp -- the shape-unlimited tier: min-sum, integer min-sum, layered min-sum, sum-product and TDMP sum-product with the message state in
// GLOBAL memory.
//
// The LDS/VGPR-resident kernels (ldpc_spec.hpp, ldpc_kernels.hpp) need the a-posteriori values of a frame in one CU's LDS
// (N * 8 B <= 160 KiB), M <= 512, <= 64 block rows / columns, row weight <= 16 and no empty block column.  Upstream's decod_open
// (decoders.cpp:348-791) has none of these limits, so any other valid binary QC-LDPC code runs HERE instead of being refused:
// one workgroup of 256 threads per frame, soft values / check records / edge signs in a per-workgroup slice of a workspace in
// HBM (it is re-read every iteration, so it lives in L2 / Infinity Cache for the usual sizes), workgroup barriers between the
// phases.  The grid is capped and strides over the frames, so the workspace does not grow with the batch.
//
//   sp_global_kernel   sum_prod_decod_qc_lm  decoders.cpp:1923-2185: the four phases of ldpc_sumprod.hpp, arrays in the workspace.
//   ims_global_kernel  imin_sum_decod_qc_lm  decoders.cpp:5430-5690: min-sum on ints, saturation after every add of STATE1 (:5568).
//   bp_global_kernel   bp_decod_qc_lm        decoders.cpp:1708-1920, with the frame chain of the resident kernel.
//   asp_global_kernel  sum_prod_gf2_decod_qc_lm  decoders.cpp:2324-2581 (the general branch, and the branch for codes whose block
//                      columns all hold exactly two circulants, :2431-2480).
//   tasp_global_kernel tdmp_sum_prod_gf2_decod_qc_lm  decoders.cpp:2584-2744 (decoder 7, the decoder of upstream's shipped scenarios):
//                      per-edge lambda / rho / forward / backward products in the workspace instead of VGPRs, so row weight and
//                      the number of circulants are unbounded (the resident tasp_body holds the Z state of ~300 circulants in the registers of two lanes per check).
// The arithmetic is upstream's, literally (comparisons `< 0`, explicit branches, additions in upstream's order):
//   ms_global_kernel   min_sum_decod_qc_lm   decoders.cpp:4554-4767.  STATE1 is done from the VARIABLE side (a thread walks its
//                      column's circulants in ascending block row = upstream's order of additions into soft[], :4633-4667), which
//                      needs no atomics and no zeroing pass; STATE3 from the check side.
//   lms_global_kernel  lmin_sum_decod_qc_lm  decoders.cpp:5064-5425 (MY_VERSION branch): layers strictly in sequence, inside a
//                      layer every variable belongs to exactly one check.
// Bit-identical hard decisions, return values and soft values (tests/test_gpu_shapes.py: against the CPU oracle on shapes the
// resident kernels refuse, and against the golden vectors of the compiled reference with the tier forced on).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ldpc_kernels.hpp"
#include "ldpc_spec.hpp"   // exp_glibc

namespace ldpc {

constexpr int kGlobThreads = 1024;   // launch bound; the host launches 256 threads per frame for small codes, 1024 for large ones
inline int glob_threads(int N) { return N >= 16384 ? 1024 : 256; }

struct GlobArgs {
    DecArgs d;          // tables: row_start, edges, col_start, col_edges, col_slot
    char *ws;           // gridDim.x slices of ws_stride bytes
    size_t ws_stride;
    int ne;             // circulants
    // bp_global_kernel only: the frame chain (see SpecArgs in ldpc_spec.hpp; same layouts)
    const uint32_t *stale;    // [B][rh * ceil(M/64)] u64 words, bit = lane: syndrome the previous frame left behind, or null
    uint32_t *synd_out;       // same layout: what this frame leaves behind, or null
    const int *frame_idx;     // [slots] frame decoded by each workgroup slot, or null = the slot index
    long long slots;          // workgroup slots of this launch (B, or the number of frames of a re-decode pass)
    int asp_cw2;              // asp_global_kernel: every block column holds exactly two circulants -> upstream's own branch (decoders.cpp:2431-2480)
    const double *ims_coef;   // ims_global_kernel: [B] sqrt(N / sum y^2) per frame (ims_coef_kernel: the sum is sequential, its rounding is part of the result)
};

// slice layout (all offsets 16-byte aligned)
struct GlobView {
    double *soft, *aux, *m1, *m2, *tmp;   // aux: [N]; tmp: `edge_arrays` arrays of ne * M doubles, one after the other
    int32_t *pos;
    uint8_t *par, *sgn;
};
__host__ __device__ inline size_t glob_align(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ inline size_t glob_ws_bytes(int N, int R, int ne, int M, int edge_arrays) {
    return 2 * glob_align(sizeof(double) * (size_t)N) + 2 * glob_align(sizeof(double) * (size_t)R) + glob_align(sizeof(int32_t) * (size_t)R) +
           glob_align((size_t)R) + glob_align((size_t)ne * M) + (size_t)edge_arrays * glob_align(sizeof(double) * (size_t)ne * M);
}
__device__ inline GlobView glob_view(char *p, int N, int R, int ne, int M, int edge_arrays) {
    GlobView v;
    v.soft = reinterpret_cast<double *>(p); p += glob_align(sizeof(double) * (size_t)N);
    v.aux = reinterpret_cast<double *>(p); p += glob_align(sizeof(double) * (size_t)N);
    v.m1 = reinterpret_cast<double *>(p); p += glob_align(sizeof(double) * (size_t)R);
    v.m2 = reinterpret_cast<double *>(p); p += glob_align(sizeof(double) * (size_t)R);
    v.pos = reinterpret_cast<int32_t *>(p); p += glob_align(sizeof(int32_t) * (size_t)R);
    v.par = reinterpret_cast<uint8_t *>(p); p += glob_align((size_t)R);
    v.sgn = reinterpret_cast<uint8_t *>(p); p += glob_align((size_t)ne * M);
    v.tmp = edge_arrays ? reinterpret_cast<double *>(p) : nullptr;
    return v;
}

template <int KIND = 0>   // hard decision: 0 soft < 0 (LLR decoders), 1 soft > 0.5 (probability domain, :2734), 2 soft < 1.0 (likelihood ratios, :2172)
__device__ inline void glob_outputs(const DecArgs &a, const GlobView &w, long long fr, int res) {
    const int N = a.N;
    if (threadIdx.x == 0 && a.iters) a.iters[fr] = res;
    if (a.hard) {
        for (int wd = threadIdx.x; wd < a.hard_words; wd += (int)blockDim.x) {
            uint32_t bits = 0;
            for (int b = 0; b < 32; ++b) {
                const int v = 32 * wd + b;
                if (v < N) bits |= (uint32_t)(KIND == 1 ? w.soft[v] > 0.5 : KIND == 2 ? w.soft[v] < 1.0 : w.soft[v] < 0) << b;      // decword[k] = soft[k] < 0  (:4683, :5418)
            }
            a.hard[fr * a.hard_words + wd] = bits;
        }
    }
    if (a.soft_out)
        for (int v = threadIdx.x; v < N; v += (int)blockDim.x) a.soft_out[fr * N + v] = w.soft[v];
}

__global__ void __launch_bounds__(kGlobThreads) ms_global_kernel(const GlobArgs g) {
    const DecArgs &a = g.d;
    const int M = a.M, N = a.N, R = a.rh * M, ne = g.ne;
    const double alpha = a.alpha;
    const GlobView w = glob_view(g.ws + (size_t)blockIdx.x * g.ws_stride, N, R, ne, M, 0);
    for (long long fr = blockIdx.x; fr < a.B; fr += gridDim.x) {
        const double *y = a.llr + fr * N;
        for (int c = threadIdx.x; c < R; c += (int)blockDim.x) { w.m1[c] = 0.0; w.m2[c] = 0.0; w.pos[c] = 0; w.par[c] = 0; }   // :4579-4596
        for (size_t i = threadIdx.x; i < (size_t)ne * M; i += (int)blockDim.x) w.sgn[i] = 0;
        __syncthreads();
        int res = -a.maxiter;
        for (int iter = 0; iter < a.maxiter; ++iter) {
            // ---- STATE1 + STATE2 from the variable side (:4633-4685)
            for (int v = threadIdx.x; v < N; v += (int)blockDim.x) {
                const int k = v / M, i = v - k * M;
                double acc = 0.0;                                                 // memset(soft, 0) :4630
                for (int q = a.col_start[k]; q < a.col_start[k + 1]; ++q) {       // block rows ascending
                    const uint32_t d = a.col_edges[q];
                    const int j = (int)(d >> 16), c = (int)(d & 0xffffu), e = (int)a.col_slot[q];
                    int n = i - c;                                                // check n of row j sees variable (k, (n + c) mod M)
                    if (n < 0) n += M;
                    const int chk = j * M + n;
                    const double tmp = w.pos[chk] == e - a.row_start[j] ? w.m2[chk] : w.m1[chk];
                    const double cv = (w.sgn[(size_t)e * M + n] ^ w.par[chk]) ? -tmp : tmp;
                    acc = acc + cv;                                               // :4660
                }
                w.soft[v] = y[v] + acc * alpha;                                   // :4674 / :4682, two roundings
            }
            __syncthreads();
            // ---- STATE3 from the check side (:4690-4755)
            int fail = 0;
            for (int chk = threadIdx.x; chk < R; chk += (int)blockDim.x) {
                const int j = chk / M, n = chk - j * M;
                const int e0 = a.row_start[j], e1 = a.row_start[j + 1];
                const int po = w.pos[chk];
                const uint8_t pa = w.par[chk];
                const double o1 = w.m1[chk], o2 = w.m2[chk];
                double nm1 = kMaxVal, nm2 = kMaxVal;
                int np = 0, synd = 0;
                uint8_t npar = 0;
                for (int e = e0; e < e1; ++e) {
                    const uint32_t d = a.edges[e];
                    const int k = (int)(d >> 16), c = (int)(d & 0xffffu);
                    int i = n + c;
                    if (i >= M) i -= M;
                    const double r = w.soft[k * M + i];
                    synd ^= (int)(r < 0);                                         // :4712
                    double val = (po == e - e0 ? o2 : o1) * alpha;               // :4720
                    double t = (w.sgn[(size_t)e * M + n] ^ pa) ? -val : val;
                    t = r - t;
                    const uint8_t sign = t < 0;
                    w.sgn[(size_t)e * M + n] = sign;
                    npar ^= sign;
                    val = t < 0.0 ? -t : t;
                    val = (val > kMaxVal) ? kMaxVal : val;                        // :4730
                    if (val < nm1) { np = e - e0; nm2 = nm1; nm1 = val; }
                    else if (val < nm2) nm2 = val;
                }
                w.m1[chk] = nm1; w.m2[chk] = nm2; w.pos[chk] = np; w.par[chk] = npar;
                fail |= synd;
            }
            if (!__syncthreads_or(fail)) { res = iter + 1; break; }               // :4757-4766 (the barrier also fences the next STATE1)
        }
        glob_outputs(a, w, fr, res);
        __syncthreads();                                                          // the slice is reused by this workgroup's next frame
    }
}

__global__ void __launch_bounds__(kGlobThreads) lms_global_kernel(const GlobArgs g) {
    const DecArgs &a = g.d;
    const int M = a.M, N = a.N, R = a.rh * M, ne = g.ne;
    const GlobView w = glob_view(g.ws + (size_t)blockIdx.x * g.ws_stride, N, R, ne, M, 1);
    auto syndrome = [&]() -> int {                                                // check_syndrome, decoders.cpp:793-814
        int fail = 0;
        for (int chk = threadIdx.x; chk < R; chk += (int)blockDim.x) {
            const int j = chk / M, n = chk - j * M;
            int synd = 0;
            for (int e = a.row_start[j]; e < a.row_start[j + 1]; ++e) {
                const uint32_t d = a.edges[e];
                int i = n + (int)(d & 0xffffu);
                if (i >= M) i -= M;
                synd ^= (int)(w.soft[(int)(d >> 16) * M + i] < 0);
            }
            fail |= synd;
        }
        return __syncthreads_or(fail);
    };
    for (long long fr = blockIdx.x; fr < a.B; fr += gridDim.x) {
        const double *y = a.llr + fr * N;
        for (int v = threadIdx.x; v < N; v += (int)blockDim.x) w.soft[v] = y[v];    // :5088
        for (int c = threadIdx.x; c < R; c += (int)blockDim.x) { w.m1[c] = 0.0; w.m2[c] = 0.0; w.pos[c] = 0; w.par[c] = 0; }
        for (size_t i = threadIdx.x; i < (size_t)ne * M; i += (int)blockDim.x) w.sgn[i] = 0;
        __syncthreads();
        int parity = syndrome();                                                   // :5111-5115
        int iter = 0;
        for (; iter < a.maxiter; ++iter) {
            if (parity == 0) break;                                                // :5119
            for (int j = 0; j < a.rh; ++j) {                                       // layers in sequence
                const int e0 = a.row_start[j], e1 = a.row_start[j + 1];
                for (int n = threadIdx.x; n < M; n += (int)blockDim.x) {
                    const int chk = j * M + n;
                    const int po = w.pos[chk];
                    const uint8_t pa = w.par[chk];
                    const double o1 = w.m1[chk], o2 = w.m2[chk];
                    double nm1 = kMaxVal, nm2 = kMaxVal;                           // :5133-5134
                    int np = 0;
                    uint8_t npar = 0;
                    for (int e = e0; e < e1; ++e) {                                // :5141-5177
                        const uint32_t d = a.edges[e];
                        int i = n + (int)(d & 0xffffu);
                        if (i >= M) i -= M;
                        const double prev_abs = po == e - e0 ? o2 : o1;
                        const double prev_val = (w.sgn[(size_t)e * M + n] ^ pa) ? -prev_abs : prev_abs;
                        const double t = w.soft[(int)(d >> 16) * M + i] - prev_val;
                        const uint8_t sign = t < 0;
                        double mag = t < 0.0 ? -t : t;
                        mag -= 0.4;                                                 // beta :5163
                        mag = mag < 0 ? 0 : mag;
                        w.tmp[(size_t)e * M + n] = t;
                        w.sgn[(size_t)e * M + n] = sign;
                        npar ^= sign;                                              // process_check_node :5009-5027
                        if (mag < nm1) { np = e - e0; nm2 = nm1; nm1 = mag; }
                        else if (mag < nm2) nm2 = mag;
                    }
                    w.m1[chk] = nm1; w.m2[chk] = nm2; w.pos[chk] = np; w.par[chk] = npar;
                    for (int e = e0; e < e1; ++e) {                                // :5182-5206
                        const uint32_t d = a.edges[e];
                        int i = n + (int)(d & 0xffffu);
                        if (i >= M) i -= M;
                        const double c_abs = np == e - e0 ? nm2 : nm1;
                        const double c_val = (w.sgn[(size_t)e * M + n] ^ npar) ? -c_abs : c_abs;
                        w.soft[(int)(d >> 16) * M + i] = w.tmp[(size_t)e * M + n] + c_val;
                    }
                }
                __syncthreads();
            }
            parity = syndrome();                                                   // :5281-5288
            if (parity == 0) break;
        }
        glob_outputs(a, w, fr, parity ? -iter : iter + 1);                         // :5424
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kGlobThreads) tasp_global_kernel(const GlobArgs g) {
    const DecArgs &a = g.d;
    const int M = a.M, N = a.N, R = a.rh * M, ne = g.ne;
    const double T = 0.0001, TT = 0;                                                // :2597-2598
    const GlobView w = glob_view(g.ws + (size_t)blockIdx.x * g.ws_stride, N, R, ne, M, 4);
    const size_t EM = glob_align(sizeof(double) * (size_t)ne * M) / sizeof(double);
    double *const Z = w.tmp, *const Y = w.tmp + EM, *const SF = w.tmp + 2 * EM, *const SB = w.tmp + 3 * EM;   // [e * M + k]
    auto syndrome = [&]() -> int {                                                  // check_syndrome_thr :2274-2306, thr 0.5
        int fail = 0;
        for (int chk = threadIdx.x; chk < R; chk += (int)blockDim.x) {
            const int j = chk / M, n = chk - j * M;
            int synd = 0;
            for (int e = a.row_start[j]; e < a.row_start[j + 1]; ++e) {
                const uint32_t d = a.edges[e];
                int i = n + (int)(d & 0xffffu);
                if (i >= M) i -= M;
                synd ^= (int)(w.soft[(int)(d >> 16) * M + i] > 0.5);
            }
            fail |= synd;
        }
        return __syncthreads_or(fail);
    };
    for (long long fr = blockIdx.x; fr < a.B; fr += gridDim.x) {
        for (int v = threadIdx.x; v < N; v += (int)blockDim.x) {                       // :2611-2618
            const double x = a.llr[fr * N + v] * 0.5;
            const double y = x < 20.0 ? (x < -20.0 ? -20.0 : x) : 20.0;             // maxd(mind(x, INPUT_LIMIT), -INPUT_LIMIT)
            const double e0 = ldpc_spec::exp_glibc(y), e1 = ldpc_spec::exp_glibc(-y);
            w.soft[v] = e1 / (e0 + e1);
        }
        for (size_t i = threadIdx.x; i < (size_t)ne * M; i += (int)blockDim.x) Z[i] = 0.5;   // :2637
        __syncthreads();
        int synd = syndrome();                                                      // :2653-2660
        int steps = 0;
        if (synd != 0) {
            while (steps < a.maxiter) {
                for (int j = 0; j < a.rh; ++j) {                                    // layers in sequence (:2668)
                    const int e0 = a.row_start[j], e1 = a.row_start[j + 1], rw = e1 - e0;
                    for (int k = threadIdx.x; k < M; k += (int)blockDim.x) {
                        for (int e = e0; e < e1; ++e) {                             // :2676-2697
                            const uint32_t d = a.edges[e];
                            int i = k + (int)(d & 0xffffu);
                            if (i >= M) i -= M;
                            const double x = w.soft[(int)(d >> 16) * M + i];
                            const double aa = Z[(size_t)e * M + k];
                            double v = x * (1.0 - aa) / (aa + x - 2.0 * aa * x);    // rho = gamma - lambda
                            if (v < TT) v = TT;
                            if (v > 1 - TT) v = 1 - TT;
                            Y[(size_t)e * M + k] = v;
                        }
                        // map_bin(a, cnt, 1) :2191-2228 with P[i] = 1 - 2 * y[i]
                        auto P = [&](int i) { return 1 - 2 * Y[(size_t)(e0 + i) * M + k]; };
                        auto sf = [&](int i) -> double & { return SF[(size_t)(e0 + i) * M + k]; };
                        auto sb = [&](int i) -> double & { return SB[(size_t)(e0 + i) * M + k]; };
                        sf(0) = P(0);
                        for (int i = 1; i < rw - 1; ++i) sf(i) = P(i) * sf(i - 1);
                        sb(rw - 1) = P(rw - 1);
                        for (int i = rw - 2; i > 0; --i) sb(i) = P(i) * sb(i + 1);
                        for (int i = 0; i < rw; ++i) {
                            double q;
                            if (i == 0) q = (1 - sb(1)) / 2;
                            else if (i == rw - 1) q = (1 - sf(rw - 2)) / 2;
                            else q = (1 - sf(i - 1) * sb(i + 1)) / 2;
                            if (q < T) q = T;                                        // :2703-2704
                            if (q > 1.0 - T) q = 1.0 - T;
                            Z[(size_t)(e0 + i) * M + k] = q;
                        }
                        for (int e = e0; e < e1; ++e) {                             // :2707-2720
                            const uint32_t d = a.edges[e];
                            int i = k + (int)(d & 0xffffu);
                            if (i >= M) i -= M;
                            const double y = Y[(size_t)e * M + k], q = Z[(size_t)e * M + k];
                            w.soft[(int)(d >> 16) * M + i] = y * q / (1.0 - y - q + 2 * y * q);   // gamma = rho + lambda
                        }
                    }
                    __syncthreads();
                }
                synd = syndrome();                                                  // :2723 (the value after the last layer)
                steps = steps + 1;
                if (synd == 0) break;
            }
        }
        glob_outputs<1>(a, w, fr, synd ? -steps : steps);                        // :2734-2743 (0 when the input was a codeword)
        __syncthreads();
    }
}

// sum_prod_decod_qc_lm, decoders.cpp:1923-2185 (likelihood-ratio domain), the four phases of ldpc_sumprod.hpp with every array in
// the workspace: ZZ[e][t] per edge and variable position, yd / soft per variable, the check products in m1[].
__global__ void __launch_bounds__(kGlobThreads) sp_global_kernel(const GlobArgs g) {
    const DecArgs &a = g.d;
    const int M = a.M, N = a.N, R = a.rh * M, ne = g.ne;
    const GlobView w = glob_view(g.ws + (size_t)blockIdx.x * g.ws_stride, N, R, ne, M, 1);
    double *const ZZ = w.tmp, *const yd = w.aux, *const S = w.m1;
    auto mind = [](double x, double y) { return x < y ? x : y; };   // decoders.cpp:104-105
    auto maxd = [](double x, double y) { return x < y ? y : x; };
    auto syndrome = [&]() -> int {                                                  // :1964-2002 / :2129-2149
        int fail = 0;
        for (int chk = threadIdx.x; chk < R; chk += (int)blockDim.x) {
            const int j = chk / M, n = chk - j * M;
            int synd = 0;
            for (int e = a.row_start[j]; e < a.row_start[j + 1]; ++e) {
                const uint32_t d = a.edges[e];
                int i = n + (int)(d & 0xffffu);
                if (i >= M) i -= M;
                synd ^= (int)(w.soft[(int)(d >> 16) * M + i] < 1.0);
            }
            fail |= synd;
        }
        return __syncthreads_or(fail);
    };
    for (long long fr = blockIdx.x; fr < a.B; fr += gridDim.x) {
        for (int v = threadIdx.x; v < N; v += (int)blockDim.x) {                       // :1947-1951
            const double yl = maxd(mind(a.llr[fr * N + v], 20.0), -20.0);
            yd[v] = w.soft[v] = ldpc_spec::exp_glibc(yl);
        }
        for (size_t i = threadIdx.x; i < (size_t)ne * M; i += (int)blockDim.x) ZZ[i] = 1.0;   // :1957-1959
        __syncthreads();
        int res = -a.maxiter;
        bool conv = syndrome() == 0;
        if (conv) res = 0;
        for (int iter = 0; !conv && iter < a.maxiter; ++iter) {
            for (int v = threadIdx.x; v < N; v += (int)blockDim.x) {                   // phase A :2017-2060
                const int k = v / M, t = v - k * M;
                const int c0 = a.col_start[k], c1 = a.col_start[k + 1];
                double prefix = yd[v];
                for (int u = c0; u < c1; ++u) {
                    const size_t zi = (size_t)a.col_slot[u] * M + t;
                    const double orig = ZZ[zi];
                    double AA = prefix;
                    for (int x = u + 1; x < c1; ++x) AA *= ZZ[(size_t)a.col_slot[x] * M + t];
                    ZZ[zi] = (AA - 1) / (AA + 1);
                    prefix *= orig;
                }
            }
            __syncthreads();
            for (int chk = threadIdx.x; chk < R; chk += (int)blockDim.x) {             // phase B :2047-2050
                const int j = chk / M, n = chk - j * M;
                double s = 1.0;
                for (int e = a.row_start[j]; e < a.row_start[j + 1]; ++e) {
                    int i = n + (int)(a.edges[e] & 0xffffu);
                    if (i >= M) i -= M;
                    s *= ZZ[(size_t)e * M + i];
                }
                S[chk] = s;
            }
            __syncthreads();
            for (int v = threadIdx.x; v < N; v += (int)blockDim.x) {                   // phase C :2103-2127
                const int k = v / M, t = v - k * M;
                double soft = yd[v];
                for (int u = a.col_start[k]; u < a.col_start[k + 1]; ++u) {
                    const uint32_t d = a.col_edges[u];
                    const int j = (int)(d >> 16), c = (int)(d & 0xffffu);
                    int nn = t - c;
                    if (nn < 0) nn += M;
                    const size_t zi = (size_t)a.col_slot[u] * M + t;
                    double A = S[j * M + nn] / ZZ[zi];
                    A = (1 + A) / (1 - A);
                    A = maxd(mind(A, 1.9e+8), -5.2e-9);                             // :2120 (the negative lower clamp is upstream's)
                    ZZ[zi] = A;
                    soft *= A;
                }
                w.soft[v] = soft;
            }
            __syncthreads();
            if (syndrome() == 0) { conv = true; res = iter + 1; }                   // :2151-2166
        }
        glob_outputs<2>(a, w, fr, res);
        __syncthreads();
    }
}

// imin_sum_decod_qc_lm, decoders.cpp:5430-5690 (MS_MUL_CORRECTION build: c2v magnitude (min * ialpha) >> 4 in STATE1 and STATE3).
// IMS_DATA is a short upstream; every intermediate fits (|values| <= 2 * max_data < 2^15), so int arithmetic gives the same numbers.
// Workspace: soft / iy as ints in the soft / aux arrays, min1 / min2 as ints in m1 / m2.
__global__ void __launch_bounds__(kGlobThreads) ims_global_kernel(const GlobArgs g) {
    const DecArgs &a = g.d;
    const int M = a.M, N = a.N, R = a.rh * M, ne = g.ne;
    const GlobView w = glob_view(g.ws + (size_t)blockIdx.x * g.ws_stride, N, R, ne, M, 0);
    int *const soft = reinterpret_cast<int *>(w.soft), *const iy = reinterpret_cast<int *>(w.aux);
    int *const m1 = reinterpret_cast<int *>(w.m1), *const m2 = reinterpret_cast<int *>(w.m2);
    const int max_data = (1 << (a.ims_dbits - 1)) - 1;   // :5445
    const int max_quant = (1 << (a.ims_qbits - 1)) - 1;  // :5446
    const int ialpha = (int)(a.alpha * (1 << 4));         // :5458 MS_ALPHA_FPP = 4
    auto sat = [&](int x) { return x > max_data ? max_data : (x < -max_data ? -max_data : x); };   // limit_val :4308
    for (long long fr = blockIdx.x; fr < a.B; fr += gridDim.x) {
        const double coef = g.ims_coef[fr];
        for (int v = threadIdx.x; v < N; v += (int)blockDim.x) {                    // :5472-5500
            double val = a.llr[fr * N + v];
            int sign = 0;
            if (val < 0) { val = -val; sign = 1; }
            val *= coef;
            if (val > a.ims_thr) val = a.ims_thr;
            const int ival = (int)(short)floor(val * max_quant / a.ims_thr + 0.5);
            iy[v] = sign ? -ival : ival;
        }
        for (int c = threadIdx.x; c < R; c += (int)blockDim.x) { m1[c] = 0; m2[c] = 0; w.pos[c] = 0; w.par[c] = 0; }   // :5462-5470
        for (size_t i = threadIdx.x; i < (size_t)ne * M; i += (int)blockDim.x) w.sgn[i] = 0;
        __syncthreads();
        int res = -a.maxiter;
        for (int iter = 0; iter < a.maxiter; ++iter) {
            for (int v = threadIdx.x; v < N; v += (int)blockDim.x) {                // STATE1 + STATE2 from the variable side (:5540-5604)
                const int k = v / M, i = v - k * M;
                int acc = 0;
                for (int q = a.col_start[k]; q < a.col_start[k + 1]; ++q) {         // block rows ascending: saturation after EVERY add (:5568)
                    const uint32_t d = a.col_edges[q];
                    const int j = (int)(d >> 16), c = (int)(d & 0xffffu), e = (int)a.col_slot[q];
                    int n = i - c;
                    if (n < 0) n += M;
                    const int chk = j * M + n;
                    int tmp = w.pos[chk] == e - a.row_start[j] ? m2[chk] : m1[chk];
                    tmp = (tmp * ialpha) >> 4;                                       // :5554
                    const int cv = (w.sgn[(size_t)e * M + n] ^ w.par[chk]) ? -tmp : tmp;
                    acc = sat(acc + cv);
                }
                soft[v] = sat(iy[v] + acc);                                         // :5590-5601
            }
            __syncthreads();
            int fail = 0;
            for (int chk = threadIdx.x; chk < R; chk += (int)blockDim.x) {          // STATE3 (:5610-5678)
                const int j = chk / M, n = chk - j * M;
                const int e0 = a.row_start[j], e1 = a.row_start[j + 1];
                const int po = w.pos[chk], o1 = m1[chk], o2 = m2[chk];
                const uint8_t pa = w.par[chk];
                int nm1 = max_data, nm2 = max_data, np = 0, synd = 0;
                uint8_t npar = 0;
                for (int e = e0; e < e1; ++e) {
                    const uint32_t d = a.edges[e];
                    int i = n + (int)(d & 0xffffu);
                    if (i >= M) i -= M;
                    const int r = soft[(int)(d >> 16) * M + i];
                    synd ^= (int)(r < 0);
                    const int val = ((po == e - e0 ? o2 : o1) * ialpha) >> 4;       // :5640
                    const int t = (w.sgn[(size_t)e * M + n] ^ pa) ? -val : val;
                    const int msg = r - t;
                    const uint8_t sign = msg < 0;
                    w.sgn[(size_t)e * M + n] = sign;
                    npar ^= sign;
                    int v = msg < 0 ? -msg : msg;
                    v = v > max_data ? max_data : v;
                    if (v < nm1) { np = e - e0; nm2 = nm1; nm1 = v; }
                    else if (v < nm2) nm2 = v;
                }
                m1[chk] = nm1; m2[chk] = nm2; w.pos[chk] = np; w.par[chk] = npar;
                fail |= synd;
            }
            if (!__syncthreads_or(fail)) { res = iter + 1; break; }                 // :5684-5689
        }
        if (threadIdx.x == 0 && a.iters) a.iters[fr] = res;
        if (a.hard) {
            for (int wd = threadIdx.x; wd < a.hard_words; wd += (int)blockDim.x) {
                uint32_t bits = 0;
                for (int b = 0; b < 32; ++b) if (32 * wd + b < N) bits |= (uint32_t)(soft[32 * wd + b] < 0) << b;
                a.hard[fr * a.hard_words + wd] = bits;
            }
        }
        if (a.soft_out)
            for (int v = threadIdx.x; v < N; v += (int)blockDim.x) a.soft_out[fr * N + v] = (double)soft[v];
        __syncthreads();
    }
}

// sum_prod_gf2_decod_qc_lm, decoders.cpp:2324-2581 (probability domain, flooding): the general branch :2482-2556 and -- GlobArgs::asp_cw2
// -- the branch upstream takes for codes whose block columns ALL hold exactly two circulants (:2431-2480; only this tier runs it).
// Workspace: state[e][n] per circulant and check (upstream's state[slot][row*m + n]) + two scratch arrays for map_bin's forward /
// backward products; channel P(bit = 1) in aux, a-posteriori values in soft.
__global__ void __launch_bounds__(kGlobThreads) asp_global_kernel(const GlobArgs g) {
    const DecArgs &a = g.d;
    const int M = a.M, N = a.N, R = a.rh * M, ne = g.ne;
    const GlobView w = glob_view(g.ws + (size_t)blockIdx.x * g.ws_stride, N, R, ne, M, 3);
    const size_t EM = glob_align(sizeof(double) * (size_t)ne * M) / sizeof(double);
    double *const ST = w.tmp, *const SF = w.tmp + EM, *const SB = w.tmp + 2 * EM, *const p1ch = w.aux;
    auto mind = [](double x, double y) { return x < y ? x : y; };
    auto maxd = [](double x, double y) { return x < y ? y : x; };
    auto syndrome = [&]() -> int {                                                  // check_syndrome_thr :2274-2306, thr 0.5
        int fail = 0;
        for (int chk = threadIdx.x; chk < R; chk += (int)blockDim.x) {
            const int j = chk / M, n = chk - j * M;
            int synd = 0;
            for (int e = a.row_start[j]; e < a.row_start[j + 1]; ++e) {
                const uint32_t d = a.edges[e];
                int i = n + (int)(d & 0xffffu);
                if (i >= M) i -= M;
                synd ^= (int)(w.soft[(int)(d >> 16) * M + i] > 0.5);
            }
            fail |= synd;
        }
        return __syncthreads_or(fail);
    };
    for (long long fr = blockIdx.x; fr < a.B; fr += gridDim.x) {
        for (int v = threadIdx.x; v < N; v += (int)blockDim.x) {                    // :2351-2379
            const int k = v / M, t = v - k * M;
            const double x = a.llr[fr * N + v] * 0.5;
            const double y = maxd(mind(x, 20.0), -20.0);
            const double e0 = ldpc_spec::exp_glibc(y), e1 = ldpc_spec::exp_glibc(-y);
            const double p = e1 / (e0 + e1);
            p1ch[v] = w.soft[v] = p;
            for (int q = a.col_start[k]; q < a.col_start[k + 1]; ++q) {             // state <- rotated channel probability
                int nn = t - (int)(a.col_edges[q] & 0xffffu);
                if (nn < 0) nn += M;
                ST[(size_t)a.col_slot[q] * M + nn] = p;
            }
        }
        __syncthreads();
        int synd = syndrome();                                                      // :2393-2399
        int steps = 0;
        while (synd != 0 && steps < a.maxiter) {
            for (int chk = threadIdx.x; chk < R; chk += (int)blockDim.x) {          // check nodes: map_bin(&state[0][i*m+k], rw, r) :2191-2228
                const int j = chk / M, n = chk - j * M;
                const int e0 = a.row_start[j], rw = a.row_start[j + 1] - e0;
                auto st = [&](int i) -> double & { return ST[(size_t)(e0 + i) * M + n]; };
                auto sf = [&](int i) -> double & { return SF[(size_t)(e0 + i) * M + n]; };
                auto sb = [&](int i) -> double & { return SB[(size_t)(e0 + i) * M + n]; };
                auto P = [&](int i) { return 1 - 2 * st(i); };
                sf(0) = P(0);
                for (int i = 1; i < rw - 1; ++i) sf(i) = P(i) * sf(i - 1);
                sb(rw - 1) = P(rw - 1);
                for (int i = rw - 2; i > 0; --i) sb(i) = P(i) * sb(i + 1);
                st(0) = (1 - sb(1)) / 2;
                for (int i = 1; i < rw - 1; ++i) st(i) = (1 - sf(i - 1) * sb(i + 1)) / 2;
                st(rw - 1) = (1 - sf(rw - 2)) / 2;
            }
            __syncthreads();
            if (g.asp_cw2) {
                // every block column has exactly two circulants: upstream's own branch (:2431-2480) -- the messages are formed from the
                // channel value and the OTHER edge directly, and nothing is clamped
                for (int v = threadIdx.x; v < N; v += (int)blockDim.x) {
                    const int k = v / M, t = v - k * M;
                    const int q0 = a.col_start[k], q1 = q0 + 1;                     // hci[i][0] < hci[i][1]: rows ascending
                    int n0 = t - (int)(a.col_edges[q0] & 0xffffu), n1 = t - (int)(a.col_edges[q1] & 0xffffu);
                    if (n0 < 0) n0 += M;
                    if (n1 < 0) n1 += M;
                    const size_t z0 = (size_t)a.col_slot[q0] * M + n0, z1 = (size_t)a.col_slot[q1] * M + n1;
                    const double d0 = ST[z0], d1 = ST[z1];                          // data0[k], data1[k] :2449-2450
                    double p1 = p1ch[v];
                    double q10 = p1, q11 = p1, p0 = 1.0 - p1, q00 = 1.0 - p1, q01 = 1.0 - p1;   // :2454-2459
                    q10 = q10 * d1;                                                 // :2461-2466
                    q00 = q00 * (1 - d1);
                    q11 = q11 * d0;
                    q01 = q01 * (1 - d0);
                    p1 = q10 * d0;
                    p0 = q00 * (1 - d0);
                    w.soft[v] = p1 / (p0 + p1);                                     // :2469
                    ST[z0] = q10 / (q10 + q00);                                     // :2471
                    ST[z1] = q11 / (q11 + q01);                                     // :2472
                }
            } else
            for (int v = threadIdx.x; v < N; v += (int)blockDim.x) {                // symbol nodes + local data update :2482-2556
                const int k = v / M, t = v - k * M;
                double P1 = p1ch[v], P0 = 1 - p1ch[v];
                for (int q = a.col_start[k]; q < a.col_start[k + 1]; ++q) {         // rows ascending
                    int nn = t - (int)(a.col_edges[q] & 0xffffu);
                    if (nn < 0) nn += M;
                    const double d = ST[(size_t)a.col_slot[q] * M + nn];
                    P1 *= d;
                    P0 *= 1 - d;
                }
                const double so = P1 / (P0 + P1);
                w.soft[v] = so;
                for (int q = a.col_start[k]; q < a.col_start[k + 1]; ++q) {
                    int nn = t - (int)(a.col_edges[q] & 0xffffu);
                    if (nn < 0) nn += M;
                    const size_t zi = (size_t)a.col_slot[q] * M + nn;
                    const double sos = ST[zi];
                    const double p1 = so / sos;
                    const double p0 = (1 - so) / (1 - sos);
                    const double dd = p1 / (p1 + p0);
                    ST[zi] = maxd(mind(dd, 1.0 - 0.000001), 0.000001);              // SP_DEC_MAX_VAL / SP_DEC_MIN_VAL :96-97
                }
            }
            __syncthreads();
            synd = syndrome();                                                      // :2566
            steps = steps + 1;
        }
        glob_outputs<1>(a, w, fr, synd ? -steps : steps);                           // 0: codeword at the input; converged after `steps`; else -steps
        __syncthreads();
    }
}

// bp_decod_qc_lm, decoders.cpp:1708-1920 (Gallager BP in the log domain), the phases of bp_body (ldpc_spec.hpp) with every array in
// the workspace: ZZ[e][t] per edge and variable position (tmp), BB in sgn, the check sums s in m1 and their signs in par, yd in aux,
// the a-posteriori LLRs in soft.  exp / log are glibc's algorithms (tables read from global memory here).  The frame chain
// (stale syndrome in, syndrome left behind out, re-decode passes) works exactly as in the resident kernel.
__global__ void __launch_bounds__(kGlobThreads) bp_global_kernel(const GlobArgs g) {
    const DecArgs &a = g.d;
    const int M = a.M, N = a.N, R = a.rh * M, ne = g.ne, CH = (M + 63) / 64;
    const GlobView w = glob_view(g.ws + (size_t)blockIdx.x * g.ws_stride, N, R, ne, M, 1);
    double *const ZZ = w.tmp, *const yd = w.aux, *const S = w.m1;
    uint8_t *const BB = w.sgn, *const bs = w.par;
    uint8_t *const left = reinterpret_cast<uint8_t *>(w.pos);                       // [R] syndrome bits as last computed (upstream's st->syndr)
    auto mind = [](double x, double y) { return x < y ? x : y; };
    auto maxd = [](double x, double y) { return x < y ? y : x; };
    for (long long slot = blockIdx.x; slot < g.slots; slot += gridDim.x) {
        const long long fr = g.frame_idx ? g.frame_idx[slot] : slot;
        auto syndrome = [&](bool with_stale) -> int {                               // :1742-1766 / :1869-1893
            int fail = 0;
            for (int chk = threadIdx.x; chk < R; chk += (int)blockDim.x) {
                const int j = chk / M, n = chk - j * M;
                int sy = 0;
                if (with_stale && g.stale) sy = (int)((reinterpret_cast<const unsigned long long *>(g.stale)[fr * (a.rh * CH) + j * CH + (n >> 6)] >> (n & 63)) & 1ull);
                for (int e = a.row_start[j]; e < a.row_start[j + 1]; ++e) {
                    const uint32_t d = a.edges[e];
                    int i = n + (int)(d & 0xffffu);
                    if (i >= M) i -= M;
                    sy ^= (int)(w.soft[(int)(d >> 16) * M + i] < 0);
                }
                left[chk] = (uint8_t)sy;
                fail |= sy;
            }
            return __syncthreads_or(fail);
        };
        for (int v = threadIdx.x; v < N; v += (int)blockDim.x) {
            const double y = maxd(mind(a.llr[fr * N + v], 20.0), -20.0);            // :1738 INPUT_LIMIT
            yd[v] = w.soft[v] = y;
        }
        for (size_t i = threadIdx.x; i < (size_t)ne * M; i += (int)blockDim.x) ZZ[i] = 0.0;   // :1731-1733
        __syncthreads();
        int fail = syndrome(true);
        // re-decode pass: nothing can change unless the frame was a codeword at its input (see bp_body)
        if (g.frame_idx && fail && a.iters && a.iters[fr] != 0) { __syncthreads(); continue; }
        int iter = 0;
        while (fail && iter < a.maxiter) {
            for (int v = threadIdx.x; v < N; v += (int)blockDim.x) {                // A: variable-node activation :1803-1812
                const int k = v / M, t = v - k * M;
                const double so = w.soft[v];
                for (int q = a.col_start[k]; q < a.col_start[k + 1]; ++q) {
                    const size_t zi = (size_t)a.col_slot[q] * M + t;
                    const double A = ldpc_spec::exp_glibc_wide(so - ZZ[zi], ldpc_spec::kExpTab);
                    ZZ[zi] = ldpc_spec::log_glibc(fabs((A - 1) / (A + 1)));
                    BB[zi] = A < 1;
                }
            }
            __syncthreads();
            for (int chk = threadIdx.x; chk < R; chk += (int)blockDim.x) {          // A': check sums :1815-1824, columns ascending
                const int j = chk / M, n = chk - j * M;
                double s = 0.0;
                unsigned b = 0;
                for (int e = a.row_start[j]; e < a.row_start[j + 1]; ++e) {
                    int i = n + (int)(a.edges[e] & 0xffffu);
                    if (i >= M) i -= M;
                    s += ZZ[(size_t)e * M + i];
                    b ^= BB[(size_t)e * M + i];
                }
                S[chk] = s;
                bs[chk] = (uint8_t)b;
            }
            __syncthreads();
            for (int v = threadIdx.x; v < N; v += (int)blockDim.x) {                // B: check-node activation seen from the variable :1834-1866
                const int k = v / M, t = v - k * M;
                double soft = yd[v];
                for (int q = a.col_start[k]; q < a.col_start[k + 1]; ++q) {         // rows ascending
                    const uint32_t d = a.col_edges[q];
                    const int j = (int)(d >> 16);
                    int nn = t - (int)(d & 0xffffu);
                    if (nn < 0) nn += M;
                    const size_t zi = (size_t)a.col_slot[q] * M + t;
                    double A = ldpc_spec::exp_glibc_wide(S[j * M + nn] - ZZ[zi], ldpc_spec::kExpTab);
                    const int b = bs[j * M + nn] ^ BB[zi];
                    A = (double)(1 - 2 * b) * ldpc_spec::log_glibc((1 + A) / (1 - A));
                    const double zn = maxd(mind(A, 19.07), -19.07);
                    ZZ[zi] = zn;
                    soft += zn;
                }
                w.soft[v] = soft;
            }
            __syncthreads();
            fail = syndrome(false);
            iter = iter + 1;
        }
        if (g.synd_out) {
            for (int u = threadIdx.x; u < a.rh * CH; u += (int)blockDim.x) {        // one u64 per block row and 64-lane chunk
                const int j = u / CH, ch = u - j * CH;
                unsigned long long bits = 0;
                for (int l = 0; l < 64 && ch * 64 + l < M; ++l) bits |= (unsigned long long)left[j * M + ch * 64 + l] << l;
                reinterpret_cast<unsigned long long *>(g.synd_out)[fr * (a.rh * CH) + u] = bits;
            }
        }
        glob_outputs<0>(a, w, fr, fail ? -iter : iter);
        __syncthreads();
    }
}

}  // namespace ldpc
