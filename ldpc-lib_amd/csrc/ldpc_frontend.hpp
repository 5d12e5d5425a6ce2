// ldpc_frontend.hpp -- channel front end and error accounting kernels around the decoders (gfx950).
//
//   channel_llr_kernel<H>  the whole transmit / receive chain of one frame, bp_simulation.cpp:566-577,596-710:
//                          codeword -> direct interleaver -> mapper (H = 0: BPSK / QAM4 formula :603,:610; H = 2,3,4: GrayPAM
//                          16/64/256-QAM, QAM_modulator.cpp:129-194) -> AWGN -> soft demapper (QAM_demodulator.cpp:203-561,
//                          negated :627-628) -> inverse interleaver (:684) -> puncturing (:697-710)
//   qam_modulate_kernel    QAM_modulator.cpp:142-194 QAM_modulator(), function level
//   qam_demod_kernel       QAM_demodulator.cpp:99-566 Demodulate(), Q in {4,16,64,256}
//   count_errors_kernel    bp_simulation.cpp:731-759,805-810 (against the transmitted codeword)
//
// These are streaming, HBM-bound byte/word kernels: one element (pair) per lane, coalesced 8/16-byte accesses,
// grid-stride loops; no LDS.  Noise comes from a counter-based Philox4x32-10 generator keyed by
// (seed, global frame index, element index), so a frame's noise does not depend on how frames are batched or
// sharded over GPUs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ldpc {

// ---- Philox4x32-10 (Salmon et al., SC'11): public algorithm, restated here; the CPU twin is in tests/ ----
struct Philox4 { uint32_t x[4]; };

__host__ __device__ inline void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
    uint32_t c[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Philox4 o;
    o.x[0] = c[0]; o.x[1] = c[1]; o.x[2] = c[2]; o.x[3] = c[3];
    return o;
}

// two 53-bit uniforms in (0,1) from one Philox block
__host__ __device__ inline void philox_uniform2(const Philox4 &p, double &u1, double &u2) {
    const uint64_t a = (((uint64_t)p.x[0] << 32) | p.x[1]) >> 11;
    const uint64_t b = (((uint64_t)p.x[2] << 32) | p.x[3]) >> 11;
    u1 = ((double)a + 0.5) * (1.0 / 9007199254740992.0);
    u2 = ((double)b + 0.5) * (1.0 / 9007199254740992.0);
}

// Box-Muller pair for (frame, pair index) of stream `tag`
__device__ __forceinline__ void gauss_pair(uint64_t seed, uint64_t frame, uint32_t pair, uint32_t tag, double &g0,
                                           double &g1) {
    const Philox4 p = philox4x32_10((uint32_t)frame, (uint32_t)(frame >> 32), pair, tag, (uint32_t)seed,
                                    (uint32_t)(seed >> 32));
    double u1, u2;
    philox_uniform2(p, u1, u2);
    const double rad = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    g0 = rad * c;
    g1 = rad * s;
}

// ---- soft demapper --------------------------------------------------------------------------------------
// One PAM rail of a 16-QAM symbol: QAM_demodulator.cpp:171-275.  lattice = {-3,-1,1,3} (levels ordered 00 01 11 10).
__device__ __forceinline__ double llr_or_p(double p0, double p1, double T, int out_type) {
    if (p0 == 0.0) return out_type == 0 ? T : 1.0;
    if (p1 == 0.0) return out_type == 0 ? -T : 0.0;
    return out_type == 0 ? log(p1 / p0) : p1;
}

// One PAM rail of H bits (SQ = 2^H levels 2i - (SQ-1), anti-Gray labelled): posterior level probabilities with the
// exp cut-off T, then per bit the two partition sums in the reference's association order
// (QAM_demodulator.cpp:203-275 H = 2, :276-393 H = 3, :395-561 H = 4).
template <int H>
__device__ __forceinline__ void demod_rail(double x, double N0, double T, int out_type, double (&b)[H]) {
    constexpr int SQ = 1 << H;
    double P[SQ], sum = 0;
#pragma unroll
    for (int i = 0; i < SQ; ++i) {
        double t = x - (double)(2 * i - (SQ - 1));
        t *= t;
        t /= N0;
        P[i] = (t < T) ? exp(-t) : 0.0;
        sum += P[i];
    }
#pragma unroll
    for (int i = 0; i < SQ; ++i) P[i] /= sum;
    if constexpr (H == 2) {
        b[0] = llr_or_p(P[0] + P[1], P[2] + P[3], T, out_type);
        b[1] = llr_or_p(P[0] + P[3], P[1] + P[2], T, out_type);
    } else if constexpr (H == 3) {
        const double p12 = P[0] + P[1], p34 = P[2] + P[3], p56 = P[4] + P[5], p78 = P[6] + P[7];
        b[0] = llr_or_p(p12 + p34, p56 + p78, T, out_type);
        b[1] = llr_or_p(p12 + p78, p34 + p56, T, out_type);
        b[2] = llr_or_p(P[0] + P[3] + P[4] + P[7], P[1] + P[2] + P[5] + P[6], T, out_type);
    } else {
        const double p12 = P[0] + P[1], p34 = P[2] + P[3], p56 = P[4] + P[5], p78 = P[6] + P[7];
        const double p9A = P[8] + P[9], pBC = P[10] + P[11], pDE = P[12] + P[13], pFG = P[14] + P[15];
        const double p1234 = p12 + p34, p5678 = p56 + p78, p9ABC = p9A + pBC, pDEFG = pDE + pFG;
        b[0] = llr_or_p(p1234 + p5678, p9ABC + pDEFG, T, out_type);
        b[1] = llr_or_p(p1234 + pDEFG, p5678 + p9ABC, T, out_type);
        b[2] = llr_or_p(p12 + p78 + p9A + pFG, p34 + p56 + pBC + pDE, T, out_type);
        b[3] = llr_or_p(P[0] + P[3] + P[4] + P[7] + P[8] + P[11] + P[12] + P[15],
                        P[1] + P[2] + P[5] + P[6] + P[9] + P[10] + P[13] + P[14], T, out_type);
    }
}
__device__ __forceinline__ void demod_rail16(double x, double N0, double T, int out_type, double &b0, double &b1) {
    double b[2];
    demod_rail<2>(x, N0, T, out_type, b);
    b0 = b[0]; b1 = b[1];
}

struct DemodArgs {
    const double *x;  // [ns][2]
    double *out;      // [ns][m]
    long long ns;
    int Q, out_type;
    double T, sigma;
};

__global__ void __launch_bounds__(256) qam_demod_kernel(const DemodArgs a) {
    const double N0 = 2.0 * a.sigma * a.sigma;  // QAM_demodulator.cpp:142
    for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s < a.ns;
         s += (long long)gridDim.x * blockDim.x) {
        const double2 xy = *reinterpret_cast<const double2 *>(a.x + 2 * s);
        if (a.Q == 4) {  // :113-139 (out_type 1 normalises over the whole block: done by the caller's second pass)
            const double sigma2 = a.sigma * a.sigma;
            a.out[2 * s] = 2.0 * xy.x / sigma2;
            a.out[2 * s + 1] = 2.0 * xy.y / sigma2;
        } else if (a.Q == 16) {
            double b0, b1, b2, b3;
            demod_rail16(xy.x, N0, a.T, a.out_type, b0, b1);
            demod_rail16(xy.y, N0, a.T, a.out_type, b2, b3);
            *reinterpret_cast<double2 *>(a.out + 4 * s) = make_double2(b0, b1);
            *reinterpret_cast<double2 *>(a.out + 4 * s + 2) = make_double2(b2, b3);
        } else if (a.Q == 64) {
            double bi[3], bq[3];
            demod_rail<3>(xy.x, N0, a.T, a.out_type, bi);
            demod_rail<3>(xy.y, N0, a.T, a.out_type, bq);
            double *o = a.out + 6 * s;
            *reinterpret_cast<double2 *>(o) = make_double2(bi[0], bi[1]);
            *reinterpret_cast<double2 *>(o + 2) = make_double2(bi[2], bq[0]);
            *reinterpret_cast<double2 *>(o + 4) = make_double2(bq[1], bq[2]);
        } else {
            double bi[4], bq[4];
            demod_rail<4>(xy.x, N0, a.T, a.out_type, bi);
            demod_rail<4>(xy.y, N0, a.T, a.out_type, bq);
            double *o = a.out + 8 * s;
            *reinterpret_cast<double2 *>(o) = make_double2(bi[0], bi[1]);
            *reinterpret_cast<double2 *>(o + 2) = make_double2(bi[2], bi[3]);
            *reinterpret_cast<double2 *>(o + 4) = make_double2(bq[0], bq[1]);
            *reinterpret_cast<double2 *>(o + 6) = make_double2(bq[2], bq[3]);
        }
    }
}

// anti-Gray PAM labelling of QAM_modulator.cpp:127-139: level = 2*gray[x] - (2^order - 1), x = the rail's bits MSB first
__device__ __forceinline__ int gray_pam_level(int x, int order) {
    const unsigned long long gray = 0xAB98DCEF54672310ull;   // nibble x of {0,1,3,2,7,6,4,5,15,14,12,13,8,9,11,10}
    return 2 * (int)((gray >> (4 * x)) & 15ull) - ((1 << order) - 1);
}

struct ChannelArgs {
    double *llr;            // [B][N] out, decoder order
    const uint8_t *tx;      // [ncw][ntx] transmitted bits in CHANNEL order (after the direct interleaver, zero padded to whole
                            // symbols), or null = the all-zero codeword upstream sends (bp_simulation.cpp:568)
    const int32_t *scatter; // [N] decoder index of channel bit j (= the direct map: y[i] = buffer[inverse[i]], :684), or null = identity
    long long B, first_frame;
    int N, ntx, ncw;
    int punct_start;        // decoder index where the punctured tail begins (N when nothing is punctured)
    double sigma;           // bp_simulation.cpp:445 (BPSK) or :449 (QAM)
    double punct_val;       // :700
    double T;               // demapper cut-off (QAM16+)
    uint64_t seed;
};

// H = 0: one Box-Muller pair -> two BPSK / QAM4 LLRs, llr = -2.0 * (sigma*g + 2.0*bit - 1.0) / (sigma*sigma) (:603 / :610).
// H = 2, 3, 4: one pair -> the two rails of one 16/64/256-QAM symbol: x = GrayPAM(bits) + sigmaQAM*g (fresh per frame, the
// evidently intended chain), LLR = -Demodulate(x) (:626-628); the last symbol's missing bits are zero
// (:575) and only the first N LLRs are kept.  Noise keys: (seed, global frame, pair or symbol index, tag 0 / 1) -- independent of
// batching, sharding, interleaver and codeword.
template <int H>
__global__ void __launch_bounds__(256) channel_llr_kernel(const ChannelArgs a) {
    constexpr int m = H == 0 ? 2 : 2 * H;           // channel bits per noise pair
    const int units = (a.N + m - 1) / m;
    const long long total = a.B * (long long)units;
    const double s2 = a.sigma * a.sigma, N0 = 2.0 * s2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / units;
        const int u = (int)(i - b * units);
        double g0, g1;
        gauss_pair(a.seed, (uint64_t)(a.first_frame + b), (uint32_t)u, H == 0 ? 0u : 1u, g0, g1);
        const uint8_t *tx = a.tx ? a.tx + (size_t)((a.first_frame + b) % a.ncw) * a.ntx + (size_t)m * u : nullptr;
        double bit[m];
        if constexpr (H == 0) {
            const double c0 = tx ? (double)tx[0] : 0.0, c1 = (tx && 2 * u + 1 < a.ntx) ? (double)tx[1] : 0.0;
            bit[0] = -2.0 * (a.sigma * g0 + 2.0 * c0 - 1.0) / s2;
            bit[1] = -2.0 * (a.sigma * g1 + 2.0 * c1 - 1.0) / s2;
        } else {
            int xi = 0, xq = 0;
            if (tx) {
#pragma unroll
                for (int h = 0; h < H; ++h) { xi = 2 * xi + tx[h]; xq = 2 * xq + tx[H + h]; }   // z1 = p * bits, p = {2^(H-1) .. 1}
            }
            double bi[H], bq[H];
            demod_rail<H>((double)gray_pam_level(xi, H) + a.sigma * g0, N0, a.T, 0, bi);
            demod_rail<H>((double)gray_pam_level(xq, H) + a.sigma * g1, N0, a.T, 0, bq);
#pragma unroll
            for (int h = 0; h < H; ++h) { bit[h] = -bi[h]; bit[H + h] = -bq[h]; }
        }
        double *row = a.llr + b * (long long)a.N;
        const int j0 = m * u;
        if (!a.scatter && j0 + m <= a.N && (a.N & 1) == 0) {
#pragma unroll
            for (int h = 0; h < m; h += 2) {   // 16-byte stores: N and m even keep every pair aligned
                const double v0 = j0 + h >= a.punct_start ? a.punct_val : bit[h], v1 = j0 + h + 1 >= a.punct_start ? a.punct_val : bit[h + 1];
                *reinterpret_cast<double2 *>(row + j0 + h) = make_double2(v0, v1);
            }
        } else {
#pragma unroll
            for (int h = 0; h < m; ++h)
                if (j0 + h < a.N) {
                    const int o = a.scatter ? a.scatter[j0 + h] : j0 + h;
                    row[o] = o >= a.punct_start ? a.punct_val : bit[h];
                }
        }
    }
}

// QAM_modulator() at function level: bits [ns][m] (0/1 bytes, I rail first, MSB first) -> x [ns][2] (I, Q)
struct ModArgs {
    const uint8_t *bits;
    double *x;
    long long ns;
    int H;
};
__global__ void __launch_bounds__(256) qam_modulate_kernel(const ModArgs a) {
    for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s < a.ns; s += (long long)gridDim.x * blockDim.x) {
        const uint8_t *b = a.bits + s * (2 * a.H);
        int xi = 0, xq = 0;
        for (int h = 0; h < a.H; ++h) { xi = 2 * xi + (b[h] & 1); xq = 2 * xq + (b[a.H + h] & 1); }
        *reinterpret_cast<double2 *>(a.x + 2 * s) = make_double2((double)gray_pam_level(xi, a.H), (double)gray_pam_level(xq, a.H));
    }
}

// ---- error accounting -----------------------------------------------------------------------------------
struct CountArgs {
    const uint32_t *hard;  // [B][hard_words]
    const int32_t *iters;  // [B]
    int32_t *frame_info;   // [B] or null
    unsigned long long *counters;  // [5]: nse, nde, nue, frames, sum |iters|
    long long B;
    int hard_words, R;
    const uint32_t *cw;    // [ncw][hard_words] transmitted codewords, packed like `hard`, or null = all-zero (bp_simulation.cpp:568)
    int ncw;
    long long first_frame; // global index of frame 0: frame f carried codeword (first_frame + f) % ncw
};

// One wavefront per frame: lanes read the frame's packed words coalesced, popcount, butterfly-reduce.
// Block totals go out as one atomic per counter per block (Guideline 12).
__global__ void __launch_bounds__(256) count_errors_kernel(const CountArgs a) {
    __shared__ unsigned long long part[4][5];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long nse = 0, nde = 0, nue = 0, frames = 0, sit = 0;
    for (long long fr = (long long)blockIdx.x * 4 + wv; fr < a.B; fr += (long long)gridDim.x * 4) {
        uint32_t all = 0, info = 0;
        for (int w = lane; w < a.hard_words; w += 64) {
            uint32_t x = a.hard[fr * a.hard_words + w];
            if (a.cw) x ^= a.cw[(size_t)((a.first_frame + fr) % a.ncw) * a.hard_words + w];   // decword[i] != codeword[i] (:735-742)
            all += __popc(x);
            // information bits are indices >= R (bp_simulation.cpp:738)
            const int lo = 32 * w;
            uint32_t m = 0xffffffffu;
            if (lo + 32 <= a.R) m = 0u;
            else if (lo < a.R) m = 0xffffffffu << (a.R - lo);
            info += __popc(x & m);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            all += __shfl_xor(all, o);
            info += __shfl_xor(info, o);
        }
        if (lane == 0) {
            const int it = a.iters[fr];
            if (a.frame_info) a.frame_info[fr] = (int32_t)info | (all ? (1 << 30) : 0);
            frames += 1;
            sit += (unsigned long long)(it < 0 ? -it : it);
            if (all) {                       // :805-810
                nse += info;
                nde += 1;
                if (it >= 0) nue += 1;
            }
        }
    }
    if (lane == 0) { part[wv][0] = nse; part[wv][1] = nde; part[wv][2] = nue; part[wv][3] = frames; part[wv][4] = sit; }
    __syncthreads();
    if (threadIdx.x < 5) {
        const unsigned long long t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (t) atomicAdd(&a.counters[threadIdx.x], t);
    }
}

// ---- integer min-sum quantiser scale ----------------------------------------------------------------------
// imin_sum_decod_qc_lm normalises a frame by its energy, en = sum y[i]^2 accumulated in index order (decoders.cpp:5472-5480),
// so the rounding of that sequential sum is part of the result.  One LANE per frame walks its frame in order; the
// [64 frames][64 values] tiles go through LDS so that the global loads stay coalesced (row stride 65: conflict free).
struct ImsCoefArgs {
    const double *llr;   // [B][N]
    double *coef;        // [B] sqrt(N / en)
    long long B;
    int N;
};

__global__ void __launch_bounds__(64) ims_coef_kernel(const ImsCoefArgs a) {
    __shared__ double tile[64 * 65];
    const int lane = threadIdx.x;
    const long long f0 = (long long)blockIdx.x * 64;
    double en = 0;
    for (int c0 = 0; c0 < a.N; c0 += 64) {
        const int i = c0 + lane;
        double v[64];
#pragma unroll
        for (int f = 0; f < 64; ++f) {          // 64 coalesced 512-byte rows in flight per wave: the kernel is a pure HBM stream
            const long long fr = f0 + f;
            v[f] = (fr < a.B && i < a.N) ? a.llr[fr * a.N + i] : 0.0;   // + 0*0 leaves the sum untouched
        }
#pragma unroll
        for (int f = 0; f < 64; ++f) tile[f * 65 + lane] = v[f];
        __syncthreads();
#pragma unroll 16
        for (int j = 0; j < 64; ++j) { const double x = tile[lane * 65 + j]; en += x * x; }
        __syncthreads();
    }
    if (f0 + lane < a.B) a.coef[f0 + lane] = sqrt((double)a.N / en);                    // :5481
}

// ---- interleaver application --------------------------------------------------------------------------------
// out[b][i] = in[b][map[i]]  (Permutation(), direct_inverse_perm.cpp:785-900, with the per-mode index arithmetic folded into
// one map by ldpc/interleaver.h).  Streaming gather: writes coalesced, reads scattered inside one frame (L2 resident).
struct PermuteArgs {
    const double *in;
    double *out;
    const int32_t *map;   // [N]
    long long B;
    int N;
};

__global__ void __launch_bounds__(256) permute_kernel(const PermuteArgs a) {
    const long long total = a.B * (long long)a.N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / a.N;
        const int v = (int)(i - b * a.N);
        a.out[i] = a.in[b * a.N + a.map[v]];
    }
}

}  // namespace ldpc
