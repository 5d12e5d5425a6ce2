// ldpc_mt.hpp -- upstream's channel noise, bit for bit, generated on the device (included by ldpc_hip.hip).
//
// bp_simulation() draws the noise of frame after frame from ONE std::mt19937 through a FRESH std::normal_distribution<double> per
// sample (commons_portable.cpp:140,174-178; frame loop bp_simulation.cpp:600-611).  On the host that caps an exact replay at
// ~7e3 frames/s per core (68.8 ns per sample).  Both pieces are public algorithms with no hidden state, so the device
// can reproduce the stream itself:
//
//   mt19937 (Matsumoto & Nishimura 1998): x[i+624] = x[i+397] ^ twist(x[i], x[i+1]) is linear over GF(2), so the state J words
//       ahead is g_J(F) applied to the state, g_J = x^J mod phi, phi = the minimal polynomial of the recurrence (degree 19937)
//       [Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer 2008].  With x^J = g_J mod phi every word of the sequence obeys
//       x[t+J] = XOR over the set bits i of g_J of x[t+i] -- a GF(2) convolution over 19937 + 624 consecutive words
//       (mt_jump_kernel, one workgroup per jump, the words in LDS).  The host computes phi (Berlekamp-Massey on an output bit
//       sequence) and g_J for J = 2^20 .. 2^29 by repeated squaring once per process (~60 ms) and checks the first against a plain
//       2^20-step walk.  S stream states are reached in ceil(log2 S) doubling rounds; then one WAVEFRONT per stream runs the
//       recurrence through a 1024-word LDS ring, 192 words per step (mt_generate_kernel), writing the untempered words to HBM.
//   libstdc++'s normal_distribution (bits/random.tcc: Marsaglia polar method): an attempt takes four 32-bit words (two
//       generate_canonical<double,53> values), is accepted with probability pi/4 and, because upstream constructs a new distribution
//       per call, yields ONE value (the cached second one is dropped).  So sample k of the run is the k-th
//       accepted attempt: count per block -> exclusive scan -> emit (mt_polar_kernel), log() by glibc's algorithm
//       (ldpc_spec::log_glibc), sqrt and division IEEE-correct on the device, operation order of random.tcc:1821-1833.
//       The emit pass applies upstream's LLR formula, inverse interleaver and puncturing (bp_simulation.cpp:603/610, :684,
//       :697-710) and notes where each frame's draws end, so the generator can be handed back to the host at any frame boundary.
//
// All of it is HBM / LDS / integer-VALU work: 5.1 words (20 B) written and read twice per LLR (8 B) produced.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

namespace ldpc_mt {

constexpr int MTN = 624;
constexpr int kLog2Stride = 20;                       // words per stream
constexpr long long kStride = 1ll << kLog2Stride;
constexpr int kLevels = 10;                           // jump polynomials for 2^20 .. 2^29 words: up to 1024 streams per round
constexpr int kMaxStreams = 1 << kLevels;
constexpr int kDegree = 19937;
constexpr int kSeqWords = 20608;                      // >= 19937 + 624 + 31: the LDS image of one jump (82 432 bytes)
constexpr unsigned long long kRoundItems = 1ull << 27;   // samples per generation round (5.1 words each: <= 2.8 GB of words)

__host__ __device__ inline uint32_t mt_next(uint32_t x0, uint32_t x1, uint32_t xm) {   // x[i+624] from x[i], x[i+1], x[i+397]
    const uint32_t y = (x0 & 0x80000000u) | (x1 & 0x7fffffffu);
    return xm ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__host__ __device__ inline uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// ---- host: the jump polynomials -------------------------------------------------------------------------------------------
struct JumpPolys {
    std::vector<uint32_t> poly;   // [kLevels][624]: bit i of level e = coefficient of x^i in x^(2^(20+e)) mod phi
    bool ok = false;
    std::string err;
};

// x[0..n) of the sequence that starts with the 624 state words
inline void sequence_host(const uint32_t *state, uint32_t *x, int n) {
    for (int i = 0; i < MTN; ++i) x[i] = state[i];
    for (int i = MTN; i < n; ++i) x[i] = mt_next(x[i - 624], x[i - 623], x[i - 227]);
}

// out[k] = x[J + k]: exact for k >= 1; of word 0 only bit 31 belongs to the state (the recurrence never reads the rest)
inline void jump_host(const uint32_t *state, const uint32_t *poly, uint32_t *out) {
    std::vector<uint32_t> x((size_t)kSeqWords);
    sequence_host(state, x.data(), kSeqWords);
    for (int k = 0; k < MTN; ++k) out[k] = 0;
    for (int i = 0; i < kDegree; ++i)
        if ((poly[i >> 5] >> (i & 31)) & 1u)
            for (int k = 0; k < MTN; ++k) out[k] ^= x[(size_t)i + k];
}

inline void compute_jump_polys(JumpPolys &J) {
    typedef unsigned long long u64;
    const int K = kDegree, W = (K + 64) / 64 + 1;       // 313 words hold degree <= 19967
    // 1. an output bit sequence of the recurrence: bit 0 of x[t], t >= 1 (a linear functional of the 19937-bit state)
    std::vector<uint32_t> st(MTN);
    st[0] = 5489u;
    for (int i = 1; i < MTN; ++i) st[(size_t)i] = 1812433253u * (st[(size_t)i - 1] ^ (st[(size_t)i - 1] >> 30)) + (uint32_t)i;
    const int NB = 2 * K + 64;
    std::vector<uint32_t> x((size_t)NB + MTN + 1);
    sequence_host(st.data(), x.data(), NB + MTN + 1);
    // 2. Berlekamp-Massey over GF(2), bit-packed.  rev holds s[n-i] at bit i.
    std::vector<u64> Cp((size_t)2 * W, 0), Bp((size_t)2 * W, 0), Tp((size_t)2 * W, 0), rev((size_t)2 * W, 0);
    Cp[0] = 1; Bp[0] = 1;
    int L = 0, m = 1;
    auto xor_shifted = [&](std::vector<u64> &dst, const std::vector<u64> &src, int sh) {
        const int ws = sh >> 6, bs = sh & 63;
        for (int w = 2 * W - 1 - ws; w >= 0; --w) {
            u64 v = src[(size_t)w] << bs;
            if (bs && w > 0) v |= src[(size_t)w - 1] >> (64 - bs);
            dst[(size_t)w + ws] ^= v;
        }
    };
    for (int n = 0; n < NB; ++n) {
        for (int w = 2 * W - 1; w > 0; --w) rev[(size_t)w] = (rev[(size_t)w] << 1) | (rev[(size_t)w - 1] >> 63);
        rev[0] = (rev[0] << 1) | (u64)(x[(size_t)n + 1] & 1u);
        u64 acc = 0;
        const int lw = L / 64 + 1;
        for (int w = 0; w < lw; ++w) acc ^= Cp[(size_t)w] & rev[(size_t)w];
        if (__builtin_parityll(acc)) {
            if (2 * L <= n) {
                Tp = Cp;
                xor_shifted(Cp, Bp, m);
                L = n + 1 - L;
                Bp = Tp;
                m = 1;
            } else {
                xor_shifted(Cp, Bp, m);
                ++m;
            }
        } else {
            ++m;
        }
    }
    if (L != K) { J.err = "Berlekamp-Massey did not find a degree-19937 recurrence"; return; }
    // 3. phi(x) = x^K C(1/x); phis[s] = phi << s
    std::vector<u64> phi((size_t)W + 1, 0);
    for (int j = 0; j <= K; ++j)
        if ((Cp[(size_t)(K - j) >> 6] >> ((K - j) & 63)) & 1ull) phi[(size_t)j >> 6] |= 1ull << (j & 63);
    std::vector<std::vector<u64>> phis(64, std::vector<u64>((size_t)W + 1, 0));
    for (int s = 0; s < 64; ++s)
        for (int w = 0; w <= W; ++w) {
            u64 v = phi[(size_t)w] << s;
            if (s && w > 0) v |= phi[(size_t)w - 1] >> (64 - s);
            phis[(size_t)s][(size_t)w] = v;
        }
    // 4. q <- q^2 mod phi, starting from q = x
    auto spread = [](uint32_t v) {
        u64 t = v;
        t = (t | (t << 16)) & 0x0000FFFF0000FFFFull;
        t = (t | (t << 8)) & 0x00FF00FF00FF00FFull;
        t = (t | (t << 4)) & 0x0F0F0F0F0F0F0F0Full;
        t = (t | (t << 2)) & 0x3333333333333333ull;
        t = (t | (t << 1)) & 0x5555555555555555ull;
        return t;
    };
    std::vector<u64> q((size_t)W, 0), sq((size_t)2 * W + 2, 0);
    q[0] = 2;
    J.poly.assign((size_t)kLevels * MTN, 0u);
    for (int e = 1; e < kLog2Stride + kLevels; ++e) {
        std::fill(sq.begin(), sq.end(), 0ull);
        for (int w = 0; w < W; ++w) {
            sq[(size_t)2 * w] = spread((uint32_t)q[(size_t)w]);
            sq[(size_t)2 * w + 1] = spread((uint32_t)(q[(size_t)w] >> 32));
        }
        for (int d = 2 * K - 2; d >= K; --d)
            if ((sq[(size_t)d >> 6] >> (d & 63)) & 1ull) {
                const int off = d - K, ws = off >> 6;
                const std::vector<u64> &p = phis[(size_t)off & 63];
                for (int w = 0; w <= W; ++w) sq[(size_t)ws + w] ^= p[(size_t)w];
            }
        for (int w = 0; w < W; ++w) q[(size_t)w] = sq[(size_t)w];
        if (e >= kLog2Stride) {
            uint32_t *out = J.poly.data() + (size_t)(e - kLog2Stride) * MTN;
            for (int i = 0; i < K; ++i)
                if ((q[(size_t)i >> 6] >> (i & 63)) & 1ull) out[i >> 5] |= 1u << (i & 31);
        }
    }
    // 5. self-check of level 0 against a plain walk of 2^20 steps
    {
        std::vector<uint32_t> jumped(MTN);
        // walk: a sliding window of 624 words
        std::vector<uint32_t> win((size_t)MTN);
        for (int i = 0; i < MTN; ++i) win[(size_t)i] = st[(size_t)i];
        int head = 0;   // win[(head + k) % 624] = x[t + k]
        for (long long t = 0; t < kStride; ++t) {
            const uint32_t nx = mt_next(win[(size_t)head], win[(size_t)(head + 1) % MTN], win[(size_t)(head + 397) % MTN]);
            win[(size_t)head] = nx;
            head = (head + 1) % MTN;
        }
        jump_host(st.data(), J.poly.data(), jumped.data());
        bool same = ((jumped[0] ^ win[(size_t)head]) & 0x80000000u) == 0;
        for (int k = 1; k < MTN; ++k) same = same && jumped[(size_t)k] == win[(size_t)(head + k) % MTN];
        if (!same) { J.err = "jump polynomial self-check failed"; return; }
    }
    J.ok = true;
}

inline const JumpPolys &jump_polys() {
    static JumpPolys J;
    static std::once_flag once;
    std::call_once(once, [] { compute_jump_polys(J); });
    return J;
}

// ---- device: jump --------------------------------------------------------------------------------------------------------
struct JumpArgs {
    uint32_t *states;        // [S][624]
    const uint32_t *poly;    // [624] of this level
    int src_first, dst_first;
};

__global__ void __launch_bounds__(640) mt_jump_kernel(const JumpArgs a) {
    extern __shared__ uint32_t mt_x[];   // kSeqWords
    const int tid = threadIdx.x;
    const uint32_t *src = a.states + (size_t)(a.src_first + blockIdx.x) * MTN;
    uint32_t *dst = a.states + (size_t)(a.dst_first + blockIdx.x) * MTN;
    if (tid < MTN) mt_x[tid] = src[tid];
    __syncthreads();
    for (int base = MTN; base < kSeqWords; base += 224) {   // 224 < 227: every operand was written in an earlier step
        const int i = base + tid;
        if (tid < 224 && i < kSeqWords) mt_x[i] = mt_next(mt_x[i - 624], mt_x[i - 623], mt_x[i - 227]);
        __syncthreads();
    }
    if (tid < MTN) {
        uint32_t acc = 0;
        for (int iw = 0; iw < MTN; ++iw) {
            const uint32_t w = a.poly[iw];   // uniform: a scalar load
            if (w == 0) continue;
            const uint32_t *xp = mt_x + iw * 32 + tid;
#pragma unroll
            for (int b = 0; b < 32; ++b) acc ^= xp[b] & (0u - ((w >> b) & 1u));
        }
        dst[tid] = acc;
    }
}

// ---- device: one wavefront per stream ---------------------------------------------------------------------------------------
struct GenArgs {
    const uint32_t *states;   // [S][624]
    uint32_t *xraw;           // [624 + S * kStride] untempered words; stream j writes [624 + j*kStride, 624 + (j+1)*kStride)
    int S;
};

__global__ void __launch_bounds__(256) mt_generate_kernel(const GenArgs a) {
    __shared__ uint32_t ring_all[4][1024];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wv;
    if (j >= a.S) return;    // no workgroup barrier below: a wave is on its own
    uint32_t *ring = ring_all[wv];
    const uint32_t *st = a.states + (size_t)j * MTN;
    for (int i = lane; i < MTN; i += 64) {
        const uint32_t v = st[i];
        ring[i] = v;
        if (j == 0) a.xraw[i] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t *out = a.xraw + (size_t)j * kStride;
    const uint32_t end = 624u + (uint32_t)kStride;
    for (uint32_t idx0 = 624u; idx0 < end;) {
        const int ns = end - idx0 >= 192u ? 3 : 1;       // 2^20 / 64 = 3 * 5461 + 1
        uint32_t x0[3], x1[3], xm[3];
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (u < ns) {
                const uint32_t i = idx0 + 64u * u + lane;    // LDS operations of one wave execute in order; all operands are >= 227 back
                x0[u] = ring[(i - 624u) & 1023u];
                x1[u] = ring[(i - 623u) & 1023u];
                xm[u] = ring[(i - 227u) & 1023u];
            }
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (u < ns) {
                const uint32_t i = idx0 + 64u * u + lane;
                const uint32_t v = mt_next(x0[u], x1[u], xm[u]);
                ring[i & 1023u] = v;
                out[i] = v;
            }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        idx0 += 64u * ns;
    }
}

// ---- device: the polar method over the word stream --------------------------------------------------------------------------
// std::generate_canonical<double, 53>(mt19937) (random.tcc:3348-3380): two words, sum = w0 + w1 * 2^32 rounded once, / 2^64
__device__ __forceinline__ double mt_canonical(uint32_t w0, uint32_t w1) {
    double sum = (double)w0;
    sum += (double)w1 * 4294967296.0;
    double r = sum * 0x1p-64;                 // exact, as the division by 2^64 is
    if (r >= 1.0) r = 0x1.fffffffffffffp-1;   // nextafter(1, 0)
    return r;
}

struct PolarArgs {
    const uint32_t *xraw;
    long long p;              // first unread word
    long long attempts;       // attempts available: words p + 4a .. p + 4a + 3, a < attempts
    uint32_t *blockcnt;                     // pass 0 out
    const unsigned long long *blockbase;    // pass 1 in
    const unsigned long long *total;        // pass 1 in: accepted attempts of the round
    unsigned long long need;                // items wanted from this round
    int per_frame;                          // 0: items are samples; else samples per frame (only whole frames are emitted)
    long long *end_t;                       // pass 1 out: word after the last emitted item's draws
    double *out;                            // samples, or LLR rows [row_hi - row_lo][N]; may be null (skip)
    // frame mode
    long long first_frame;                  // global index of frame 0 of this round (codeword choice)
    long long row_lo, row_hi;               // frames of this round whose rows are written (a shard's slice)
    const uint8_t *tx;
    const int32_t *scatter;
    int ntx, ncw, punct_start;
    double sigma, punct_val;
};

constexpr int kPolarSub = 8;   // 256 * 8 attempts per workgroup

template <int PASS>
__global__ void __launch_bounds__(256) mt_polar_kernel(const PolarArgs a) {
    __shared__ uint32_t wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long a0 = (long long)blockIdx.x * (256 * kPolarSub);
    unsigned long long running = 0, limit = 0;
    if (PASS == 1) {
        running = a.blockbase[blockIdx.x];
        limit = *a.total < a.need ? *a.total : a.need;
        if (a.per_frame) limit -= limit % (unsigned long long)a.per_frame;
        if (running >= limit) return;
    }
    uint32_t mine = 0;
    for (int sub = 0; sub < kPolarSub; ++sub) {
        const long long at = a0 + sub * 256 + tid;
        bool ok = false;
        double y = 0, r2 = 1;
        if (at < a.attempts) {
            const uint32_t *w = a.xraw + a.p + 4 * at;
            const double u0 = mt_canonical(mt_temper(w[0]), mt_temper(w[1]));
            const double u1 = mt_canonical(mt_temper(w[2]), mt_temper(w[3]));
            const double x = 2.0 * u0 - 1.0;               // random.tcc:1821-1825
            y = 2.0 * u1 - 1.0;
            r2 = x * x + y * y;
            ok = !(r2 > 1.0 || r2 == 0.0);
        }
        const unsigned long long bal = __ballot(ok);
        if (PASS == 0) {
            if (lane == 0) mine += (uint32_t)__popcll(bal);
        } else {
            if (lane == 0) wsum[wv] = (uint32_t)__popcll(bal);
            __syncthreads();
            unsigned long long g = running + (unsigned long long)__popcll(bal & ((1ull << lane) - 1ull));
            for (int v = 0; v < wv; ++v) g += wsum[v];
            const unsigned long long tot = (unsigned long long)wsum[0] + wsum[1] + wsum[2] + wsum[3];
            __syncthreads();
            if (ok && g < limit) {
                if (g + 1 == limit) *a.end_t = a.p + 4 * (at + 1);
                if (a.out) {
                    const double mult = sqrt(-2.0 * ldpc_spec::log_glibc(r2) / r2);   // random.tcc:1827
                    double ret = y * mult;                                               // :1830
                    ret = ret * 1.0 + 0.0;                                               // :1833 stddev 1, mean 0
                    if (!a.per_frame) {
                        a.out[g] = ret;
                    } else {
                        const long long f = (long long)(g / (unsigned long long)a.per_frame);
                        const int i = (int)(g - (unsigned long long)f * (unsigned long long)a.per_frame);
                        if (f >= a.row_lo && f < a.row_hi) {
                            const double c = a.tx ? (double)a.tx[(size_t)((a.first_frame + f) % a.ncw) * a.ntx + i] : 0.0;
                            const double v = -2.0 * (a.sigma * ret + 2.0 * c - 1.0) / (a.sigma * a.sigma);   // bp_simulation.cpp:603 / :610
                            const int o = a.scatter ? a.scatter[i] : i;                                      // :684
                            a.out[(f - a.row_lo) * (long long)a.per_frame + o] = o >= a.punct_start ? a.punct_val : v;   // :697-710
                        }
                    }
                }
            }
            running += tot;
            if (running >= limit) return;   // uniform
        }
    }
    if (PASS == 0) {
        if (lane == 0) wsum[wv] = mine;
        __syncthreads();
        if (tid == 0) a.blockcnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

// exclusive scan of the per-block counts (one workgroup; nb is a few 10^4)
__global__ void __launch_bounds__(1024) mt_scan_kernel(const uint32_t *cnt, unsigned long long *base, unsigned long long *total, long long nb) {
    __shared__ unsigned long long part[1024];
    const int tid = threadIdx.x;
    const long long per = (nb + 1023) / 1024, lo = per * tid, hi = lo + per < nb ? lo + per : nb;
    unsigned long long s = 0;
    for (long long i = lo; i < hi; ++i) s += cnt[i];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const unsigned long long v = tid >= o ? part[tid - o] : 0ull;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned long long run = part[tid] - s;
    for (long long i = lo; i < hi; ++i) { base[i] = run; run += cnt[i]; }
    if (tid == 1023) *total = part[1023];
}

// the generator continues at word *end_t: its state is the 624 words from there, position 0
__global__ void __launch_bounds__(640) mt_adopt_kernel(const uint32_t *xraw, const long long *end_t, uint32_t *state) {
    if (threadIdx.x < MTN) state[threadIdx.x] = xraw[*end_t + threadIdx.x];
}

// per-context device state of the generator (ldpc_hip_mt_* entry points)
struct DeviceState {
    bool set = false;
    int pos = 0;                         // next word of d_state to draw (0..624); 0 after every generation round
    long long h_end = 0;                 // staging for the end-offset preset
    uint32_t *d_state = nullptr;         // [624]
    uint32_t *d_poly = nullptr;          // [kLevels][624]
    uint32_t *d_states = nullptr;        // [cap_streams][624]
    int cap_streams = 0;
    uint32_t *d_xraw = nullptr;          // [cap_words]
    size_t cap_words = 0;
    uint32_t *d_blockcnt = nullptr;
    unsigned long long *d_blockbase = nullptr;
    long long cap_blocks = 0;
    unsigned long long *d_total = nullptr;
    long long *d_end_t = nullptr;
    long long frames_taken = 0;          // frames drawn since ldpc_hip_mt_set_state (codeword choice f % ncw)
    int32_t *d_info = nullptr, *d_iters = nullptr;
    long long cap_rec = 0;
};

inline void release(DeviceState &m) {
    void *ptrs[] = {m.d_state, m.d_poly, m.d_states, m.d_xraw, m.d_blockcnt, m.d_blockbase, m.d_total, m.d_end_t, m.d_info, m.d_iters};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    m = DeviceState();
}

}  // namespace ldpc_mt
