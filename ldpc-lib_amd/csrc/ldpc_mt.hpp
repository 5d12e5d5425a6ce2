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
//       (mt_jump_kernel: the words in LDS, up to 8 workgroups per jump, each taking a share of g_J's ~10^4 set coefficients).  The host computes phi (Berlekamp-Massey on an output bit
//       sequence) and g_J for J = 2^18 .. 2^30 by repeated squaring once per process (~60 ms) and checks the first against a plain
//       2^18-step walk.  S stream states are reached in ceil(log2 S) doubling rounds; then one WAVEFRONT per stream runs the
//       recurrence in place over its 624-word block in LDS, 192 words per step with every LDS address an immediate offset
//       (mt_generate_kernel), writing the untempered words to HBM.
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
constexpr int kLog2Stride = 18;                       // shortest stream: 2^18 words (a round picks 2^18, 2^19 or 2^20 by its size)
constexpr int kLog2StrideMax = 20;
constexpr long long kStride = 1ll << kLog2Stride;
constexpr int kLevels = 13;                           // jump polynomials for 2^18 .. 2^30 words
constexpr int kMaxStreams = 1024;                     // streams per round
constexpr int kDegree = 19937;
constexpr int kMaxBits = 10752;                        // set coefficients per polynomial: 19937 / 2 +- a few hundred
constexpr int kBitsPad = 192;                          // readable entries behind the last table (the jump kernel requests exponents ahead)
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
    std::vector<uint32_t> poly;   // [kLevels][624]: bit i of level e = coefficient of x^i in x^(2^(18+e)) mod phi
    std::vector<uint32_t> bits;   // [kLevels][kMaxBits]: the exponents i with a set coefficient, ascending (what the device walks)
    std::vector<int> nbits;       // [kLevels]
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
    J.bits.assign((size_t)kLevels * kMaxBits, 0u);
    J.nbits.assign((size_t)kLevels, 0);
    for (int e = 0; e < kLevels; ++e) {
        const uint32_t *pl = J.poly.data() + (size_t)e * MTN;
        int n = 0;
        for (int i = 0; i < K; ++i)
            if ((pl[i >> 5] >> (i & 31)) & 1u) {
                if (n == kMaxBits) { J.err = "jump polynomial has more set coefficients than expected"; return; }
                J.bits[(size_t)e * kMaxBits + n++] = (uint32_t)i;
            }
        J.nbits[(size_t)e] = n;
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
// One jump = `parts` workgroups: each rebuilds the words its share of the polynomial's set coefficients reaches (the recurrence
// from the source state, in LDS) and XORs its partial sums into the destination state (zeroed by the host beforehand).
struct JumpArgs {
    uint32_t *states;        // [S][624]
    const uint32_t *bits;    // ascending exponents of this level's polynomial
    int nbits, parts;
    int src_first, dst_first;
};

constexpr int kJumpThreads = 640;   // ten waves: thread t < 624 makes state word t

// Per set coefficient p, state word k gains x[k + p]: an LDS read at a uniform offset per thread.  What bounds the loop is the LDS
// (about six cycles per wave-wide 32-bit read on this part: ~60 cycles per coefficient for the ten waves), so the exponents must
// not add latency in front of the reads.  As scalar loads they share the LDS's counter (lgkmcnt) and cannot be requested ahead;
// they therefore come through the VECTOR memory path (its own counter) from an address the compiler cannot see to be uniform, 64
// at a time as sixteen 16-byte loads with progressive waits -- the table is padded, so reading ahead of a part's end is harmless.
// Measured per 256 full jumps: 276 us with one s_load per eight exponents in front of every eight reads (round 2), 240 us this
// way; 305 us with three words per thread and v_readlane broadcasts (four waves); 220 us with 8-byte reads of word PAIRS (exponent
// lists split by parity so that every pair is aligned; five waves) -- but 43 instead of 38 us for the small early rounds, no net
// gain, not kept.  The LDS delivers ~45 bytes per clock and CU to this access pattern whatever the access width.
__global__ void __launch_bounds__(kJumpThreads) mt_jump_kernel(const JumpArgs a) {
    extern __shared__ uint32_t mt_x[];   // kSeqWords
    const int tid = threadIdx.x;
    const int jump = blockIdx.x / a.parts, part = blockIdx.x - jump * a.parts;
    const uint32_t *src = a.states + (size_t)(a.src_first + jump) * MTN;
    uint32_t *dst = a.states + (size_t)(a.dst_first + jump) * MTN;
    const int j_lo = (int)((long long)a.nbits * part / a.parts) & ~3;   // multiples of 4: the exponent loads are 16 bytes wide
    const int j_hi = part + 1 == a.parts ? a.nbits : (int)((long long)a.nbits * (part + 1) / a.parts) & ~3;
    if (j_lo >= j_hi) return;
    const int need = (int)a.bits[j_hi - 1] + MTN;   // words x[0 .. need) cover this part
    if (tid < MTN) mt_x[tid] = src[tid];
    __syncthreads();
    for (int base = MTN; base < need; base += 224) {   // 224 < 227: every operand was written in an earlier step
        const int i = base + tid;
        if (tid < 224 && i < need) mt_x[i] = mt_next(mt_x[i - 624], mt_x[i - 623], mt_x[i - 227]);
        __syncthreads();
    }
    if (tid < MTN) {
        uint32_t acc = 0;
        const uint32_t *xp = mt_x + tid;
        int opaque = 0;
        asm volatile("" : "+v"(opaque));
        const uint4 *bq = reinterpret_cast<const uint4 *>(a.bits + j_lo + opaque);   // j_lo is a multiple of 4: 16-byte aligned
        const int n = j_hi - j_lo, nb32 = n >> 5;
        uint4 e[8], f[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) e[u] = bq[u];
        for (int b = 0; b < nb32; ++b) {
#pragma unroll
            for (int u = 0; u < 8; ++u) f[u] = bq[(b + 1) * 8 + u];
            uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                t0 ^= xp[e[u].x]; t1 ^= xp[e[u].y]; t2 ^= xp[e[u].z]; t3 ^= xp[e[u].w];
            }
            acc ^= (t0 ^ t1) ^ (t2 ^ t3);
#pragma unroll
            for (int u = 0; u < 8; ++u) e[u] = f[u];
        }
        const uint32_t *bp = a.bits + j_lo;
        for (int j = nb32 << 5; j < n; ++j) acc ^= xp[bp[j]];
        if (a.parts == 1) dst[tid] = acc;
        else atomicXor(&dst[tid], acc);
    }
}

// ---- device: one wavefront per stream ---------------------------------------------------------------------------------------
struct GenArgs {
    const uint32_t *states;   // [S][624] start states of the window's streams
    uint32_t *xraw;           // the window: xraw[0] = tape word first * stride; stream j (of the window) writes [624 + j*stride, 624 + (j+1)*stride)
    int S;                    // streams in the window
    long long first;          // tape index of the window's first stream
    int log2_stride;          // 18 .. 20
    long long words;          // tape words wanted behind the first 624 (a multiple of 64): streams stop there
};

// The twist of the recurrence: x[i+624] = x[i+397] ^ mt_twist(x[i], x[i+1]).
__device__ __forceinline__ uint32_t mt_twist(uint32_t x0, uint32_t x1) {
    const uint32_t y = (x0 & 0x80000000u) | (x1 & 0x7fffffffu);
    return (y >> 1) ^ ((x1 & 1u) ? 0x9908b0dfu : 0u);
}

// One WORKGROUP per stream, a whole 624-word block per step.  The textbook update walks a block in three dependent stretches (a new
// word needs the new word 227 places back); substituting the recurrence into itself removes that dependence inside a block:
// with Tv[i] = twist(x[i], x[i+1]) of the OLD block,
//     new[i] = old[i+397] ^ Tv[i]                                   i <  227
//            = old[i+170] ^ Tv[i-227] ^ Tv[i]                       227 <= i < 454      (x[i+397] = new[i-227])
//            = old[i-57]  ^ Tv[i-454] ^ Tv[i-227] ^ Tv[i]           454 <= i < 624      (twice)
// (Tv[623] needs x[624] = new[0] = old[397] ^ Tv[0], which the thread of word 623 forms itself.)  So a block costs two barrier
// steps -- every thread twists its own words, then combines three or four LDS values -- instead of ten dependent sub-steps of one
// wave, and a stream of 2^20 words is 1681 steps long instead of ~17 000.  Same words, bit for bit.
constexpr int kGenThreads = 320;   // five waves, two words per thread (624 = 2 * 312)

// workgroup barrier that orders LDS traffic only: the words stored to global memory in the step before must not be waited for
// (__syncthreads() waits for every outstanding store, ~1 us per block here)
__device__ __forceinline__ void mt_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// (Measured: 0.99 ms per 2^27-sample round for round 2's one-wave-per-stream kernel, 0.69 ms for this one; skipping the LDS accesses
// a whole wave does not need -- the zero reads, the operands of word 623 -- behind wave-uniform branches made it slower, 0.78 ms.)
// one block: x (old, LDS) -> nx (new, LDS) and out (HBM); xr = the thread's own words of the old block on entry, of the new on exit.
// Everything address-like is a function of (thread, k) only, i.e. invariant over the block loop; no branch depends on data.
template <int T>
__device__ __forceinline__ void mt_block(const uint32_t *x, uint32_t *nx, uint32_t *tv, const uint32_t *zero, uint32_t (&xr)[(MTN + T - 1) / T],
                                         uint32_t *ob, const uint32_t remaining, const int tid) {
    constexpr int W = (MTN + T - 1) / T, KS = (MTN - 1) / T;   // KS: the k of word 623
    // ---- step 1: every thread twists its own words with their right-hand neighbours.  All loads first, then the arithmetic: one
    // LDS round trip per step, not one per word.
    uint32_t b[W], tw[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const int i = tid + k * T;
        b[k] = x[i < MTN ? i + 1 : 0];
    }
    const uint32_t s0 = x[0], s1 = x[1], s397 = x[397];   // x[624] is the new word 0 = old[397] ^ twist(old[0], old[1])
    __builtin_amdgcn_sched_barrier(0);
    b[KS] = tid + KS * T == MTN - 1 ? (s397 ^ mt_twist(s0, s1)) : b[KS];
#pragma unroll
    for (int k = 0; k < W; ++k) tw[k] = mt_twist(xr[k], b[k]);
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const int i = tid + k * T;
        if (i < 397) tv[i] = tw[k];   // the twists other words need: tv[i - 227] (i - 227 <= 396), tv[i - 454]
    }
    mt_lds_barrier();
    // ---- step 2: combine
    uint32_t o[W], q1[W], q2[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const int i = tid + k * T, ic = i < MTN ? i : 0;
        const int src = ic + 397 - (ic >= 227 ? 227 : 0) - (ic >= 454 ? 227 : 0);
        const uint32_t *p1 = ic >= 227 ? tv + (ic - 227) : zero, *p2 = ic >= 454 ? tv + (ic - 454) : zero;
        o[k] = x[src]; q1[k] = *p1; q2[k] = *p2;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < W; ++k) xr[k] = (o[k] ^ tw[k]) ^ (q1[k] ^ q2[k]);
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const int i = tid + k * T;
        if (i < MTN) {
            nx[i] = xr[k];
            if ((uint32_t)i < remaining) ob[i] = xr[k];   // per word: a stream may end anywhere inside its last block
        }
    }
    mt_lds_barrier();
}

template <int T>
__device__ __forceinline__ void mt_generate_body(const GenArgs &a) {
    constexpr int W = (MTN + T - 1) / T;
    __shared__ uint32_t xb[2][MTN + 16];
    __shared__ uint32_t tv[400];
    __shared__ uint32_t zero[4];
    const int tid = threadIdx.x;
    const int j = blockIdx.x;
    const uint32_t *st = a.states + (size_t)j * MTN;
    const long long gj = a.first + j;   // the stream's index on the tape
    uint32_t xr[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const int i = tid + k * T;
        xr[k] = 0;
        if (i < MTN) {
            xr[k] = st[i];
            xb[0][i] = xr[k];
            if (gj == 0) a.xraw[i] = xr[k];
        }
    }
    if (tid < 4) zero[tid] = 0u;
    if (tid < 16) { xb[0][MTN + tid] = 0u; xb[1][MTN + tid] = 0u; }
    __syncthreads();
    const long long stride = 1ll << a.log2_stride;
    uint32_t *out = a.xraw + (size_t)j * (size_t)stride + MTN;   // the stream's words [0, stride)
    const long long left = a.words - gj * stride;
    const uint32_t nwords = left >= stride ? (uint32_t)stride : left > 0 ? (uint32_t)left : 0u;   // a short round stops early
    for (uint32_t w0 = 0; w0 < nwords; w0 += 2 * MTN) {   // two blocks per trip: the buffers' roles are fixed inside it
        mt_block<T>(xb[0], xb[1], tv, zero, xr, out + w0, nwords - w0, tid);
        if (w0 + MTN >= nwords) break;
        mt_block<T>(xb[1], xb[0], tv, zero, xr, out + w0 + MTN, nwords - w0 - MTN, tid);
    }
}

__global__ void __launch_bounds__(kGenThreads) mt_generate_kernel(const GenArgs a) { mt_generate_body<kGenThreads>(a); }

// ---- device: the polar method over the word stream --------------------------------------------------------------------------
// std::generate_canonical<double, 53>(mt19937) (random.tcc:3348-3380): two words, sum = w0 + w1 * 2^32 rounded once, / 2^64
__device__ __forceinline__ double mt_canonical(uint32_t w0, uint32_t w1) {
    double sum = (double)w0;
    sum += (double)w1 * 4294967296.0;
    double r = sum * 0x1p-64;                 // exact, as the division by 2^64 is
    if (r >= 1.0) r = 0x1.fffffffffffffp-1;   // nextafter(1, 0)
    return r;
}

// The word tape of a round is addressed by GLOBAL word index: word 0 = first of the 624 state words the round starts from, the
// generated words follow from 624 on.  A context holds a WINDOW of it -- xraw[0] is tape word xbase -- which is the whole round on one
// GPU and the sub-streams a shard needs when several GPUs share a round.  Attempt a (global) draws tape words p + 4a .. p + 4a + 3.
struct PolarArgs {
    const uint32_t *xraw;
    long long xbase;          // tape index of xraw[0]
    long long p;              // tape index of the first unread word
    long long at_lo, at_hi;   // this launch covers attempts [at_lo, at_hi)
    unsigned long long base0; // accepted attempts before at_lo == item index of the first accepted attempt of the range
    unsigned long long need;  // items wanted from this round
    int per_frame;            // 0: items are samples; else samples per frame (the caller keeps whole frames only)
    double *out;              // samples, or LLR rows [row_hi - row_lo][N]; may be null (skip)
    // look-back state of the fused pass
    unsigned long long *status;   // [blocks] (flag << 62) | value; zeroed by the host
    unsigned *ticket;             // block id dispenser; zeroed by the host
    // count-only mode: accepted attempts below / from split_at
    long long split_at;
    unsigned long long *counters; // [2]
    // frame mode
    long long first_frame;                  // global index of frame 0 of this round (codeword choice)
    long long row_lo, row_hi;               // frames of this round whose rows are written (a shard's slice)
    const uint8_t *tx;
    const int32_t *scatter;
    int ntx, ncw, punct_start;
    double sigma, punct_val;
};

constexpr int kPolarSub = 8;                       // a wave takes 64 * 8 consecutive attempts, a workgroup 4 waves
constexpr int kPolarBlock = 256 * kPolarSub;       // attempts per workgroup
constexpr unsigned long long kStAgg = 1ull << 62, kStPrefix = 2ull << 62, kStMask = (1ull << 62) - 1ull;

__device__ __forceinline__ unsigned long long mt_wave_sum(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// MODE 0: count only (two counters, split at an attempt index).  MODE 1: the fused pass -- every workgroup tests its attempts,
// obtains the number of accepted attempts in front of it by DECOUPLED LOOK-BACK over the status words of its predecessors
// (Merrill & Garland 2016: a workgroup publishes its own count at once, then sums its predecessors' counts backwards until it
// meets one that already knows its prefix), and emits.  The word stream is read once.  Workgroups number themselves from an atomic
// ticket, so every predecessor of a running workgroup is itself running or done and none of them waits for a successor: the
// look-back always terminates.
template <int MODE>
__global__ void __launch_bounds__(256) mt_polar_kernel(const PolarArgs a) {
    __shared__ uint32_t wsum[4];
    __shared__ unsigned long long s_excl;
    __shared__ unsigned s_block;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned block = blockIdx.x;
    if (MODE == 1) {
        if (tid == 0) s_block = atomicAdd(a.ticket, 1u);
        __syncthreads();
        block = s_block;
    }
    const long long a0 = a.at_lo + (long long)block * kPolarBlock + (long long)wv * (64 * kPolarSub);
    const uint32_t *const xw = a.xraw + (a.p - a.xbase);
    double ys[kPolarSub], r2s[kPolarSub];
    unsigned long long bal[kPolarSub];
    uint32_t mine = 0, below = 0;
#pragma unroll
    for (int sub = 0; sub < kPolarSub; ++sub) {
        const long long at = a0 + sub * 64 + lane;
        bool ok = false;
        ys[sub] = 0; r2s[sub] = 1;
        if (at < a.at_hi) {
            const uint32_t *w = xw + 4 * at;
            const double u0 = mt_canonical(mt_temper(w[0]), mt_temper(w[1]));
            const double u1 = mt_canonical(mt_temper(w[2]), mt_temper(w[3]));
            const double x = 2.0 * u0 - 1.0;               // random.tcc:1821-1825
            const double y = 2.0 * u1 - 1.0;
            const double r2 = x * x + y * y;
            ok = !(r2 > 1.0 || r2 == 0.0);
            ys[sub] = y; r2s[sub] = r2;
        }
        bal[sub] = __ballot(ok);
        mine += (uint32_t)__popcll(bal[sub]);
        if (MODE == 0) below += (uint32_t)__popcll(__ballot(ok && at < a.split_at));
    }
    if (MODE == 0) {
        if (lane == 0 && mine) {
            if (below) atomicAdd(&a.counters[0], (unsigned long long)below);
            if (mine - below) atomicAdd(&a.counters[1], (unsigned long long)(mine - below));
        }
        return;
    }
    if (lane == 0) wsum[wv] = mine;
    __syncthreads();
    if (wv == 0) {
        const unsigned long long agg = (unsigned long long)wsum[0] + wsum[1] + wsum[2] + wsum[3];
        unsigned long long excl = a.base0;
        if (block > 0) {
            if (lane == 0) __hip_atomic_store(&a.status[block], kStAgg | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long run = 0;
            long long win = (long long)block - 1;   // nearest predecessor of the window; lane i looks at win - i
            for (;;) {
                const long long pred = win - lane;
                // block "-1" knows its prefix: base0; anything in front of it is never reached
                const unsigned long long st = pred >= 0 ? __hip_atomic_load(&a.status[pred], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                        : (kStPrefix | (pred == -1 ? a.base0 : 0ull));
                const unsigned flag = (unsigned)(st >> 62);
                const unsigned long long inval = __ballot(flag == 0u), pref = __ballot(flag == 2u);
                if (pref) {
                    const int fp = __ffsll((long long)pref) - 1;                 // nearest predecessor that knows its prefix
                    if (inval & ((1ull << fp) - 1ull)) { __builtin_amdgcn_s_sleep(1); continue; }   // someone nearer has not published yet
                    run += mt_wave_sum(lane <= fp ? (st & kStMask) : 0ull);
                    break;
                }
                if (inval) { __builtin_amdgcn_s_sleep(1); continue; }
                run += mt_wave_sum(st & kStMask);
                win -= 64;
            }
            excl = run;
        }
        if (lane == 0) {
            __hip_atomic_store(&a.status[block], kStPrefix | (excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_excl = excl;
        }
    }
    __syncthreads();
    if (!a.out) return;
    const unsigned long long excl = s_excl;
    if (excl >= a.need) return;
    // frame and position of the workgroup's first item, by ONE 64-bit division; the items of a workgroup are at most 2048 further on,
    // so theirs follow with 32-bit arithmetic
    unsigned long long f0 = 0;
    uint32_t i0 = 0, c0 = 0;
    const uint32_t pf = (uint32_t)a.per_frame;
    if (pf) {
        f0 = excl / pf;
        i0 = (uint32_t)(excl - f0 * pf);
        if (a.tx) c0 = (uint32_t)(((unsigned long long)a.first_frame + f0) % (unsigned long long)a.ncw);
    }
    uint32_t r0 = 0;   // items of the workgroup in front of this lane's
    for (int v = 0; v < wv; ++v) r0 += wsum[v];
    const double inv_s2_den = a.sigma * a.sigma;
#pragma unroll
    for (int sub = 0; sub < kPolarSub; ++sub) {
        const bool ok = (bal[sub] >> lane) & 1ull;
        const uint32_t r = r0 + (uint32_t)__popcll(bal[sub] & ((1ull << lane) - 1ull));
        r0 += (uint32_t)__popcll(bal[sub]);
        const unsigned long long g = excl + r;
        if (!ok || g >= a.need) continue;
        long long f = 0;
        uint32_t i = 0, df = 0;
        if (pf) {   // a shard only evaluates the samples of its own frames
            const uint32_t t = i0 + r;
            df = t / pf;
            i = t - df * pf;
            f = (long long)(f0 + df);
            if (f < a.row_lo || f >= a.row_hi) continue;
        }
        const double r2 = r2s[sub];
        // random.tcc:1827  mult = sqrt(-2 log(r2) / r2).  0 < r2 <= 1 and r2 >= 2^-106, 0 <= -2 log(r2) < 150: far from every case the
        // scaling / fix-up instructions of the compiler's division exist for, so the shorter sequence returns the same bits
        // (ldpc_spec::div_ranged; a numerator of -0.0, r2 == 1, ends as +0.0 here and there once `ret * 1.0 + 0.0` has been applied)
        const double mult = sqrt(ldpc_spec::div_ranged(-2.0 * ldpc_spec::log_glibc(r2), r2));
        double ret = ys[sub] * mult;                                         // :1830
        ret = ret * 1.0 + 0.0;                                               // :1833 stddev 1, mean 0
        if (!pf) {
            a.out[g] = ret;
        } else {
            double c = 0.0;
            if (a.tx) {
                uint32_t cw = c0 + df;
                cw = cw >= (uint32_t)a.ncw ? cw % (uint32_t)a.ncw : cw;
                c = (double)a.tx[(size_t)cw * a.ntx + i];
            }
            const double v = -2.0 * (a.sigma * ret + 2.0 * c - 1.0) / inv_s2_den;   // bp_simulation.cpp:603 / :610
            const int o = a.scatter ? a.scatter[i] : (int)i;                       // :684
            a.out[(f - a.row_lo) * (long long)pf + o] = o >= a.punct_start ? a.punct_val : v;   // :697-710
        }
    }
}

// After the fused pass: where does the generator continue?  Item `limit` - 1 is the last one the round keeps (limit = min(accepted,
// need), whole frames only in frame mode; the host passes it when several GPUs share the round, else it is computed here from this
// launch's own total).  The workgroup that holds that item is found by bisection over the inclusive prefixes the look-back left in
// status[]; its attempts are tested once more to locate the attempt itself; the 624 words behind its draws become the state.
struct FinishArgs {
    PolarArgs pa;
    long long blocks;
    long long limit_in;        // -1: compute from this launch's total
    long long end_preset;      // tape index where the generator stays if nothing is kept
    long long xwords;          // words in the window (bounds check for the adopt)
    unsigned long long *total; // [0] accepted attempts in [at_lo, at_hi); [1] limit used
    long long *end_t;          // [0] tape index of the next unread word; [1] 1 if this launch found it (state adopted)
    uint32_t *state;           // [624] out
};

__global__ void __launch_bounds__(256) mt_finish_kernel(const FinishArgs f) {
    __shared__ uint32_t wsum[4];
    __shared__ long long s_end;
    const PolarArgs &a = f.pa;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    auto incl = [&](long long b) -> unsigned long long { return b < 0 ? a.base0 : (a.status[b] & kStMask); };
    const unsigned long long last = incl(f.blocks - 1);   // items in front of the end of this range
    unsigned long long limit;
    if (f.limit_in >= 0) limit = (unsigned long long)f.limit_in;
    else {
        limit = last < a.need ? last : a.need;
        if (a.per_frame) limit -= limit % (unsigned long long)a.per_frame;
    }
    if (tid == 0) { f.total[0] = last - a.base0; f.total[1] = limit; s_end = -1; }
    __syncthreads();
    if (limit == 0) {
        if (a.at_lo == 0 && tid == 0) s_end = f.end_preset;   // nothing kept: the range that starts the round leaves the generator where it was
    } else if (limit - 1 >= a.base0 && limit - 1 < last) {
        long long lo = 0, hi = f.blocks - 1;                  // smallest block whose inclusive prefix reaches `limit`
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (incl(mid) >= limit) hi = mid; else lo = mid + 1;
        }
        const unsigned long long rank = limit - 1 - incl(lo - 1);   // which accepted attempt of that block (0-based)
        const long long a0 = a.at_lo + lo * kPolarBlock + (long long)wv * (64 * kPolarSub);
        const uint32_t *const xw = a.xraw + (a.p - a.xbase);
        unsigned long long bal[kPolarSub];
        uint32_t mine = 0;
#pragma unroll
        for (int sub = 0; sub < kPolarSub; ++sub) {
            const long long at = a0 + sub * 64 + lane;
            bool ok = false;
            if (at < a.at_hi) {
                const uint32_t *w = xw + 4 * at;
                const double u0 = mt_canonical(mt_temper(w[0]), mt_temper(w[1]));
                const double u1 = mt_canonical(mt_temper(w[2]), mt_temper(w[3]));
                const double x = 2.0 * u0 - 1.0, y = 2.0 * u1 - 1.0;
                const double r2 = x * x + y * y;
                ok = !(r2 > 1.0 || r2 == 0.0);
            }
            bal[sub] = __ballot(ok);
            mine += (uint32_t)__popcll(bal[sub]);
        }
        if (lane == 0) wsum[wv] = mine;
        __syncthreads();
        unsigned long long g0 = 0;
        for (int v = 0; v < wv; ++v) g0 += wsum[v];
#pragma unroll
        for (int sub = 0; sub < kPolarSub; ++sub) {
            const bool ok = (bal[sub] >> lane) & 1ull;
            const unsigned long long g = g0 + (unsigned long long)__popcll(bal[sub] & ((1ull << lane) - 1ull));
            g0 += (unsigned long long)__popcll(bal[sub]);
            if (ok && g == rank) s_end = a.p + 4 * (a0 + sub * 64 + lane + 1);   // the word after this attempt's four
        }
    }
    __syncthreads();
    const long long end = s_end;
    const bool found = end >= 0 && end - a.xbase >= 0 && end - a.xbase + MTN <= f.xwords;
    if (found)
        for (int i = tid; i < MTN; i += 256) f.state[i] = a.xraw[end - a.xbase + i];   // the generator continues there, position 0
    if (tid == 0) { f.end_t[0] = end; f.end_t[1] = found ? 1 : 0; }
}

// per-context device state of the generator (ldpc_hip_mt_* entry points)
struct DeviceState {
    bool set = false;
    int pos = 0;                         // next word of d_state to draw (0..624); 0 after every generation round
    uint32_t *d_state = nullptr;         // [624]
    uint32_t *d_state_next = nullptr;    // [624] where the finish kernel leaves the state the round ends in
    uint32_t *d_bits = nullptr;          // [kLevels][kMaxBits] exponents of the jump polynomials
    uint32_t *d_states = nullptr;        // [cap_streams][624]
    int cap_streams = 0;
    uint32_t *d_xraw = nullptr;          // [cap_words]
    size_t cap_words = 0;
    unsigned long long *d_status = nullptr;   // [cap_blocks] look-back status words of the fused pass
    long long cap_blocks = 0;
    unsigned *d_ticket = nullptr;
    unsigned long long *d_counters = nullptr; // [2] count-only mode
    unsigned long long *d_total = nullptr;    // [2] accepted in range, limit used
    long long *d_end_t = nullptr;             // [2] next unread word, found flag
    long long frames_taken = 0;          // frames drawn since ldpc_hip_mt_set_state (codeword choice f % ncw)
    int32_t *d_info = nullptr, *d_iters = nullptr;
    long long cap_rec = 0;
};

inline void release(DeviceState &m) {
    void *ptrs[] = {m.d_state, m.d_state_next, m.d_bits, m.d_states, m.d_xraw, m.d_status, m.d_ticket, m.d_counters, m.d_total, m.d_end_t, m.d_info, m.d_iters};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    m = DeviceState();
}

}  // namespace ldpc_mt

// ---- host-side descriptions of a generation round (used by ldpc_mt_api.hpp, ldpc_multi.hpp and, as a member, by ldpc_hip_ctx)
namespace {

// The round as a whole: how many attempts are on offer and how far the tape reaches (the same for every shard count).
struct MtPlan {
    unsigned long long need = 0;   // items wanted
    long long pos = 0;             // tape index of the first unread word (0..624)
    long long attempts = 0;        // attempts on offer: tape words pos + 4a .. pos + 4a + 3, a < attempts
    long long tape_words = 0;      // generated words behind the first 624 (the last 624 of them stay unread: they can become the next state)
    long long margin = 0;          // attempts by which the position of an item may be off its expectation (8 standard deviations + slack)
};

// A context's window of the tape: whole sub-streams of 2^ls words on the grid that starts at tape word 624.
struct MtWindow {
    int ls = ldpc_mt::kLog2Stride;
    long long stride = 0, first = 0, S = 0;   // streams [first, first + S)
    long long xbase = 0, xwords = 0;          // tape index of xraw[0], words in the window
    long long gen_words = 0;                  // tape words behind the first 624 the window's streams produce (a multiple of 64)
    long long at_lo = 0, at_hi = 0;           // attempts whose four words all lie inside the window
};

// ---- one round of the shared-out generator with the exchange done by the CALLER (one process per GPU; ldpc_multi.hpp does the same
// over the shards of one process).  State between the three calls:
struct MtShardRound {
    bool open = false;
    MtPlan pl;
    MtWindow w;
    ldpc_mt::PolarArgs proto;
    long long frames = 0, row_lo = 0, row_hi = 0, cut_lo = 0, cut_hi = 0;
    unsigned long long left = 0, own = 0, base0 = 0, limit = 0;
    int rank = 0, n = 1;
};

}  // namespace
