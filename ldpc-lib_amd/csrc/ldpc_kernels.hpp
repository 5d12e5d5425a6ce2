// ldpc_kernels.hpp -- hand-written gfx950 (CDNA4) kernels for the QC-LDPC belief-propagation hot path.
//
// Work decomposition (all decoders):
//   * a codeword ("frame") never leaves the CU: its whole message-passing state is on chip for all
//     iterations.  HBM carries only the compulsory traffic (LLR in, packed hard bits / iteration count out).
//   * one lane = one check row n of every circulant (the "check lane" view); the same lane is also
//     variable i = n of every block column (the "variable lane" view).  With lifting M = 64 a frame is exactly
//     one wavefront.  M < 64 packs F = floor(64/M) frames into one wavefront; 64 < M <= 512 spreads one frame
//     over ceil(M/64) wavefronts of one workgroup (barriers between dependent phases).
//   * a-posteriori values `soft[N]` live in LDS (fp64, 8 B per variable).  A circulant shift c is an indexed LDS
//     access at ((n+c) mod M): consecutive lanes -> consecutive 8-byte words -> bank-conflict free, and it
//     replaces the two memcpy rotations per edge of the CPU code (decoders.cpp:327-346).
//   * per-check records {min1,min2,pos,sign} and per-edge sign bits never touch memory: they are VGPRs of
//     the check lane (min1/min2 fp64, pos + the row's edge-sign bits packed in one dword per block row).
//     Because a lane owns a whole check row, min1/min2 is a serial in-register scan over the row's <= 16
//     circulants: no cross-lane reduction is needed at all; the only cross-lane traffic is the rotation
//     (LDS) and the syndrome vote (ballot).
//   * arithmetic is IEEE fp64 in the reference's exact operation order with contraction OFF
//     (-ffp-contract=off): the hard decisions are bit-identical to the CPU code, not approximately equal.
//
// Sign tests use the sign BIT of the high dword.  That equals the reference's `x < 0` for every value except
// -0.0 (and NaN); inputs are canonicalised (y + 0.0) on load, after which no intermediate of the algorithms
// can be -0.0 (x - y is -0.0 only for x = -0.0, y = +0.0; x + y only for x = y = -0.0), and the sign of a
// zero never influences a non-zero value or a comparison.  Inputs must be finite.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ldpc {

constexpr double kMaxVal = 32767.0;  // decoders.cpp:4299-4301 MAX_VAL = (1L<<15)-1
constexpr int kRowBits = 16;         // edge-sign bits per block row kept in the low half of `meta`

struct DecArgs {
    const double *llr;        // [B][N]
    uint32_t *hard;           // [B][hard_words] or null
    int32_t *iters;           // [B] or null
    double *soft_out;         // [B][N] or null
    const int32_t *row_start; // [rh+1]
    const uint32_t *edges;    // [ne]  (block column << 16) | shift, row-major order (rows asc, columns asc)
    const int32_t *col_start; // [nh+1]           (sum-product only)
    const uint32_t *col_edges;// [ne] (block row << 16) | shift, column-major order (columns asc, rows asc)
    const uint32_t *col_slot; // [ne] for the q-th entry of a column: row-major id of that edge
    const uint32_t *edge_row; // [ne] block row of each row-major edge
    long long B;
    int rh, nh, M, N, F, maxiter, hard_words;
    double alpha;
    double ims_thr;           // integer min-sum quantiser (decoders.h:46-48): threshold, bits of the input, bits of the data path
    int ims_qbits, ims_dbits;
};

__device__ __forceinline__ uint32_t hi32(double x) { return (uint32_t)__double2hiint(x); }
__device__ __forceinline__ uint32_t lo32(double x) { return (uint32_t)__double2loint(x); }
__device__ __forceinline__ double mkdouble(uint32_t hi, uint32_t lo) { return __hiloint2double((int)hi, (int)lo); }
// x with its sign flipped when bit0 of s is set
__device__ __forceinline__ double flip_if(double x, uint32_t s) { return mkdouble(hi32(x) ^ (s << 31), lo32(x)); }

// Lane -> (check row n, frame slot f).  Single-wave kernels interleave the F frames of a wave across lanes
// (lane = n*F + f) so that the LDS image ((k*M+i)*F + f) is hit with consecutive addresses by consecutive lanes.
// Lanes beyond the last check row of the wave/workgroup are "invalid": they compute on row 0 (so every address
// stays in range) but never store and never vote.
template <bool MW>
__device__ __forceinline__ bool lane_map(int F, int M, int &n, int &f) {
    if (MW) { n = threadIdx.x; f = 0; }
    else    { n = threadIdx.x / F; f = threadIdx.x - n * F; }
    const bool valid = n < M;
    if (!valid) n = 0;
    return valid;
}

// bit q*F set for every q: the lanes of frame slot 0 in a single-wave kernel (slot f = same mask << f)
__device__ __forceinline__ unsigned long long slot_mask(int F) {
    unsigned long long per = 0ull;
    for (int q = 0; q < 64; q += F) per |= 1ull << q;
    return per;
}

__device__ __forceinline__ int rot_idx(int n, int c, int M) {
    int t = n + c;
    return t >= M ? t - M : t;
}

// "Does any check of MY frame fail?"  fail/valid are per lane.  Returns a per-lane bool (uniform per frame).
template <bool MW>
__device__ __forceinline__ bool frame_vote(bool fail, int F, int f, unsigned long long per, int *sh_flag) {
    if (MW) {
        // several waves, one frame: OR through LDS.  sh_flag is reset by the barrier protocol below.
        if (threadIdx.x == 0) *sh_flag = 0;
        __syncthreads();
        if (__any(fail) && (threadIdx.x & 63) == 0) atomicOr(sh_flag, 1);
        __syncthreads();
        bool r = *sh_flag != 0;
        __syncthreads();
        return r;
    } else {
        const unsigned long long b = __ballot(fail);
        if (F == 1) return b != 0ull;
        return ((b >> f) & per) != 0ull;  // lanes of frame slot f are f, f+F, f+2F, ...
    }
}

// Pack the hard decisions of a frame from the sign bits of the LDS-resident soft values and write the
// frame's outputs.  Called by all lanes of the frame.
template <bool MW>
__device__ __forceinline__ void write_outputs(const DecArgs &a, const double *lds, long long fr, int n, int f,
                                               bool live, int res, double thr_is_one) {
    const int M = a.M, F = a.F, N = a.N;
    if (!live) return;
    if (n == 0 && a.iters) a.iters[fr] = res;
    if (a.hard) {
        for (int w = n; w < a.hard_words; w += M) {
            uint32_t bits = 0;
            for (int b = 0; b < 32; ++b) {
                const int v = 32 * w + b;
                if (v < N) {
                    const double s = lds[v * F + f];
                    // MS/LMS: soft < 0 (sign bit, soft is never -0.0); SP: soft < 1.0
                    const uint32_t bit = thr_is_one != 0.0 ? (uint32_t)(s < 1.0) : (hi32(s) >> 31);
                    bits |= bit << b;
                }
            }
            a.hard[fr * a.hard_words + w] = bits;
        }
    }
    if (a.soft_out) {
        for (int k = 0; k < a.nh; ++k) a.soft_out[fr * N + k * M + n] = lds[(k * M + n) * F + f];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Flooding normalised min-sum  (min_sum_decod_qc_lm, decoders.cpp:4554-4767; semantics SURVEY Appendix A.1)
//
// per iteration:  STATE1  acc[v] = sum_j c2v(j,v)   (check lanes scatter-add into LDS in ascending block row)
//                 STATE2  soft = y + acc*alpha       (variable lanes, y in VGPRs, two roundings)
//                 STATE3  new records per check row from v2c = soft - alpha*c2v_old ; syndrome = xor of signs
// Algorithmic state traffic that the CPU code moves through memory every iteration (4.25*E + 30*R + 8.125*N
// bytes at 4-byte LLR width) stays in VGPRs/LDS here.
// ---------------------------------------------------------------------------------------------------------
template <int RHM, int NHM, bool MW>
__global__ void __launch_bounds__(MW ? 512 : 64) ms_flood_kernel(const DecArgs a) {
    extern __shared__ double lds[];  // [N][F] soft / acc, then one flag word (kept behind, so the base stays aligned)
    const int M = a.M, F = a.F, N = a.N, rh = a.rh, nh = a.nh;
    int *const sh_flag = (int *)(lds + (size_t)N * F);
    const double alpha = a.alpha;
    int n, f;
    const bool valid = lane_map<MW>(F, M, n, f);
    const unsigned long long per = MW ? 0ull : slot_mask(F);
    const long long fr = (long long)blockIdx.x * F + f;
    const bool inb = fr < a.B;          // uniform per frame (and per workgroup when MW)
    const bool live = valid && inb;

    double y[NHM];
#pragma unroll
    for (int k = 0; k < NHM; ++k) y[k] = (k < nh && live) ? a.llr[fr * N + k * M + n] + 0.0 : 0.0;

    double m1[RHM], m2[RHM];
    uint32_t meta[RHM];  // [15:0] v2c sign bit of the row's idx-th edge, [23:16] idx of the min1 edge
#pragma unroll
    for (int j = 0; j < RHM; ++j) { m1[j] = 0.0; m2[j] = 0.0; meta[j] = 0u; }  // :4579-4596

    bool done = !inb;   // per FRAME: every lane of a frame (valid or not) carries the same value
    int res = -a.maxiter;

    for (int iter = 0; iter < a.maxiter; ++iter) {
        const bool wr = !done && valid;  // this lane may store
        // ---- STATE1 (:4633-4667)
        if (wr) {
#pragma unroll
            for (int k = 0; k < NHM; ++k) if (k < nh) lds[(k * M + n) * F + f] = 0.0;
        }
        if (MW) __syncthreads();
#pragma unroll
        for (int j = 0; j < RHM; ++j) {
            if (j < rh) {
                const int e0 = a.row_start[j], rw = a.row_start[j + 1] - e0;
                const uint32_t mt = meta[j];
                const uint32_t par = __popc(mt & 0xffffu) & 1u;  // row sign = xor of the row's edge signs
                const uint32_t pos = mt >> kRowBits;
                for (int idx = 0; idx < rw; ++idx) {
                    const uint32_t d = a.edges[e0 + idx];
                    const int k = d >> 16, c = d & 0xffffu;
                    const int addr = (k * M + rot_idx(n, c, M)) * F + f;
                    const double aa = (pos == (uint32_t)idx) ? m2[j] : m1[j];
                    const double cv = flip_if(aa, ((mt >> idx) ^ par) & 1u);
                    if (wr) lds[addr] = lds[addr] + cv;
                }
                if (MW) __syncthreads();  // the next block row adds into the same variables
            }
        }
        // ---- STATE2 (:4670-4685): multiply, then add -- two roundings (no FMA)
        if (wr) {
#pragma unroll
            for (int k = 0; k < NHM; ++k) {
                if (k < nh) {
                    const int o = (k * M + n) * F + f;
                    const double p = lds[o] * alpha;
                    lds[o] = y[k] + p;
                }
            }
        }
        if (MW) __syncthreads();
        // ---- STATE3 (:4690-4755)
        uint32_t failw = 0;
#pragma unroll
        for (int j = 0; j < RHM; ++j) {
            if (j < rh) {
                const int e0 = a.row_start[j], rw = a.row_start[j + 1] - e0;
                const uint32_t mt = meta[j];
                const uint32_t par = __popc(mt & 0xffffu) & 1u;
                const uint32_t pos = mt >> kRowBits;
                const double a1 = m1[j] * alpha, a2 = m2[j] * alpha;
                double nm1 = kMaxVal, nm2 = kMaxVal;
                uint32_t npos = 0, nS = 0, sy = 0;
                for (int idx = 0; idx < rw; ++idx) {
                    const uint32_t d = a.edges[e0 + idx];
                    const int k = d >> 16, c = d & 0xffffu;
                    const int addr = (k * M + rot_idx(n, c, M)) * F + f;
                    const double r = lds[addr];
                    sy ^= hi32(r);
                    const double aa = (pos == (uint32_t)idx) ? a2 : a1;
                    const double x = flip_if(aa, ((mt >> idx) ^ par) & 1u);
                    const double t = r - x;                       // v2c
                    nS |= (hi32(t) >> 31) << idx;
                    const double v = fmin(fabs(t), kMaxVal);      // :4729-4730
                    const bool c1 = v < nm1;                      // strict: first minimum keeps the position
                    nm2 = fmin(fmax(v, nm1), nm2);                // = c1 ? nm1 : min(v, nm2)
                    npos = c1 ? (uint32_t)idx : npos;
                    nm1 = fmin(v, nm1);
                }
                failw |= sy;
                if (!done) { m1[j] = nm1; m2[j] = nm2; meta[j] = nS | (npos << kRowBits); }
            }
        }
        const bool fail = valid && (failw >> 31);
        const bool frame_fail = frame_vote<MW>(fail, F, f, per, sh_flag);
        if (!done && !frame_fail) { done = true; res = iter + 1; }  // :4761-4766
        if (MW) { if (done) break; }
        else if (__all(done)) break;
    }
    write_outputs<MW>(a, lds, fr, n, f, live, res, 0.0);
}

// ---------------------------------------------------------------------------------------------------------
// Layered offset min-sum  (lmin_sum_decod_qc_lm, decoders.cpp:5064-5425, active branch :5106-5290 with
// MY_VERSION; semantics SURVEY Appendix A.3).  Block rows (layers) are strictly sequential; inside a layer
// every variable is touched by exactly one check, so the layer needs no synchronisation of its own.
// RWM = compile-time bound on the row weight: the v2c values of a layer stay in VGPRs between its two passes.
// ---------------------------------------------------------------------------------------------------------
template <int RHM, int RWM, bool MW>
__global__ void __launch_bounds__(MW ? 512 : 64) lms_layered_kernel(const DecArgs a) {
    extern __shared__ double lds[];
    const int M = a.M, F = a.F, N = a.N, rh = a.rh, nh = a.nh;
    int *const sh_flag = (int *)(lds + (size_t)N * F);
    const double beta = 0.4;  // :5163 (the alpha/beta arguments are dead upstream)
    int n, f;
    const bool valid = lane_map<MW>(F, M, n, f);
    const unsigned long long per = MW ? 0ull : slot_mask(F);
    const long long fr = (long long)blockIdx.x * F + f;
    const bool inb = fr < a.B;
    const bool live = valid && inb;

    if (valid) {
        for (int k = 0; k < nh; ++k)  // :5088 soft = y
            lds[(k * M + n) * F + f] = live ? a.llr[fr * N + k * M + n] + 0.0 : 0.0;
    }
    double m1[RHM], m2[RHM];
    uint32_t meta[RHM];
#pragma unroll
    for (int j = 0; j < RHM; ++j) { m1[j] = 0.0; m2[j] = 0.0; meta[j] = 0u; }
    if (MW) __syncthreads();

    // syndrome of the current soft values (check_syndrome, decoders.cpp:793-814)
    auto syndrome_fail = [&]() -> bool {
        uint32_t failw = 0;
        for (int j = 0; j < rh; ++j) {
            const int e0 = a.row_start[j], e1 = a.row_start[j + 1];
            uint32_t sy = 0;
            for (int e = e0; e < e1; ++e) {
                const uint32_t d = a.edges[e];
                const int k = d >> 16, c = d & 0xffffu;
                sy ^= hi32(lds[(k * M + rot_idx(n, c, M)) * F + f]);
            }
            failw |= sy;
        }
        return valid && (failw >> 31);
    };

    bool done = !inb;
    int res = -a.maxiter;                                              // :5424 when the loop runs dry
    bool frame_fail = frame_vote<MW>(syndrome_fail(), F, f, per, sh_flag);  // :5111-5115
    if (!done && !frame_fail) { done = true; res = 1; }                // :5119 at iter 0 -> returns 0+1
    for (int iter = 0; iter < a.maxiter; ++iter) {
        if (MW) { if (done) break; }
        else if (__all(done)) break;
        const bool wr = !done && valid;
#pragma unroll
        for (int j = 0; j < RHM; ++j) {
            if (j < rh) {
                const int e0 = a.row_start[j], rw = a.row_start[j + 1] - e0;
                const uint32_t mt = meta[j];
                const uint32_t par = __popc(mt & 0xffffu) & 1u;
                const uint32_t pos = mt >> kRowBits;
                double nm1 = kMaxVal, nm2 = kMaxVal;  // :5133-5134
                uint32_t npos = 0, nS = 0;
                double tv[RWM];
                int addr[RWM];
#pragma unroll
                for (int idx = 0; idx < RWM; ++idx) {
                    if (idx < rw) {                     // :5141-5177
                        const uint32_t d = a.edges[e0 + idx];
                        const int k = d >> 16, c = d & 0xffffu;
                        addr[idx] = (k * M + rot_idx(n, c, M)) * F + f;
                        const double r = lds[addr[idx]];
                        const double aa = (pos == (uint32_t)idx) ? m2[j] : m1[j];
                        const double pc = flip_if(aa, ((mt >> idx) ^ par) & 1u);
                        const double t = r - pc;
                        tv[idx] = t;
                        nS |= (hi32(t) >> 31) << idx;   // sign kept even when the magnitude clips to 0 (:5164-5168)
                        double mag = fabs(t) - beta;
                        mag = mag < 0 ? 0 : mag;
                        const bool c1 = mag < nm1;      // process_check_node :5012-5027
                        nm2 = fmin(fmax(mag, nm1), nm2);
                        npos = c1 ? (uint32_t)idx : npos;
                        nm1 = fmin(mag, nm1);
                    }
                }
                const uint32_t npar = __popc(nS) & 1u;
#pragma unroll
                for (int idx = 0; idx < RWM; ++idx) {
                    if (idx < rw) {                     // :5182-5206
                        const double aa = (npos == (uint32_t)idx) ? nm2 : nm1;
                        const double cv = flip_if(aa, ((nS >> idx) ^ npar) & 1u);
                        if (wr) lds[addr[idx]] = tv[idx] + cv;
                    }
                }
                if (!done) { m1[j] = nm1; m2[j] = nm2; meta[j] = nS | (npos << kRowBits); }
                if (MW) __syncthreads();  // the next layer reads what this one wrote
            }
        }
        frame_fail = frame_vote<MW>(syndrome_fail(), F, f, per, sh_flag);  // :5281-5284
        if (!done && !frame_fail) { done = true; res = iter + 1; }          // :5287, returns iter+1
    }
    write_outputs<MW>(a, lds, fr, n, f, live, res, 0.0);
}

// ---------------------------------------------------------------------------------------------------------
// Integer min-sum (imin_sum_decod_qc_lm, decoders.cpp:5430-5690; semantics SURVEY Appendix A.4): the flooding
// schedule of ms_flood_kernel on int16 values.  Input quantiser: coef = sqrt(N / sum y^2) with the sum taken in
// index order (a parallel reduction would round differently and could move a value across a quantiser boundary),
// q = floor(min(|y|*coef, thr)*max_quant/thr + 0.5) with the sign of y; c2v magnitudes are scaled (a*ialpha)>>4 in
// STATE1 and STATE3, STATE1 saturates at +-max_data after every add, STATE2 is sat(iy + acc) (no alpha).  Everything
// after the quantiser is integer arithmetic, so the GPU result is exact by construction; values never leave
// [-2*max_data, 2*max_data], so 32-bit lanes reproduce the reference's int16 arithmetic without wrap-around.
// LDS: one int32 per variable.
// ---------------------------------------------------------------------------------------------------------
template <int RHM, int NHM, bool MW>
__global__ void __launch_bounds__(MW ? 512 : 64) ims_flood_kernel(const DecArgs a) {
    extern __shared__ double lds_raw[];
    int *const lds = reinterpret_cast<int *>(lds_raw);  // [N][F] soft / acc
    const int M = a.M, F = a.F, N = a.N, rh = a.rh, nh = a.nh;
    int *const sh_flag = lds + (size_t)N * F;
    const int max_data = (1 << (a.ims_dbits - 1)) - 1;   // :5445
    const int max_quant = (1 << (a.ims_qbits - 1)) - 1;  // :5446
    const int ialpha = (int)(a.alpha * (1 << 4));         // :5458 MS_ALPHA_FPP = 4
    int n, f;
    const bool valid = lane_map<MW>(F, M, n, f);
    const unsigned long long per = MW ? 0ull : slot_mask(F);
    const long long fr = (long long)blockIdx.x * F + f;
    const bool inb = fr < a.B;
    const bool live = valid && inb;
    auto sat = [&](int x) { return x > max_data ? max_data : (x < -max_data ? -max_data : x); };  // limit_val :4308

    // :5472-5500 energy-normalised quantiser.  Every lane of a frame walks the frame's LLRs in index order (same
    // addresses within the frame: broadcast loads), so all of them hold the identical, sequentially rounded sum.
    int iy[NHM];
    {
        double en = 0;
        const double *yf = a.llr + (inb ? fr : 0) * (long long)N;
        for (int i = 0; i < N; ++i) { const double v = yf[i]; en += v * v; }
        const double coef = sqrt((double)N / en);
#pragma unroll
        for (int k = 0; k < NHM; ++k) {
            iy[k] = 0;
            if (k < nh && live) {
                double val = yf[k * M + n];
                int sign = 0;
                if (val < 0) { val = -val; sign = 1; }
                val *= coef;
                if (val > a.ims_thr) val = a.ims_thr;
                const int ival = (int)(short)floor(val * max_quant / a.ims_thr + 0.5);
                iy[k] = sign ? -ival : ival;
            }
        }
    }
    int m1[RHM], m2[RHM];
    uint32_t meta[RHM];
#pragma unroll
    for (int j = 0; j < RHM; ++j) { m1[j] = 0; m2[j] = 0; meta[j] = 0u; }

    bool done = !inb;
    int res = -a.maxiter;
    for (int iter = 0; iter < a.maxiter; ++iter) {
        const bool wr = !done && valid;
        if (wr) {
#pragma unroll
            for (int k = 0; k < NHM; ++k) if (k < nh) lds[(k * M + n) * F + f] = 0;
        }
        if (MW) __syncthreads();
#pragma unroll
        for (int j = 0; j < RHM; ++j) {   // STATE1 :5540-5576
            if (j < rh) {
                const int e0 = a.row_start[j], rw = a.row_start[j + 1] - e0;
                const uint32_t mt = meta[j];
                const uint32_t par = __popc(mt & 0xffffu) & 1u;
                const uint32_t pos = mt >> kRowBits;
                for (int idx = 0; idx < rw; ++idx) {
                    const uint32_t d = a.edges[e0 + idx];
                    const int k = d >> 16, c = d & 0xffffu;
                    const int addr = (k * M + rot_idx(n, c, M)) * F + f;
                    int tmp = (pos == (uint32_t)idx) ? m2[j] : m1[j];
                    tmp = (tmp * ialpha) >> 4;
                    const int cv = (((mt >> idx) ^ par) & 1u) ? -tmp : tmp;
                    if (wr) lds[addr] = sat(lds[addr] + cv);
                }
                if (MW) __syncthreads();
            }
        }
        if (wr) {                          // STATE2 :5579-5604
#pragma unroll
            for (int k = 0; k < NHM; ++k) {
                if (k < nh) {
                    const int o = (k * M + n) * F + f;
                    lds[o] = sat(iy[k] + lds[o]);
                }
            }
        }
        if (MW) __syncthreads();
        uint32_t failw = 0;
#pragma unroll
        for (int j = 0; j < RHM; ++j) {   // STATE3 :5610-5678
            if (j < rh) {
                const int e0 = a.row_start[j], rw = a.row_start[j + 1] - e0;
                const uint32_t mt = meta[j];
                const uint32_t par = __popc(mt & 0xffffu) & 1u;
                const uint32_t pos = mt >> kRowBits;
                int nm1 = max_data, nm2 = max_data;
                uint32_t npos = 0, nS = 0, sy = 0;
                for (int idx = 0; idx < rw; ++idx) {
                    const uint32_t d = a.edges[e0 + idx];
                    const int k = d >> 16, c = d & 0xffffu;
                    const int r = lds[(k * M + rot_idx(n, c, M)) * F + f];
                    sy ^= (uint32_t)r;
                    const int aa = (pos == (uint32_t)idx) ? m2[j] : m1[j];
                    const int val = (aa * ialpha) >> 4;
                    const int t = (((mt >> idx) ^ par) & 1u) ? -val : val;
                    const int msg = r - t;
                    nS |= ((uint32_t)msg >> 31) << idx;
                    int v = msg < 0 ? -msg : msg;
                    v = v > max_data ? max_data : v;
                    if (v < nm1) { npos = idx; nm2 = nm1; nm1 = v; }
                    else if (v < nm2) nm2 = v;
                }
                failw |= sy;
                if (!done) { m1[j] = nm1; m2[j] = nm2; meta[j] = nS | (npos << kRowBits); }
            }
        }
        const bool fail = valid && (failw >> 31);
        const bool frame_fail = frame_vote<MW>(fail, F, f, per, sh_flag);
        if (!done && !frame_fail) { done = true; res = iter + 1; }  // :5684-5689
        if (MW) { if (done) break; }
        else if (__all(done)) break;
    }
    if (!live) return;
    if (n == 0 && a.iters) a.iters[fr] = res;
    if (a.hard) {
        for (int w = n; w < a.hard_words; w += M) {
            uint32_t bits = 0;
            for (int b = 0; b < 32; ++b) {
                const int v = 32 * w + b;
                if (v < N) bits |= ((uint32_t)lds[v * F + f] >> 31) << b;
            }
            a.hard[fr * a.hard_words + w] = bits;
        }
    }
    if (a.soft_out)
        for (int k = 0; k < nh; ++k) a.soft_out[fr * N + k * M + n] = (double)lds[(k * M + n) * F + f];
}

}  // namespace ldpc
