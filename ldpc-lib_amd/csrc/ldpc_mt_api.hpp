// ldpc_mt_api.hpp -- host side of the exact-replay noise generator (kernels: ldpc_mt.hpp); included at the end of ldpc_hip.hip.
#pragma once

namespace {

int mt_prepare(ldpc_hip_ctx *c) {
    ldpc_mt::DeviceState &m = c->mt;
    if (m.d_bits) return 0;
    const ldpc_mt::JumpPolys &J = ldpc_mt::jump_polys();
    if (!J.ok) return fail(LDPC_HIP_EUNSUPPORTED, "mt19937 jump polynomials: %s", J.err.c_str());
    HIP_TRY(hipMalloc(&m.d_state, sizeof(uint32_t) * ldpc_mt::MTN));
    HIP_TRY(hipMalloc(&m.d_state_next, sizeof(uint32_t) * ldpc_mt::MTN));
    HIP_TRY(hipMalloc(&m.d_total, sizeof(unsigned long long) * 2));
    HIP_TRY(hipMalloc(&m.d_end_t, sizeof(long long) * 2));
    HIP_TRY(hipMalloc(&m.d_counters, sizeof(unsigned long long) * 2));
    HIP_TRY(hipMalloc(&m.d_ticket, sizeof(unsigned) * 2));
    HIP_TRY(hipMalloc(&m.d_bits, sizeof(uint32_t) * (J.bits.size() + ldpc_mt::kBitsPad)));
    HIP_TRY(hipMemset(m.d_bits, 0, sizeof(uint32_t) * (J.bits.size() + ldpc_mt::kBitsPad)));
    HIP_TRY(hipMemcpy(m.d_bits, J.bits.data(), sizeof(uint32_t) * J.bits.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(ldpc_mt::mt_jump_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(sizeof(uint32_t) * ldpc_mt::kSeqWords)));
    return 0;
}

// ---- one generation round, in pieces (a single context runs them back to back; ldpc_multi.hpp runs them per shard with the
// exchange of the accepted-attempt counts in between) -------------------------------------------------------------------------

inline int mt_log2_stride(long long words) {
    // stream length by the size of the job: the generate kernel is bound by the latency of one wave walking its stream, the jumps
    // by their number -- short streams for small rounds, 2^20 words for the 65536-frame batches
    // 2^21-word streams halve the jumps of the largest rounds (0.90 -> 0.55 ms per 2^27 samples) but the word kernel then has too few
    // workgroups to cover its barrier latency (0.69 -> 1.16 ms): 2^20 stays the longest (LDPC_HIP_MT_LOG2_STRIDE: experiments)
    static const int cap = getenv("LDPC_HIP_MT_LOG2_STRIDE") ? atoi(getenv("LDPC_HIP_MT_LOG2_STRIDE")) : 20;
    const int ls = words <= (1ll << 26) ? 18 : words <= (1ll << 28) ? 19 : words <= (1ll << 29) ? 20 : 21;
    return ls < cap ? ls : (cap < 18 ? 18 : cap);
}

MtPlan mt_plan(long long pos, unsigned long long need) {
    using namespace ldpc_mt;
    MtPlan pl;
    pl.need = need; pl.pos = pos;
    const double g = (double)need;
    // attempts for `need` accepted ones at p = pi/4, plus ~8 standard deviations (negative binomial: sd = 0.59 sqrt(need))
    const long long A = (long long)(g * 1.2732395447351628 + 5.0 * std::sqrt(g) + 64.0);
    const long long want = pos + 4 * A;
    const long long stride = 1ll << mt_log2_stride(want);
    long long S = (want + stride - 1) / stride;
    if (S < 1) S = 1;
    if (S > kMaxStreams) S = kMaxStreams;
    long long attempts = (S * stride - pos) / 4;   // the last 624 words stay unread: they are the next state
    if (attempts > A) attempts = A;               // a short round neither generates nor scans a whole stream
    pl.attempts = attempts;
    pl.tape_words = S * stride;
    pl.margin = (long long)(4.8 * std::sqrt(g)) + 64;
    return pl;
}

// the window that covers tape words [w_lo, w_hi) (w_hi <= 624 + tape_words) on a grid chosen by the window's own size
MtWindow mt_window(const MtPlan &pl, long long w_lo, long long w_hi) {
    using namespace ldpc_mt;
    MtWindow w;
    if (w_lo < 0) w_lo = 0;
    if (w_hi > MTN + pl.tape_words) w_hi = MTN + pl.tape_words;
    w.ls = mt_log2_stride(w_hi - w_lo);
    w.stride = 1ll << w.ls;
    w.first = w_lo <= MTN ? 0 : (w_lo - MTN) / w.stride;
    long long last = w_hi <= MTN ? 1 : (w_hi - MTN + w.stride - 1) / w.stride;   // exclusive
    if (last <= w.first) last = w.first + 1;
    w.S = last - w.first;
    w.xbase = w.first * w.stride;
    long long gw = (w_hi - MTN + 63) / 64 * 64;                 // tape words behind the first 624 that are read or adopted
    if (gw > last * w.stride) gw = last * w.stride;
    if (gw > pl.tape_words) gw = pl.tape_words;
    if (gw < 0) gw = 0;
    w.gen_words = gw;
    w.xwords = MTN + gw - w.xbase;
    // attempts entirely inside [window start, 624 + gen_words)
    const long long t_lo = w.first == 0 ? 0 : MTN + w.xbase;    // first tape word the window holds
    w.at_lo = t_lo <= pl.pos ? 0 : (t_lo - pl.pos + 3) / 4;
    w.at_hi = (MTN + gw - pl.pos) / 4;
    if (w.at_hi > pl.attempts) w.at_hi = pl.attempts;
    if (w.at_hi < w.at_lo) w.at_hi = w.at_lo;
    return w;
}

int mt_ensure(ldpc_hip_ctx *c, const MtWindow &w) {
    using namespace ldpc_mt;
    DeviceState &m = c->mt;
    const long long slots = w.S + 2;   // + two scratch states for the chain of jumps to the window's first stream
    if (slots > m.cap_streams) {
        if (m.d_states) (void)hipFree(m.d_states);
        m.d_states = nullptr; m.cap_streams = 0;
        HIP_TRY(hipMalloc(&m.d_states, sizeof(uint32_t) * MTN * (size_t)slots));
        m.cap_streams = (int)slots;
    }
    const size_t words = (size_t)MTN + (size_t)w.S * (size_t)w.stride + 64;
    if (words > m.cap_words) {
        if (m.d_xraw) (void)hipFree(m.d_xraw);
        m.d_xraw = nullptr; m.cap_words = 0;
        HIP_TRY(hipMalloc(&m.d_xraw, sizeof(uint32_t) * words));
        m.cap_words = words;
    }
    const long long nb = (w.at_hi - w.at_lo + kPolarBlock - 1) / kPolarBlock + 1;
    if (nb > m.cap_blocks) {
        if (m.d_status) (void)hipFree(m.d_status);
        m.d_status = nullptr; m.cap_blocks = 0;
        HIP_TRY(hipMalloc(&m.d_status, sizeof(unsigned long long) * (size_t)nb));
        m.cap_blocks = nb;
    }
    return 0;
}

// start states of the window's streams (d_states[0 .. S)), then the words themselves (d_xraw)
int mt_generate(ldpc_hip_ctx *c, const MtWindow &w, hipStream_t st) {
    using namespace ldpc_mt;
    DeviceState &m = c->mt;
    const JumpPolys &J = jump_polys();
    const size_t pl0 = (size_t)(w.ls - kLog2Stride);
    uint32_t *t0 = m.d_states + (size_t)MTN * (size_t)w.S, *t1 = t0 + MTN;
    if (w.first == 0) {
        HIP_TRY(hipMemcpyAsync(m.d_states, m.d_state, sizeof(uint32_t) * MTN, hipMemcpyDeviceToDevice, st));
    } else {
        // the state first * stride words on: one jump per set bit of `first` (x^(a+b) = x^a x^b), through two scratch slots
        HIP_TRY(hipMemcpyAsync(t0, m.d_state, sizeof(uint32_t) * MTN, hipMemcpyDeviceToDevice, st));
        for (int e = 0; (w.first >> e) != 0; ++e) {
            if (!((w.first >> e) & 1)) continue;
            const size_t pl = pl0 + (size_t)e;
            if (pl >= (size_t)kLevels) return fail(LDPC_HIP_EUNSUPPORTED, "exact-replay window starts beyond 2^%d words", kLog2Stride + kLevels);
            HIP_TRY(hipMemsetAsync(t1, 0, sizeof(uint32_t) * MTN, st));
            JumpArgs ja{m.d_states, m.d_bits + pl * kMaxBits, J.nbits[pl], 8, (int)(t0 - m.d_states) / MTN, (int)(t1 - m.d_states) / MTN};
            hipLaunchKernelGGL(mt_jump_kernel, dim3(8), dim3(kJumpThreads), sizeof(uint32_t) * kSeqWords, st, ja);
            uint32_t *t = t0; t0 = t1; t1 = t;
        }
        HIP_TRY(hipMemcpyAsync(m.d_states, t0, sizeof(uint32_t) * MTN, hipMemcpyDeviceToDevice, st));
    }
    if (w.S > 1) HIP_TRY(hipMemsetAsync(m.d_states + MTN, 0, sizeof(uint32_t) * MTN * (size_t)(w.S - 1), st));   // the jumps XOR into them
    for (int level = 0; (1ll << level) < w.S; ++level) {   // stream j + 2^level from stream j, j < 2^level
        const long long have = 1ll << level, cnt = have < w.S - have ? have : w.S - have;
        const int parts = cnt >= 256 ? 1 : cnt >= 128 ? 2 : cnt >= 64 ? 4 : 8;   // few jumps: spread each over several CUs
        const size_t pl = pl0 + (size_t)level;   // the polynomial of 2^(ls + level) words
        JumpArgs ja{m.d_states, m.d_bits + pl * kMaxBits, J.nbits[pl], parts, 0, (int)have};
        hipLaunchKernelGGL(mt_jump_kernel, dim3((unsigned)(cnt * parts)), dim3(kJumpThreads), sizeof(uint32_t) * kSeqWords, st, ja);
    }
    GenArgs ga{m.d_states, m.d_xraw, (int)w.S, w.first, w.ls, w.gen_words};
    hipLaunchKernelGGL(mt_generate_kernel, dim3((unsigned)w.S), dim3(kGenThreads), 0, st, ga);
    HIP_TRY(hipGetLastError());
    return 0;
}

void mt_fill(ldpc_hip_ctx *c, const MtPlan &pl, const MtWindow &w, ldpc_mt::PolarArgs &a) {
    a.xraw = c->mt.d_xraw; a.xbase = w.xbase; a.p = pl.pos; a.need = pl.need;
    a.status = c->mt.d_status; a.ticket = c->mt.d_ticket; a.counters = c->mt.d_counters;
}

// accepted attempts in [at_lo, split) and [split, at_hi) -> d_counters[0], [1] (asynchronous)
int mt_count(ldpc_hip_ctx *c, const MtPlan &pl, const MtWindow &w, long long at_lo, long long split, long long at_hi, hipStream_t st) {
    using namespace ldpc_mt;
    PolarArgs a{};
    mt_fill(c, pl, w, a);
    a.at_lo = at_lo; a.at_hi = at_hi; a.split_at = split;
    HIP_TRY(hipMemsetAsync(c->mt.d_counters, 0, sizeof(unsigned long long) * 2, st));
    const long long nb = (at_hi - at_lo + kPolarBlock - 1) / kPolarBlock;
    if (nb > 0) hipLaunchKernelGGL(mt_polar_kernel<0>, dim3((unsigned)nb), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// the fused pass over the window's attempts; base0 = accepted attempts in front of w.at_lo (asynchronous)
int mt_emit(ldpc_hip_ctx *c, const MtPlan &pl, const MtWindow &w, ldpc_mt::PolarArgs a, unsigned long long base0, hipStream_t st) {
    using namespace ldpc_mt;
    mt_fill(c, pl, w, a);
    a.at_lo = w.at_lo; a.at_hi = w.at_hi; a.base0 = base0;
    const long long nb = (w.at_hi - w.at_lo + kPolarBlock - 1) / kPolarBlock;
    HIP_TRY(hipMemsetAsync(c->mt.d_ticket, 0, sizeof(unsigned), st));
    if (nb > 0) {
        HIP_TRY(hipMemsetAsync(c->mt.d_status, 0, sizeof(unsigned long long) * (size_t)nb, st));
        hipLaunchKernelGGL(mt_polar_kernel<1>, dim3((unsigned)nb), dim3(256), 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// where the generator continues (limit_in < 0: decided from this window's own total) -> d_total[2], d_end_t[2], d_state when found
int mt_finish(ldpc_hip_ctx *c, const MtPlan &pl, const MtWindow &w, ldpc_mt::PolarArgs a, unsigned long long base0, long long limit_in, hipStream_t st) {
    using namespace ldpc_mt;
    mt_fill(c, pl, w, a);
    a.at_lo = w.at_lo; a.at_hi = w.at_hi; a.base0 = base0;
    FinishArgs f{};
    f.pa = a;
    f.blocks = (w.at_hi - w.at_lo + kPolarBlock - 1) / kPolarBlock;
    f.limit_in = limit_in; f.end_preset = pl.pos; f.xwords = w.xwords;
    f.total = c->mt.d_total; f.end_t = c->mt.d_end_t; f.state = c->mt.d_state_next;   // committed by the caller once the round is known to be good
    hipLaunchKernelGGL(mt_finish_kernel, dim3(1), dim3(256), 0, st, f);
    HIP_TRY(hipGetLastError());
    return 0;
}

// One generation round on one context, whole tape: draws from the context's generator until `need` items exist or the round's
// words run out, emits min(need, produced) items (whole frames when proto.per_frame != 0) through proto.out, and moves the generator
// to the word after the last emitted item.  Synchronises the stream (the count of accepted attempts decides how far the round got).
int mt_round(ldpc_hip_ctx *c, unsigned long long need, ldpc_mt::PolarArgs proto, unsigned long long *emitted, hipStream_t st) {
    using namespace ldpc_mt;
    DeviceState &m = c->mt;
    const MtPlan pl = mt_plan(m.pos, need);
    const MtWindow w = mt_window(pl, 0, pl.pos + 4 * pl.attempts + MTN);
    if (int rc = mt_ensure(c, w)) return rc;
    if (int rc = mt_generate(c, w, st)) return rc;
    if (int rc = mt_emit(c, pl, w, proto, 0ull, st)) return rc;
    if (int rc = mt_finish(c, pl, w, proto, 0ull, -1, st)) return rc;
    HIP_TRY(hipMemcpyAsync(m.d_state, m.d_state_next, sizeof(uint32_t) * MTN, hipMemcpyDeviceToDevice, st));
    unsigned long long tot[2] = {0, 0};
    long long endt[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(tot, m.d_total, sizeof tot, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(endt, m.d_end_t, sizeof endt, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (endt[1] != 1) return fail(LDPC_HIP_EHIP, "exact-replay generator: the end of the round was not found (limit %llu of %llu accepted)", tot[1], tot[0]);
    m.pos = 0;
    *emitted = tot[1];
    return 0;
}

// what the emit pass needs to turn samples into decoder inputs of this context's chain (bp_simulation.cpp:603/610, :684, :697-710)
int mt_frame_proto(ldpc_hip_ctx *c, double snr_db, int modulation_type, int punctured_blocks, ldpc_mt::PolarArgs &a) {
    using namespace ldpc_mt;
    if (!c->mt.set) return fail(LDPC_HIP_EINVAL, "the generator has no state: call ldpc_hip_mt_set_state first");
    if (modulation_type != 0 && modulation_type != 1)
        return fail(LDPC_HIP_EUNSUPPORTED, "exact replay covers modulation_type 0 (BPSK) and 1 (QAM4): upstream's QAM16+ wiring is broken (SURVEY Appendix B Q5/Q6)");
    if ((unsigned long long)c->N > kRoundItems) return fail(LDPC_HIP_EUNSUPPORTED, "code length %d is beyond one generation round", c->N);
    a = PolarArgs{};
    if (int rc = awgn_sigma(c, snr_db, modulation_type, punctured_blocks, &a.sigma)) return rc;
    if (int rc = prepare_chain(c, modulation_type)) return rc;
    a.per_frame = c->N;
    a.tx = c->ncw > 0 ? c->d_tx : nullptr; a.ncw = c->ncw > 0 ? c->ncw : 1; a.ntx = c->chain_ntx;
    a.scatter = c->d_scatter;
    a.punct_start = c->N - c->M * punctured_blocks;
    a.punct_val = (c->decoder_id == LDPC_HIP_SP_DEC || c->decoder_id == LDPC_HIP_TASP_DEC || c->decoder_id == LDPC_HIP_ASP_DEC) ? 0.0 : 0.5;  // :700 (sic)
    return 0;
}

inline long long mt_frames_per_round(const ldpc_hip_ctx *c) {
    const long long per = (long long)(ldpc_mt::kRoundItems / (unsigned long long)c->N);
    return per > 0 ? per : 1;
}

// LLR rows of the next B frames of the generator's stream; of those, frames [lo, hi) are written to d_rows ([hi - lo][N]); d_rows
// may be null (the frames are drawn and dropped).
int mt_llr_rows(ldpc_hip_ctx *c, double snr_db, int modulation_type, int punctured_blocks, long long B, long long lo, long long hi,
                double *d_rows, hipStream_t st) {
    using namespace ldpc_mt;
    DeviceState &m = c->mt;
    PolarArgs a{};
    if (int rc = mt_frame_proto(c, snr_db, modulation_type, punctured_blocks, a)) return rc;
    const long long per_round = mt_frames_per_round(c);
    long long done = 0;
    int stalled = 0;
    while (done < B) {
        const long long fr = B - done < per_round ? B - done : per_round;
        const long long r_lo = lo > done ? lo : done, r_hi = hi < done + fr ? hi : done + fr;
        a.first_frame = m.frames_taken;
        if (d_rows && r_lo < r_hi) { a.row_lo = r_lo - done; a.row_hi = r_hi - done; a.out = d_rows + (r_lo - lo) * (long long)c->N; }
        else { a.row_lo = 0; a.row_hi = 0; a.out = nullptr; }
        unsigned long long emitted = 0;
        if (int rc = mt_round(c, (unsigned long long)fr * (unsigned long long)c->N, a, &emitted, st)) return rc;
        const long long fdone = (long long)(emitted / (unsigned long long)c->N);
        done += fdone;
        m.frames_taken += fdone;
        // a round covers its frames with an 8-sigma margin; one that completes NO frame twice in a row means something is broken
        stalled = fdone == 0 ? stalled + 1 : 0;
        if (stalled >= 2) return fail(LDPC_HIP_EHIP, "the exact-replay generator made no progress (frame of %d samples)", c->N);
    }
    return 0;
}

// frames [lo, hi) of the next B frames: noise -> decode -> count on `st`; records to HOST arrays info[hi - lo], iters[hi - lo]
int mt_frames_slice(ldpc_hip_ctx *c, double snr_db, int modulation_type, int punctured_blocks, int maxiter, double alpha, long long B,
                    long long lo, long long hi, int32_t *info, int32_t *iters, hipStream_t st) {
    ldpc_mt::DeviceState &m = c->mt;
    const long long mine = hi - lo;
    if (mine > m.cap_rec) {
        if (m.d_info) (void)hipFree(m.d_info);
        if (m.d_iters) (void)hipFree(m.d_iters);
        m.d_info = nullptr; m.d_iters = nullptr; m.cap_rec = 0;
        HIP_TRY(hipMalloc(&m.d_info, sizeof(int32_t) * (size_t)mine));
        HIP_TRY(hipMalloc(&m.d_iters, sizeof(int32_t) * (size_t)mine));
        m.cap_rec = mine;
    }
    long long chunk_max = ((long long)8 << 30) / ((long long)sizeof(double) * c->N);
    chunk_max = chunk_max > (1 << 16) ? (1 << 16) : (chunk_max < 1 ? 1 : chunk_max);
    const long long first = m.frames_taken;
    HIP_TRY(hipMemsetAsync(c->w_counters, 0, sizeof(unsigned long long) * 8, st));   // the count kernel accumulates; nobody reads the sum here
    for (long long c0 = 0; c0 < B; c0 += chunk_max) {
        const long long c1 = c0 + chunk_max < B ? c0 + chunk_max : B;
        const long long r_lo = lo > c0 ? lo : c0, r_hi = hi < c1 ? hi : c1, rows = r_hi > r_lo ? r_hi - r_lo : 0;
        if (rows > 0) { if (int rc = ensure_workspace(c, rows, false)) return rc; }
        if (int rc = mt_llr_rows(c, snr_db, modulation_type, punctured_blocks, c1 - c0, rows > 0 ? r_lo - c0 : 0, rows > 0 ? r_hi - c0 : 0,
                                 rows > 0 ? c->w_llr : nullptr, st))
            return rc;
        if (rows == 0) continue;
        int32_t *it = m.d_iters + (r_lo - lo);
        if (int rc = ldpc_hip_decode_dev(c, c->w_llr, rows, maxiter, alpha, c->w_hard, it, nullptr, st)) return rc;
        if (int rc = ldpc_hip_count_errors_cw_dev(c, c->w_hard, it, first + r_lo, rows, m.d_info + (r_lo - lo), c->w_counters, st)) return rc;
    }
    if (mine > 0) {
        HIP_TRY(hipMemcpyAsync(info, m.d_info, sizeof(int32_t) * (size_t)mine, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(iters, m.d_iters, sizeof(int32_t) * (size_t)mine, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

}  // namespace

extern "C" {

int ldpc_hip_mt_jump_host(const uint32_t state_in[624], int log2_words, uint32_t state_out[624]) {
    if (!state_in || !state_out || log2_words < ldpc_mt::kLog2Stride || log2_words >= ldpc_mt::kLog2Stride + ldpc_mt::kLevels)
        return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_jump_host: log2_words must be in [%d, %d]", ldpc_mt::kLog2Stride, ldpc_mt::kLog2Stride + ldpc_mt::kLevels - 1);
    const ldpc_mt::JumpPolys &J = ldpc_mt::jump_polys();
    if (!J.ok) return fail(LDPC_HIP_EUNSUPPORTED, "mt19937 jump polynomials: %s", J.err.c_str());
    ldpc_mt::jump_host(state_in, J.poly.data() + (size_t)(log2_words - ldpc_mt::kLog2Stride) * ldpc_mt::MTN, state_out);
    return 0;
}

int ldpc_hip_mt_set_state(ldpc_hip_ctx *c, const uint32_t state[624], int pos) {
    if (!c || !state || pos < 0 || pos > 624) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_set_state: bad argument");
    if (int rc = set_device(c)) return rc;
    if (int rc = mt_prepare(c)) return rc;
    HIP_TRY(hipMemcpy(c->mt.d_state, state, sizeof(uint32_t) * ldpc_mt::MTN, hipMemcpyHostToDevice));
    c->mt.pos = pos;
    c->mt.set = true;
    c->mt.frames_taken = 0;
    return 0;
}

int ldpc_hip_mt_get_state(ldpc_hip_ctx *c, uint32_t state[624], int *pos) {
    if (!c || !state || !pos) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_get_state: bad argument");
    if (!c->mt.set) return fail(LDPC_HIP_EINVAL, "the generator has no state: call ldpc_hip_mt_set_state first");
    if (int rc = set_device(c)) return rc;
    HIP_TRY(hipMemcpy(state, c->mt.d_state, sizeof(uint32_t) * ldpc_mt::MTN, hipMemcpyDeviceToHost));
    *pos = c->mt.pos;
    return 0;
}

int ldpc_hip_mt_set_frame_index(ldpc_hip_ctx *c, long long frames_taken) {
    if (!c || frames_taken < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_set_frame_index: bad argument");
    c->mt.frames_taken = frames_taken;
    return 0;
}

long long ldpc_hip_mt_get_frame_index(const ldpc_hip_ctx *c) { return c ? c->mt.frames_taken : -1; }

int ldpc_hip_mt_normal_dev(ldpc_hip_ctx *c, long long count, double *d_out, void *stream_) {
    if (!c || count < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_normal_dev: bad argument");
    if (!c->mt.set) return fail(LDPC_HIP_EINVAL, "the generator has no state: call ldpc_hip_mt_set_state first");
    if (int rc = set_device(c)) return rc;
    hipStream_t st = (hipStream_t)stream_;
    long long done = 0;
    int stalled = 0;
    while (done < count) {
        const unsigned long long want = (unsigned long long)(count - done) < ldpc_mt::kRoundItems ? (unsigned long long)(count - done) : ldpc_mt::kRoundItems;
        ldpc_mt::PolarArgs a{};
        a.out = d_out ? d_out + done : nullptr;
        unsigned long long emitted = 0;
        if (int rc = mt_round(c, want, a, &emitted, st)) return rc;
        done += (long long)emitted;
        stalled = emitted == 0 ? stalled + 1 : 0;
        if (stalled >= 2) return fail(LDPC_HIP_EHIP, "the exact-replay generator made no progress");
    }
    return 0;
}

int ldpc_hip_mt_normal_host(ldpc_hip_ctx *c, long long count, double *out) {
    if (!c || count < 0 || (count > 0 && !out)) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_normal_host: bad argument");
    if (count == 0) return 0;
    if (int rc = set_device(c)) return rc;
    double *d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(double) * (size_t)count));
    int rc = ldpc_hip_mt_normal_dev(c, count, d, nullptr);
    if (rc == 0 && hipMemcpy(out, d, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(LDPC_HIP_EHIP, "ldpc_hip_mt_normal_host: copy back failed");
    (void)hipFree(d);
    return rc;
}

int ldpc_hip_mt_llr_dev(ldpc_hip_ctx *c, double snr_db, int modulation_type, int punctured_blocks, long long B, double *d_llr, void *stream_) {
    if (!c || B < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_llr_dev: bad argument");
    if (int rc = set_device(c)) return rc;
    return mt_llr_rows(c, snr_db, modulation_type, punctured_blocks, B, 0, B, d_llr, (hipStream_t)stream_);
}

int ldpc_hip_mt_frames_slice(ldpc_hip_ctx *c, double snr_db, int modulation_type, int punctured_blocks, int maxiter, double alpha, long long B,
                             long long lo, long long hi, int32_t *frame_info, int32_t *iters) {
    if (!c || B < 0 || lo < 0 || hi < lo || hi > B || (hi > lo && (!frame_info || !iters))) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_frames_slice: bad argument");
    if (int rc = set_device(c)) return rc;
    return mt_frames_slice(c, snr_db, modulation_type, punctured_blocks, maxiter, alpha, B, lo, hi, frame_info, iters, nullptr);
}

int ldpc_hip_mt_shard_begin(ldpc_hip_ctx *c, double snr_db, int modulation_type, int punctured_blocks, long long frames, int rank, int n,
                            unsigned long long *own_count) {
    using namespace ldpc_mt;
    if (!c || frames < 1 || n < 1 || rank < 0 || rank >= n || !own_count) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_shard_begin: bad argument");
    if (frames > mt_frames_per_round(c) || frames > (1 << 16)) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_shard_begin: at most %lld frames per round", mt_frames_per_round(c) < (1 << 16) ? mt_frames_per_round(c) : (long long)(1 << 16));
    if (int rc = set_device(c)) return rc;
    MtShardRound &r = c->mt_round;
    r = MtShardRound();
    if (int rc = mt_frame_proto(c, snr_db, modulation_type, punctured_blocks, r.proto)) return rc;
    r.frames = frames; r.rank = rank; r.n = n;
    r.row_lo = frames * rank / n; r.row_hi = frames * (rank + 1) / n;
    const long long N = c->N;
    r.pl = mt_plan(c->mt.pos, (unsigned long long)frames * (unsigned long long)N);
    auto cut = [&](int i) -> long long {   // attempt index where rank i's stretch is expected to start
        if (i <= 0) return 0;
        if (i >= n) return r.pl.attempts;
        const long long at = (long long)((double)(frames * i / n) * (double)N * 1.2732395447351628);
        return at < r.pl.attempts ? at : r.pl.attempts;
    };
    r.cut_lo = cut(rank); r.cut_hi = cut(rank + 1);
    *own_count = 0;
    if (r.cut_lo >= r.cut_hi) { r.w = MtWindow(); r.open = true; return 0; }   // nothing to own (and then no rows either, or the caller falls back)
    const long long a_lo = r.cut_lo - r.pl.margin, a_hi = r.cut_hi + r.pl.margin;
    r.w = mt_window(r.pl, r.pl.pos + 4 * (a_lo < 0 ? 0 : a_lo), r.pl.pos + 4 * (a_hi > r.pl.attempts ? r.pl.attempts : a_hi) + MTN);
    if (int rc = mt_ensure(c, r.w)) return rc;
    if (int rc = mt_generate(c, r.w, nullptr)) return rc;
    if (int rc = mt_count(c, r.pl, r.w, r.w.at_lo, r.cut_lo, r.cut_hi, nullptr)) return rc;
    unsigned long long cnt[2] = {0, 0};
    HIP_TRY(hipMemcpy(cnt, c->mt.d_counters, sizeof cnt, hipMemcpyDeviceToHost));
    r.left = cnt[0]; r.own = cnt[1];
    *own_count = r.own;
    r.open = true;   // only now: a begin that failed half way leaves no round behind for emit / commit to pick up
    return 0;
}

int ldpc_hip_mt_shard_emit(ldpc_hip_ctx *c, const unsigned long long *counts, int *found, int *covered, uint32_t state_next[624], long long *frames_done) {
    using namespace ldpc_mt;
    if (!c || !counts || !found || !covered || !state_next || !frames_done) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_shard_emit: bad argument");
    MtShardRound &r = c->mt_round;
    if (!r.open) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_shard_emit: no round is open (ldpc_hip_mt_shard_begin first)");
    if (int rc = set_device(c)) return rc;
    const unsigned long long N = (unsigned long long)c->N;
    unsigned long long before = 0, total = 0;
    for (int i = 0; i < r.n; ++i) { if (i < r.rank) before += counts[i]; total += counts[i]; }
    r.limit = total < r.pl.need ? total : r.pl.need;
    r.limit -= r.limit % N;
    *frames_done = (long long)(r.limit / N);
    *found = 0;
    const unsigned long long lo_item = (unsigned long long)r.row_lo * N;
    unsigned long long hi_item = (unsigned long long)r.row_hi * N;
    if (hi_item > r.limit) hi_item = r.limit;
    const bool wants_rows = lo_item < hi_item;
    *covered = wants_rows ? 0 : 1;
    // the window must reach from its cut's margin to the next cut (it does unless the tape ended early)
    if (r.w.S == 0 || r.w.at_lo > r.cut_lo || r.w.at_hi < r.cut_hi) return 0;
    r.base0 = before - r.left;
    const long long rows = r.row_hi - r.row_lo;
    if (rows > 0) { if (int rc = ensure_workspace(c, rows, false)) return rc; }
    PolarArgs a = r.proto;
    a.first_frame = c->mt.frames_taken;
    if (rows > 0) { a.row_lo = r.row_lo; a.row_hi = r.row_hi; a.out = c->w_llr; } else { a.row_lo = 0; a.row_hi = 0; a.out = nullptr; }
    if (int rc = mt_emit(c, r.pl, r.w, a, r.base0, nullptr)) return rc;
    if (int rc = mt_finish(c, r.pl, r.w, a, r.base0, (long long)r.limit, nullptr)) return rc;
    unsigned long long tot[2] = {0, 0};
    long long endt[2] = {0, 0};
    HIP_TRY(hipMemcpy(tot, c->mt.d_total, sizeof tot, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(endt, c->mt.d_end_t, sizeof endt, hipMemcpyDeviceToHost));
    if (endt[1] == 1) {
        *found = 1;
        HIP_TRY(hipMemcpy(state_next, c->mt.d_state_next, sizeof(uint32_t) * MTN, hipMemcpyDeviceToHost));
    }
    if (wants_rows) *covered = (r.base0 <= lo_item && r.base0 + tot[0] >= hi_item) ? 1 : 0;
    return 0;
}

int ldpc_hip_mt_shard_commit(ldpc_hip_ctx *c, const uint32_t state[624], long long frames_done, int maxiter, double alpha, int32_t *frame_info,
                             int32_t *iters) {
    using namespace ldpc_mt;
    if (!c || !state || frames_done < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_shard_commit: bad argument");
    MtShardRound &r = c->mt_round;
    if (!r.open) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_shard_commit: no round is open");
    if (int rc = set_device(c)) return rc;
    r.open = false;
    HIP_TRY(hipMemcpy(c->mt.d_state, state, sizeof(uint32_t) * MTN, hipMemcpyHostToDevice));
    c->mt.pos = 0;
    const long long first = c->mt.frames_taken;
    c->mt.frames_taken += frames_done;
    const long long r_hi = r.row_hi < frames_done ? r.row_hi : frames_done, rows = r_hi - r.row_lo;
    if (rows <= 0) return 0;
    if (!frame_info || !iters) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_shard_commit: null record arrays");
    DeviceState &m = c->mt;
    if (rows > m.cap_rec) {
        if (m.d_info) (void)hipFree(m.d_info);
        if (m.d_iters) (void)hipFree(m.d_iters);
        m.d_info = nullptr; m.d_iters = nullptr; m.cap_rec = 0;
        HIP_TRY(hipMalloc(&m.d_info, sizeof(int32_t) * (size_t)rows));
        HIP_TRY(hipMalloc(&m.d_iters, sizeof(int32_t) * (size_t)rows));
        m.cap_rec = rows;
    }
    HIP_TRY(hipMemsetAsync(c->w_counters, 0, sizeof(unsigned long long) * 8, nullptr));
    if (int rc = ldpc_hip_decode_dev(c, c->w_llr, rows, maxiter, alpha, c->w_hard, m.d_iters, nullptr, nullptr)) return rc;
    if (int rc = ldpc_hip_count_errors_cw_dev(c, c->w_hard, m.d_iters, first + r.row_lo, rows, m.d_info, c->w_counters, nullptr)) return rc;
    HIP_TRY(hipMemcpy(frame_info, m.d_info, sizeof(int32_t) * (size_t)rows, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(iters, m.d_iters, sizeof(int32_t) * (size_t)rows, hipMemcpyDeviceToHost));
    return 0;
}

void ldpc_hip_mt_shard_abandon(ldpc_hip_ctx *c) { if (c) c->mt_round.open = false; }

int ldpc_hip_mt_frames(ldpc_hip_ctx *c, double snr_db, int modulation_type, int punctured_blocks, int maxiter, double alpha, long long B,
                       int32_t *frame_info, int32_t *iters) {
    if (!c || B < 0 || !frame_info || !iters) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_frames: bad argument");
    if (int rc = set_device(c)) return rc;
    return mt_frames_slice(c, snr_db, modulation_type, punctured_blocks, maxiter, alpha, B, 0, B, frame_info, iters, nullptr);
}

}  // extern "C"
