// ldpc_multi.hpp -- several GPUs of one node behind the C-ABI (included at the end of ldpc_hip.hip).
//
// north_star: "frame batches shard trivially across the 8 GPUs of one node with an RCCL all-reduce only for the FER/BER
// error counters".  One shard = one ldpc_hip_ctx + one HIP stream + one host thread; global frame f of a call that starts at
// first_frame belongs to shard ((f - first_frame) / batch) mod n.  The channel noise is keyed by the GLOBAL frame index, so the
// result does not depend on n, on `batch` or on which GPU decodes a frame.  There is no data-path collective: the only
// exchange is one all-reduce of the five uint64 counters {nse, nde, nue, frames, sum |iters|} per call (RCCL over xGMI when
// the shards sit on distinct devices; shards that share a device -- the single-GPU test configuration -- are summed on the
// host, RCCL refuses duplicate devices in one communicator).  The sequential stopping rule of bp_simulation.cpp:591,820
// needs ordered per-frame results: ldpc_hip_frames_multi returns the 8-byte records (frame_info, iters) in global frame order.
//
// RCCL is loaded with dlopen (like hiprtc): inside a PyTorch process the copy torch already mapped is reused, a C/C++ host gets
// /opt/rocm's; libldpc_hip.so itself has no link-time dependency on it.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enum values only; the entry points are resolved below

#include <string>
#include <thread>
#include <vector>

namespace ldpc_multi {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

inline Rccl *rccl(std::string &err) {
    static Rccl r;
    static bool tried = false;
    if (tried) { if (!r.lib) err = "librccl not available"; return r.lib ? &r : nullptr; }
    tried = true;
    r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);   // the copy this process already has (torch's), if any
    if (!r.lib) r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    std::vector<std::string> names;
    if (const char *p = getenv("LDPC_HIP_RCCL_PATH")) names.push_back(p);
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    names.push_back("/opt/rocm/lib/librccl.so");
    for (size_t i = 0; !r.lib && i < names.size(); ++i) r.lib = dlopen(names[i].c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) { err = "dlopen(librccl.so) failed"; return nullptr; }
#define LDPC_RCCL_SYM(f)                                                      \
    r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.lib, "nccl" #f));           \
    if (!r.f) { err = "nccl" #f " missing"; r.lib = nullptr; return nullptr; }
    LDPC_RCCL_SYM(CommInitAll) LDPC_RCCL_SYM(CommDestroy) LDPC_RCCL_SYM(AllReduce) LDPC_RCCL_SYM(GetErrorString)
#undef LDPC_RCCL_SYM
    return &r;
}

}  // namespace ldpc_multi

struct ldpc_hip_multi {
    std::vector<ldpc_hip_ctx *> shard;
    std::vector<int> device;
    std::vector<hipStream_t> stream;
    std::vector<unsigned long long *> d_red;   // [8] per shard: all-reduce result
    std::vector<int32_t *> d_info, d_iters;    // per shard: per-frame records of one call
    std::vector<long long> rec_frames;
    std::vector<ncclComm_t> comm;              // empty: counters are summed on the host
    std::string reduction = "host";
};

namespace {

struct ShardResult { int rc = 0; std::string err; };

// runs fn(shard) on one host thread per shard and reports the first failure through the calling thread's error string
template <class F>
int for_each_shard(ldpc_hip_multi *m, F fn) {
    const int n = (int)m->shard.size();
    std::vector<ShardResult> res((size_t)n);
    auto body = [&](int i) {
        res[(size_t)i].rc = fn(i);
        if (res[(size_t)i].rc != 0) res[(size_t)i].err = g_err;   // g_err is thread_local: carry the message to the caller's thread
    };
    if (n == 1) {
        body(0);
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < n; ++i) th.emplace_back(body, i);
        for (auto &t : th) t.join();
    }
    for (int i = 0; i < n; ++i)
        if (res[(size_t)i].rc != 0) return fail(res[(size_t)i].rc, "shard %d (device %d): %s", i, m->device[(size_t)i], res[(size_t)i].err.c_str());
    return 0;
}

int multi_records(ldpc_hip_multi *m, int i, long long frames) {
    if (frames <= m->rec_frames[(size_t)i]) return 0;
    if (m->d_info[(size_t)i]) (void)hipFree(m->d_info[(size_t)i]);
    if (m->d_iters[(size_t)i]) (void)hipFree(m->d_iters[(size_t)i]);
    m->d_info[(size_t)i] = nullptr; m->d_iters[(size_t)i] = nullptr; m->rec_frames[(size_t)i] = 0;
    HIP_TRY(hipMalloc(&m->d_info[(size_t)i], sizeof(int32_t) * (size_t)frames));
    HIP_TRY(hipMalloc(&m->d_iters[(size_t)i], sizeof(int32_t) * (size_t)frames));
    m->rec_frames[(size_t)i] = frames;
    return 0;
}

// frames [first_frame, first_frame + B) in batches of `batch`, batch k to shard k mod n; counters all-reduced; optional records
int multi_run(ldpc_hip_multi *m, double snr_db, int modulation_type, int punctured_blocks, int maxiter, double alpha, uint64_t seed,
              long long first_frame, long long B, long long batch, unsigned long long counters[4], unsigned long long *sum_abs_iters,
              int32_t *frame_info, int32_t *iters) {
    if (!m || B < 0 || batch <= 0 || first_frame < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_simulate_multi: bad argument");
    const int n = (int)m->shard.size();
    const long long nbatches = (B + batch - 1) / batch;
    const bool records = frame_info != nullptr || iters != nullptr;
    std::vector<unsigned long long> host_cnt((size_t)n * 8, 0ull);
    const int rc = for_each_shard(m, [&](int i) -> int {
        ldpc_hip_ctx *c = m->shard[(size_t)i];
        if (int r = set_device(c)) return r;
        hipStream_t st = m->stream[(size_t)i];
        long long mine = 0;   // frames of this shard
        for (long long k = i; k < nbatches; k += n) mine += (k + 1) * batch <= B ? batch : B - k * batch;
        if (records) { if (int r = multi_records(m, i, mine)) return r; }
        HIP_TRY(hipMemsetAsync(c->w_counters, 0, sizeof(unsigned long long) * 8, st));
        long long off = 0;
        for (long long k = i; k < nbatches; k += n) {
            const long long nb = (k + 1) * batch <= B ? batch : B - k * batch;
            if (int r = sim_enqueue(c, snr_db, modulation_type, punctured_blocks, maxiter, alpha, seed, first_frame + k * batch, nb,
                                    records ? m->d_info[(size_t)i] + off : nullptr, records ? m->d_iters[(size_t)i] + off : nullptr, st))
                return r;
            off += nb;
        }
        if (!m->comm.empty()) {   // one 40-byte all-reduce per call over RCCL / xGMI
            std::string err;
            ldpc_multi::Rccl *r = ldpc_multi::rccl(err);
            const ncclResult_t nr = r->AllReduce(c->w_counters, m->d_red[(size_t)i], 5, ncclUint64, ncclSum, m->comm[(size_t)i], st);
            if (nr != ncclSuccess) return fail(LDPC_HIP_EHIP, "ncclAllReduce: %s", r->GetErrorString(nr));
            HIP_TRY(hipMemcpyAsync(&host_cnt[(size_t)i * 8], m->d_red[(size_t)i], sizeof(unsigned long long) * 5, hipMemcpyDeviceToHost, st));
        } else {
            HIP_TRY(hipMemcpyAsync(&host_cnt[(size_t)i * 8], c->w_counters, sizeof(unsigned long long) * 5, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(hipStreamSynchronize(st));
        if (records) {   // hand the ordered records back: batch k covers global frames [k*batch, k*batch + nb)
            off = 0;
            for (long long k = i; k < nbatches; k += n) {
                const long long nb = (k + 1) * batch <= B ? batch : B - k * batch;
                if (frame_info) HIP_TRY(hipMemcpy(frame_info + k * batch, m->d_info[(size_t)i] + off, sizeof(int32_t) * (size_t)nb, hipMemcpyDeviceToHost));
                if (iters) HIP_TRY(hipMemcpy(iters + k * batch, m->d_iters[(size_t)i] + off, sizeof(int32_t) * (size_t)nb, hipMemcpyDeviceToHost));
                off += nb;
            }
        }
        return 0;
    });
    if (rc) return rc;
    unsigned long long tot[5] = {0, 0, 0, 0, 0};
    if (!m->comm.empty()) {
        for (int j = 0; j < 5; ++j) tot[j] = host_cnt[(size_t)j];   // every rank holds the sum; they must agree
        for (int i = 1; i < n; ++i)
            for (int j = 0; j < 5; ++j)
                if (host_cnt[(size_t)i * 8 + j] != tot[j]) return fail(LDPC_HIP_EHIP, "all-reduce result differs between ranks 0 and %d", i);
    } else {
        for (int i = 0; i < n; ++i) for (int j = 0; j < 5; ++j) tot[j] += host_cnt[(size_t)i * 8 + j];
    }
    if (counters) { counters[0] = tot[0]; counters[1] = tot[1]; counters[2] = tot[2]; counters[3] = tot[3]; }
    if (sum_abs_iters) *sum_abs_iters = tot[4];
    return 0;
}

}  // namespace

extern "C" {

int ldpc_hip_open_multi(int decoder_id, int rh, int nh, int M, const int16_t *hd, const int *devices, int n_shards, ldpc_hip_multi **out) {
    if (out) *out = nullptr;
    if (!out || !devices || n_shards < 1 || n_shards > 64) return fail(LDPC_HIP_EINVAL, "ldpc_hip_open_multi: bad argument");
    std::unique_ptr<ldpc_hip_multi, void (*)(ldpc_hip_multi *)> m(new ldpc_hip_multi(), ldpc_hip_close_multi);
    bool distinct = true;
    for (int i = 0; i < n_shards; ++i) {
        for (int j = 0; j < i; ++j) distinct = distinct && devices[i] != devices[j];
        ldpc_hip_ctx *c = nullptr;
        if (int rc = ldpc_hip_open(decoder_id, rh, nh, M, hd, devices[i], &c)) return rc;
        m->shard.push_back(c);
        m->device.push_back(devices[i]);
        m->stream.push_back(nullptr); m->d_red.push_back(nullptr); m->d_info.push_back(nullptr); m->d_iters.push_back(nullptr);
        m->rec_frames.push_back(0);
        HIP_TRY(hipSetDevice(devices[i]));
        HIP_TRY(hipStreamCreateWithFlags(&m->stream.back(), hipStreamNonBlocking));
        HIP_TRY(hipMalloc(&m->d_red.back(), sizeof(unsigned long long) * 8));
    }
    const char *single = getenv("LDPC_HIP_RCCL_SINGLE");   // tests: run the one-rank communicator through RCCL too
    if (distinct && (n_shards > 1 || (single && atoi(single) != 0))) {
        std::string err;
        ldpc_multi::Rccl *r = ldpc_multi::rccl(err);
        if (!r) return fail(LDPC_HIP_EHIP, "ldpc_hip_open_multi: %d devices need RCCL for the counter all-reduce: %s", n_shards, err.c_str());
        m->comm.assign((size_t)n_shards, nullptr);
        const ncclResult_t nr = r->CommInitAll(m->comm.data(), n_shards, devices);
        if (nr != ncclSuccess) { m->comm.clear(); return fail(LDPC_HIP_EHIP, "ncclCommInitAll: %s", r->GetErrorString(nr)); }
        m->reduction = "rccl";
    }
    *out = m.release();
    return 0;
}

void ldpc_hip_close_multi(ldpc_hip_multi *m) {
    if (!m) return;
    if (!m->comm.empty()) {
        std::string err;
        if (ldpc_multi::Rccl *r = ldpc_multi::rccl(err))
            for (ncclComm_t cm : m->comm) if (cm) (void)r->CommDestroy(cm);
    }
    for (size_t i = 0; i < m->shard.size(); ++i) {
        (void)hipSetDevice(m->device[i]);
        if (m->stream[i]) { (void)hipStreamSynchronize(m->stream[i]); (void)hipStreamDestroy(m->stream[i]); }
        if (m->d_red[i]) (void)hipFree(m->d_red[i]);
        if (m->d_info[i]) (void)hipFree(m->d_info[i]);
        if (m->d_iters[i]) (void)hipFree(m->d_iters[i]);
        ldpc_hip_close(m->shard[i]);
    }
    delete m;
}

int ldpc_hip_multi_shards(const ldpc_hip_multi *m) { return m ? (int)m->shard.size() : 0; }
ldpc_hip_ctx *ldpc_hip_multi_ctx(ldpc_hip_multi *m, int shard) { return (m && shard >= 0 && shard < (int)m->shard.size()) ? m->shard[(size_t)shard] : nullptr; }
const char *ldpc_hip_multi_reduction(const ldpc_hip_multi *m) { return m ? m->reduction.c_str() : ""; }

int ldpc_hip_multi_set_interleaver(ldpc_hip_multi *m, int permutation_type, int permutation_block, int permutation_inter) {
    if (!m) return fail(LDPC_HIP_EINVAL, "null multi context");
    for (ldpc_hip_ctx *c : m->shard) if (int rc = ldpc_hip_set_interleaver(c, permutation_type, permutation_block, permutation_inter)) return rc;
    return 0;
}

int ldpc_hip_multi_set_codewords(ldpc_hip_multi *m, const uint8_t *codewords, int ncw) {
    if (!m) return fail(LDPC_HIP_EINVAL, "null multi context");
    for (ldpc_hip_ctx *c : m->shard) if (int rc = ldpc_hip_set_codewords(c, codewords, ncw)) return rc;
    return 0;
}

int ldpc_hip_multi_set_random_codewords(ldpc_hip_multi *m, uint64_t seed, int ncw) {
    if (!m) return fail(LDPC_HIP_EINVAL, "null multi context");
    for (ldpc_hip_ctx *c : m->shard) if (int rc = ldpc_hip_set_random_codewords(c, seed, ncw)) return rc;   // the same table on every shard
    return 0;
}

int ldpc_hip_simulate_multi(ldpc_hip_multi *m, double snr_db, int modulation_type, int punctured_blocks, int maxiter, double alpha,
                            uint64_t seed, long long first_frame, long long B, long long batch, unsigned long long counters[4],
                            unsigned long long *sum_abs_iters) {
    if (!counters) return fail(LDPC_HIP_EINVAL, "ldpc_hip_simulate_multi: null counters");
    return multi_run(m, snr_db, modulation_type, punctured_blocks, maxiter, alpha, seed, first_frame, B, batch, counters, sum_abs_iters, nullptr, nullptr);
}

int ldpc_hip_frames_multi(ldpc_hip_multi *m, double snr_db, int modulation_type, int punctured_blocks, int maxiter, double alpha,
                          uint64_t seed, long long first_frame, long long B, long long batch, int32_t *frame_info, int32_t *iters,
                          unsigned long long counters[4], unsigned long long *sum_abs_iters) {
    if (!frame_info || !iters) return fail(LDPC_HIP_EINVAL, "ldpc_hip_frames_multi: null record arrays");
    return multi_run(m, snr_db, modulation_type, punctured_blocks, maxiter, alpha, seed, first_frame, B, batch, counters, sum_abs_iters, frame_info, iters);
}

int ldpc_hip_decode_host_multi(ldpc_hip_multi *m, double *llr, long long B, int maxiter, int decision, double alpha, double *decword,
                               int32_t *iters, int clobber_sp_input) {
    if (!m || !llr || B < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_decode_host_multi: bad argument");
    const int n = (int)m->shard.size();
    // BP_DEC with the frame chain on is sequential by definition (decoders.cpp:1742-1762): one shard decodes the whole batch
    const bool chained = m->shard[0]->decoder_id == LDPC_HIP_BP_DEC && m->shard[0]->bp_chain;
    if (n == 1 || chained || B < n) return ldpc_hip_decode_host(m->shard[0], llr, B, maxiter, decision, alpha, decword, iters, clobber_sp_input);
    const long long N = m->shard[0]->N;
    return for_each_shard(m, [&](int i) -> int {
        const long long lo = B * i / n, hi = B * (i + 1) / n;   // contiguous slices, host order preserved
        return ldpc_hip_decode_host(m->shard[(size_t)i], llr + lo * N, hi - lo, maxiter, decision, alpha, decword ? decword + lo * N : nullptr,
                                    iters ? iters + lo : nullptr, clobber_sp_input);
    });
}

// ---- exact replay over the shards: every shard runs the SAME generator over the whole batch (the stream is sequential by nature
// and generating it costs a fraction of the decode) and decodes its contiguous slice of the frames; no exchange at all.
int ldpc_hip_mt_set_state_multi(ldpc_hip_multi *m, const uint32_t state[624], int pos) {
    if (!m) return fail(LDPC_HIP_EINVAL, "null multi context");
    for (ldpc_hip_ctx *c : m->shard) if (int rc = ldpc_hip_mt_set_state(c, state, pos)) return rc;
    return 0;
}

int ldpc_hip_mt_get_state_multi(ldpc_hip_multi *m, uint32_t state[624], int *pos) {
    if (!m) return fail(LDPC_HIP_EINVAL, "null multi context");
    return ldpc_hip_mt_get_state(m->shard[0], state, pos);
}

int ldpc_hip_mt_advance_multi(ldpc_hip_multi *m, double snr_db, int modulation_type, int punctured_blocks, long long B) {
    if (!m || B < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_advance_multi: bad argument");
    return for_each_shard(m, [&](int i) -> int {
        ldpc_hip_ctx *c = m->shard[(size_t)i];
        if (int r = set_device(c)) return r;
        return mt_llr_rows(c, snr_db, modulation_type, punctured_blocks, B, 0, 0, nullptr, m->stream[(size_t)i]);
    });
}

int ldpc_hip_mt_frames_multi(ldpc_hip_multi *m, double snr_db, int modulation_type, int punctured_blocks, int maxiter, double alpha,
                             long long B, int32_t *frame_info, int32_t *iters) {
    if (!m || B < 0 || !frame_info || !iters) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_frames_multi: bad argument");
    const int n = (int)m->shard.size();
    // BP_DEC with the frame chain on is sequential by definition (decoders.cpp:1742-1762): shard 0 decodes the whole batch
    const bool chained = m->shard[0]->decoder_id == LDPC_HIP_BP_DEC && m->shard[0]->bp_chain;
    return for_each_shard(m, [&](int i) -> int {
        ldpc_hip_ctx *c = m->shard[(size_t)i];
        if (int r = set_device(c)) return r;
        long long lo = B * i / n, hi = B * (i + 1) / n;
        if (chained) { lo = i == 0 ? 0 : B; hi = B; }
        return mt_frames_slice(c, snr_db, modulation_type, punctured_blocks, maxiter, alpha, B, lo, hi, frame_info + lo, iters + lo, m->stream[(size_t)i]);
    });
}

}  // extern "C"
