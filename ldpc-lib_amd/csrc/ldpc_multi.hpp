// ldpc_multi.hpp -- several GPUs of one node behind the C-ABI (included at the end of ldpc_hip.hip).
//
// north_star: "frame batches shard trivially across the 8 GPUs of one node with an RCCL all-reduce only for the FER/BER
// error counters".  One shard = one ldpc_hip_ctx + one HIP stream + one host thread (kept for the lifetime of the multi
// context); global frame f of a call that starts at first_frame belongs to shard ((f - first_frame) / batch) mod n.  The channel
// noise is keyed by the GLOBAL frame index, so the result does not depend on n, on `batch` or on which GPU decodes a frame.
// There is no data-path collective: the only exchange is one all-reduce of the five uint64 counters {nse, nde, nue, frames,
// sum |iters|} per call (RCCL over xGMI when the shards sit on distinct devices; shards that share a device -- the single-GPU
// test configuration -- are summed on the host, RCCL refuses duplicate devices in one communicator).  The sequential stopping
// rule of bp_simulation.cpp:591,820 needs ordered per-frame results: ldpc_hip_frames_multi returns the 8-byte records
// (frame_info, iters) in global frame order.
//
// A call runs in two phases.  Phase 1: every shard's thread enqueues its work on its stream; the threads join.  Only if EVERY
// shard succeeded does phase 2 issue the all-reduce -- for all ranks at once, from the calling thread, inside one
// ncclGroupStart / ncclGroupEnd -- so a shard that failed can never leave its peers waiting inside a collective.
//
// Communicators are expensive to make (ncclCommInitAll over 8 GPUs takes far longer than a whole short bp_simulation call,
// and a code search calls bp_simulation once per candidate matrix and SNR: main_good_code_search.cpp:320), so they are cached per
// device list for the life of the process; ldpc_hip_close_multi only hands its set back.
//
// RCCL is loaded with dlopen (like hiprtc): LDPC_HIP_RCCL_PATH names the library; otherwise the copy this process already has
// (torch's) is reused, else /opt/rocm's; libldpc_hip.so itself has no link-time dependency on it.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enum values only; the entry points are resolved below

#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace ldpc_multi {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

inline Rccl *rccl(std::string &err) {
    static Rccl r;
    static bool tried = false;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (tried) { if (!r.lib) err = "librccl not available"; return r.lib ? &r : nullptr; }
    tried = true;
    const char *forced = getenv("LDPC_HIP_RCCL_PATH");
    if (forced && *forced) {   // an explicit library wins (binding.py names torch's copy; the tests name their host-memory stand-in)
        r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) { err = std::string("dlopen(") + forced + ") failed: " + dlerror(); return nullptr; }
    }
    if (!r.lib) r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);   // the copy this process already has, if any
    if (!r.lib) r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    for (size_t i = 0; !r.lib && i < sizeof names / sizeof names[0]; ++i) r.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) { err = "dlopen(librccl.so) failed"; return nullptr; }
#define LDPC_RCCL_SYM(f)                                                      \
    r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.lib, "nccl" #f));           \
    if (!r.f) { err = "nccl" #f " missing"; r.lib = nullptr; return nullptr; }
    LDPC_RCCL_SYM(CommInitAll) LDPC_RCCL_SYM(CommDestroy) LDPC_RCCL_SYM(AllReduce) LDPC_RCCL_SYM(GroupStart) LDPC_RCCL_SYM(GroupEnd)
    LDPC_RCCL_SYM(GetErrorString)
#undef LDPC_RCCL_SYM
    return &r;
}

// One communicator per device of a device list, made once per process and lent to one multi context at a time.
struct CommSet {
    std::vector<int> devices;
    std::vector<ncclComm_t> comm;
    std::mutex in_use;   // held for the duration of one collective: two multi contexts on the same device list take turns
};

struct CommCache {
    std::mutex mu;
    std::map<std::vector<int>, std::shared_ptr<CommSet>> sets;
    long long inits = 0;   // ncclCommInitAll calls so far (ldpc_hip_multi_comm_inits: the tests watch the cache work)
};
inline CommCache &comm_cache() {
    static CommCache *c = new CommCache();   // never destroyed: communicators outlive static destruction order (the process exit reclaims them)
    return *c;
}

inline std::shared_ptr<CommSet> acquire_comms(const int *devices, int n, std::string &err) {
    Rccl *r = rccl(err);
    if (!r) return nullptr;
    CommCache &cc = comm_cache();
    std::lock_guard<std::mutex> lock(cc.mu);
    const std::vector<int> key(devices, devices + n);
    auto it = cc.sets.find(key);
    if (it != cc.sets.end()) return it->second;
    auto cs = std::make_shared<CommSet>();
    cs->devices = key;
    cs->comm.assign((size_t)n, nullptr);
    const ncclResult_t nr = r->CommInitAll(cs->comm.data(), n, devices);
    ++cc.inits;
    if (nr != ncclSuccess) { err = std::string("ncclCommInitAll: ") + r->GetErrorString(nr); return nullptr; }
    cc.sets.emplace(key, cs);
    return cs;
}

inline void release_all_comms() {
    std::string err;
    Rccl *r = rccl(err);
    CommCache &cc = comm_cache();
    std::lock_guard<std::mutex> lock(cc.mu);
    for (auto &kv : cc.sets) {
        if (kv.second.use_count() > 1) continue;   // a live multi context still holds it
        if (r) for (ncclComm_t cm : kv.second->comm) if (cm) (void)r->CommDestroy(cm);
        kv.second->comm.clear();
    }
    for (auto it = cc.sets.begin(); it != cc.sets.end();) it = it->second->comm.empty() ? cc.sets.erase(it) : std::next(it);
}

// n host threads, one per shard, alive as long as the multi context: run(fn) executes fn(i) on thread i and returns when all are done.
class ShardPool {
public:
    explicit ShardPool(int n) : n_(n), rc_((size_t)n, 0), err_((size_t)n) {
        for (int i = 0; i < n; ++i) th_.emplace_back([this, i] { loop(i); });
    }
    ~ShardPool() {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    // first failure -> (shard, code, message of that shard's thread); 0 when every shard returned 0
    int run(const std::function<int(int)> &fn, int *failed_shard, std::string *msg, const std::string &(*last_error)()) {
        {
            std::lock_guard<std::mutex> lock(mu_);
            fn_ = &fn; last_error_ = last_error; pending_ = n_; ++gen_;
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> lock(mu_);
        done_.wait(lock, [this] { return pending_ == 0; });
        fn_ = nullptr;
        for (int i = 0; i < n_; ++i)
            if (rc_[(size_t)i] != 0) { *failed_shard = i; *msg = err_[(size_t)i]; return rc_[(size_t)i]; }
        return 0;
    }

private:
    void loop(int i) {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<int(int)> *fn;
            {
                std::unique_lock<std::mutex> lock(mu_);
                cv_.wait(lock, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                fn = fn_;
            }
            const int rc = (*fn)(i);
            std::lock_guard<std::mutex> lock(mu_);
            rc_[(size_t)i] = rc;
            err_[(size_t)i] = rc != 0 ? last_error_() : std::string();   // the error string is thread_local: carry it to the caller
            if (--pending_ == 0) done_.notify_one();
        }
    }
    const int n_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> th_;
    const std::function<int(int)> *fn_ = nullptr;
    const std::string &(*last_error_)() = nullptr;
    unsigned long long gen_ = 0;
    int pending_ = 0;
    bool stop_ = false;
    std::vector<int> rc_;
    std::vector<std::string> err_;
};

}  // namespace ldpc_multi

struct ldpc_hip_multi {
    std::vector<ldpc_hip_ctx *> shard;
    std::vector<int> device;
    std::vector<hipStream_t> stream;
    std::vector<unsigned long long *> d_red;   // [8] per shard: all-reduce result
    std::vector<int32_t *> d_info, d_iters;    // per shard: per-frame records of one call
    std::vector<long long> rec_frames;
    std::vector<uint32_t *> d_hard;            // per shard: hard decisions of a caller-resident batch (ldpc_hip_decode_count_multi)
    std::vector<long long> hard_frames;
    std::shared_ptr<ldpc_multi::CommSet> comms;   // null: counters are summed on the host
    std::unique_ptr<ldpc_multi::ShardPool> pool;  // n > 1: one thread per shard
    std::string reduction = "host";
    int test_fail_shard = -1;                  // LDPC_HIP_TEST_FAIL_SHARD: this shard's enqueue fails (tests of the failure path)
    long long mt_sharded_rounds = 0, mt_fallbacks = 0;   // exact replay: generation rounds shared out / redone with the whole tape per shard
};

namespace {

const std::string &thread_error() { return g_err; }

// runs fn(shard) on the shard's own thread and reports the first failure through the calling thread's error string
int for_each_shard(ldpc_hip_multi *m, const std::function<int(int)> &fn) {
    if (!m->pool) {
        const int rc = fn(0);
        if (rc != 0) { const std::string e = g_err; return fail(rc, "shard 0 (device %d): %s", m->device[0], e.c_str()); }
        return 0;
    }
    int who = 0;
    std::string msg;
    const int rc = m->pool->run(fn, &who, &msg, thread_error);
    if (rc != 0) return fail(rc, "shard %d (device %d): %s", who, m->device[(size_t)who], msg.c_str());
    return 0;
}

// waits for whatever the shards have enqueued (used before an error is returned: nothing may still be running on buffers the
// caller is about to release)
void drain_streams(ldpc_hip_multi *m) {
    for (size_t i = 0; i < m->shard.size(); ++i) {
        if (hipSetDevice(m->device[i]) != hipSuccess) continue;
        (void)hipStreamSynchronize(m->stream[i]);
    }
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

// Phase 2 of a counting call: every shard has its five counters in c->w_counters and all its work enqueued.  All-reduce them
// (RCCL, one group call for all ranks, or the host sum for shards that share a device), wait for the streams, hand back the sum.
int multi_reduce(ldpc_hip_multi *m, unsigned long long tot[5]) {
    const int n = (int)m->shard.size();
    std::vector<unsigned long long> host_cnt((size_t)n * 8, 0ull);
    if (m->comms) {   // one 40-byte all-reduce per call over RCCL / xGMI
        std::string err;
        ldpc_multi::Rccl *r = ldpc_multi::rccl(err);
        if (!r) { drain_streams(m); return fail(LDPC_HIP_EHIP, "RCCL went away: %s", err.c_str()); }
        std::lock_guard<std::mutex> turn(m->comms->in_use);
        ncclResult_t nr = r->GroupStart();
        for (int i = 0; i < n && nr == ncclSuccess; ++i)
            nr = r->AllReduce(m->shard[(size_t)i]->w_counters, m->d_red[(size_t)i], 5, ncclUint64, ncclSum, m->comms->comm[(size_t)i], m->stream[(size_t)i]);
        const ncclResult_t ne = r->GroupEnd();   // always closed, also after a failed enqueue
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess) { drain_streams(m); return fail(LDPC_HIP_EHIP, "ncclAllReduce: %s", r->GetErrorString(nr)); }
        for (int i = 0; i < n; ++i) {
            HIP_TRY(hipSetDevice(m->device[(size_t)i]));
            HIP_TRY(hipMemcpyAsync(&host_cnt[(size_t)i * 8], m->d_red[(size_t)i], sizeof(unsigned long long) * 5, hipMemcpyDeviceToHost, m->stream[(size_t)i]));
        }
        for (int i = 0; i < n; ++i) {
            HIP_TRY(hipSetDevice(m->device[(size_t)i]));
            HIP_TRY(hipStreamSynchronize(m->stream[(size_t)i]));
        }
        for (int j = 0; j < 5; ++j) tot[j] = host_cnt[(size_t)j];   // every rank holds the sum; they must agree
        for (int i = 1; i < n; ++i)
            for (int j = 0; j < 5; ++j)
                if (host_cnt[(size_t)i * 8 + j] != tot[j]) return fail(LDPC_HIP_EHIP, "all-reduce result differs between ranks 0 and %d", i);
    } else {
        for (int i = 0; i < n; ++i) {
            HIP_TRY(hipSetDevice(m->device[(size_t)i]));
            HIP_TRY(hipMemcpyAsync(&host_cnt[(size_t)i * 8], m->shard[(size_t)i]->w_counters, sizeof(unsigned long long) * 5, hipMemcpyDeviceToHost, m->stream[(size_t)i]));
        }
        for (int i = 0; i < n; ++i) {
            HIP_TRY(hipSetDevice(m->device[(size_t)i]));
            HIP_TRY(hipStreamSynchronize(m->stream[(size_t)i]));
        }
        for (int j = 0; j < 5; ++j) tot[j] = 0;
        for (int i = 0; i < n; ++i) for (int j = 0; j < 5; ++j) tot[j] += host_cnt[(size_t)i * 8 + j];
    }
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
    // ---- phase 1: enqueue on every shard, join
    int rc = for_each_shard(m, [&](int i) -> int {
        ldpc_hip_ctx *c = m->shard[(size_t)i];
        if (int r = set_device(c)) return r;
        if (i == m->test_fail_shard) return fail(LDPC_HIP_EHIP, "injected failure (LDPC_HIP_TEST_FAIL_SHARD)");
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
        return 0;
    });
    if (rc) { const std::string e = g_err; drain_streams(m); return fail(rc, "%s", e.c_str()); }   // no shard enters the collective
    // ---- phase 2: the all-reduce, for all ranks or for none
    unsigned long long tot[5] = {0, 0, 0, 0, 0};
    if ((rc = multi_reduce(m, tot))) return rc;
    if (records) {   // hand the ordered records back: batch k covers global frames [k*batch, k*batch + nb)
        rc = for_each_shard(m, [&](int i) -> int {
            if (int r = set_device(m->shard[(size_t)i])) return r;
            long long off = 0;
            for (long long k = i; k < nbatches; k += n) {
                const long long nb = (k + 1) * batch <= B ? batch : B - k * batch;
                if (frame_info) HIP_TRY(hipMemcpy(frame_info + k * batch, m->d_info[(size_t)i] + off, sizeof(int32_t) * (size_t)nb, hipMemcpyDeviceToHost));
                if (iters) HIP_TRY(hipMemcpy(iters + k * batch, m->d_iters[(size_t)i] + off, sizeof(int32_t) * (size_t)nb, hipMemcpyDeviceToHost));
                off += nb;
            }
            return 0;
        });
        if (rc) return rc;
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
        m->rec_frames.push_back(0); m->d_hard.push_back(nullptr); m->hard_frames.push_back(0);
        HIP_TRY(hipSetDevice(devices[i]));
        HIP_TRY(hipStreamCreateWithFlags(&m->stream.back(), hipStreamNonBlocking));
        HIP_TRY(hipMalloc(&m->d_red.back(), sizeof(unsigned long long) * 8));
    }
    const char *single = getenv("LDPC_HIP_RCCL_SINGLE");     // tests: run the one-rank communicator through RCCL too
    const char *dup = getenv("LDPC_HIP_RCCL_ALLOW_DUPLICATE");   // tests: logical shards on one device through the communicator path
                                                                 // (only a stand-in library accepts one device twice; real RCCL refuses)
    const bool dup_ok = dup && atoi(dup) != 0;
    if ((distinct || dup_ok) && (n_shards > 1 || (single && atoi(single) != 0))) {
        std::string err;
        m->comms = ldpc_multi::acquire_comms(devices, n_shards, err);
        if (!m->comms) return fail(LDPC_HIP_EHIP, "ldpc_hip_open_multi: %d devices need RCCL for the counter all-reduce: %s", n_shards, err.c_str());
        m->reduction = "rccl";
    }
    if (const char *f = getenv("LDPC_HIP_TEST_FAIL_SHARD")) m->test_fail_shard = atoi(f);
    if (n_shards > 1) m->pool.reset(new ldpc_multi::ShardPool(n_shards));
    *out = m.release();
    return 0;
}

void ldpc_hip_close_multi(ldpc_hip_multi *m) {
    if (!m) return;
    m->pool.reset();      // joins the shard threads
    m->comms.reset();     // the communicators stay in the per-process cache
    for (size_t i = 0; i < m->shard.size(); ++i) {
        (void)hipSetDevice(m->device[i]);
        if (m->stream[i]) { (void)hipStreamSynchronize(m->stream[i]); (void)hipStreamDestroy(m->stream[i]); }
        if (m->d_red[i]) (void)hipFree(m->d_red[i]);
        if (m->d_info[i]) (void)hipFree(m->d_info[i]);
        if (m->d_iters[i]) (void)hipFree(m->d_iters[i]);
        if (m->d_hard[i]) (void)hipFree(m->d_hard[i]);
        ldpc_hip_close(m->shard[i]);
    }
    delete m;
}

int ldpc_hip_multi_shards(const ldpc_hip_multi *m) { return m ? (int)m->shard.size() : 0; }
ldpc_hip_ctx *ldpc_hip_multi_ctx(ldpc_hip_multi *m, int shard) { return (m && shard >= 0 && shard < (int)m->shard.size()) ? m->shard[(size_t)shard] : nullptr; }
const char *ldpc_hip_multi_reduction(const ldpc_hip_multi *m) { return m ? m->reduction.c_str() : ""; }
void *ldpc_hip_multi_stream(ldpc_hip_multi *m, int shard) { return (m && shard >= 0 && shard < (int)m->shard.size()) ? (void *)m->stream[(size_t)shard] : nullptr; }
long long ldpc_hip_multi_comm_inits(void) {
    ldpc_multi::CommCache &cc = ldpc_multi::comm_cache();
    std::lock_guard<std::mutex> lock(cc.mu);
    return cc.inits;
}
void ldpc_hip_multi_release_comms(void) { ldpc_multi::release_all_comms(); }

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

int ldpc_hip_decode_count_multi(ldpc_hip_multi *m, const double *const *d_llr, long long B_per_shard, long long first_frame, int maxiter,
                                double alpha, unsigned long long counters[4], unsigned long long *sum_abs_iters) {
    if (!m || !d_llr || !counters || B_per_shard < 0 || first_frame < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_decode_count_multi: bad argument");
    int rc = for_each_shard(m, [&](int i) -> int {
        ldpc_hip_ctx *c = m->shard[(size_t)i];
        if (int r = set_device(c)) return r;
        if (i == m->test_fail_shard) return fail(LDPC_HIP_EHIP, "injected failure (LDPC_HIP_TEST_FAIL_SHARD)");
        hipStream_t st = m->stream[(size_t)i];
        HIP_TRY(hipMemsetAsync(c->w_counters, 0, sizeof(unsigned long long) * 8, st));
        if (B_per_shard == 0) return 0;
        if (!d_llr[i]) return fail(LDPC_HIP_EINVAL, "ldpc_hip_decode_count_multi: null batch for shard %d", i);
        if (int r = multi_records(m, i, B_per_shard)) return r;
        if (B_per_shard > m->hard_frames[(size_t)i]) {
            if (m->d_hard[(size_t)i]) (void)hipFree(m->d_hard[(size_t)i]);
            m->d_hard[(size_t)i] = nullptr; m->hard_frames[(size_t)i] = 0;
            HIP_TRY(hipMalloc(&m->d_hard[(size_t)i], sizeof(uint32_t) * (size_t)B_per_shard * c->hard_words));
            m->hard_frames[(size_t)i] = B_per_shard;
        }
        uint32_t *hard = m->d_hard[(size_t)i];
        int32_t *its = m->d_iters[(size_t)i];
        if (int r = ldpc_hip_decode_dev(c, d_llr[i], B_per_shard, maxiter, alpha, hard, its, nullptr, st)) return r;
        return ldpc_hip_count_errors_cw_dev(c, hard, its, first_frame + (long long)i * B_per_shard, B_per_shard, nullptr, c->w_counters, st);
    });
    if (rc) { const std::string e = g_err; drain_streams(m); return fail(rc, "%s", e.c_str()); }
    unsigned long long tot[5] = {0, 0, 0, 0, 0};
    if ((rc = multi_reduce(m, tot))) return rc;
    counters[0] = tot[0]; counters[1] = tot[1]; counters[2] = tot[2]; counters[3] = tot[3];
    if (sum_abs_iters) *sum_abs_iters = tot[4];
    return 0;
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

// ---- exact replay over the shards ---------------------------------------------------------------------------------------------
// The generator's stream is sequential, but jump-ahead reaches any point of it: the round's word tape is cut at the EXPECTED positions
// of the shards' first items, every shard makes the sub-streams around its cut (+- 8 standard deviations of where the items can
// lie) and counts the accepted polar-method attempts of its own stretch; the n counts go through the host (a counters-only
// exchange, like the error counters), after which every shard knows the item index its window starts at, emits the decoder inputs
// of its own frames and -- the shard whose window holds the round's last item -- the state the generator is left in, which the
// host copies to the others.  Per-shard generation work is ~1/n of the round.  If an estimate ever fails (an item outside its
// shard's window) the round is redone with every shard generating the whole tape, so correctness never rests on the margins.
}  // extern "C"

namespace {

bool mt_sharding_wanted(const ldpc_hip_multi *m, const MtPlan &pl) {
    const int n = (int)m->shard.size();
    if (n < 2) return false;
    if (const char *e = getenv("LDPC_HIP_MT_SHARDED")) return atoi(e) != 0 && pl.attempts >= 2 * n;
    return pl.attempts >= (long long)n * 16 * pl.margin;   // stretches much longer than the margins around them
}

// One generation round over `frames` frames for all shards (generators in the same state on entry and on return).  part[0..n]:
// frame boundaries of the shards' stretches; shard i writes the rows of its frames into its workspace when `rows` is set.
// *fdone = frames completed (whole frames; a round may come up short).
int multi_mt_round(ldpc_hip_multi *m, double snr_db, int modulation_type, int punctured_blocks, long long frames, const std::vector<long long> &part,
                   bool rows, long long *fdone) {
    using namespace ldpc_mt;
    const int n = (int)m->shard.size();
    ldpc_hip_ctx *c0 = m->shard[0];
    const long long N = c0->N;
    const unsigned long long need = (unsigned long long)frames * (unsigned long long)N;
    const MtPlan pl = mt_plan(c0->mt.pos, need);
    std::vector<PolarArgs> proto((size_t)n);
    auto shard_proto = [&](int i) -> int {
        ldpc_hip_ctx *c = m->shard[(size_t)i];
        if (int rc = mt_frame_proto(c, snr_db, modulation_type, punctured_blocks, proto[(size_t)i])) return rc;
        PolarArgs &a = proto[(size_t)i];
        a.first_frame = c->mt.frames_taken;
        if (rows && part[(size_t)i] < part[(size_t)i + 1]) { a.row_lo = part[(size_t)i]; a.row_hi = part[(size_t)i + 1]; a.out = c->w_llr; }
        else { a.row_lo = 0; a.row_hi = 0; a.out = nullptr; }
        return 0;
    };
    bool sharded = mt_sharding_wanted(m, pl);
    unsigned long long limit = 0;
    if (sharded) {
        // ---- phase A: windows, words, counts
        std::vector<long long> cut((size_t)n + 1);   // attempt index where shard i's stretch is expected to start
        for (int i = 0; i <= n; ++i) {
            const long long at = (long long)((double)part[(size_t)i] * (double)N * 1.2732395447351628);
            cut[(size_t)i] = i == 0 ? 0 : i == n ? pl.attempts : (at < pl.attempts ? at : pl.attempts);
        }
        std::vector<MtWindow> win((size_t)n);
        std::vector<unsigned long long> cnt((size_t)n * 2, 0ull);
        int rc = for_each_shard(m, [&](int i) -> int {
            ldpc_hip_ctx *c = m->shard[(size_t)i];
            if (int r = set_device(c)) return r;
            if (i == m->test_fail_shard) return fail(LDPC_HIP_EHIP, "injected failure (LDPC_HIP_TEST_FAIL_SHARD)");
            if (int r = shard_proto(i)) return r;
            if (cut[(size_t)i] >= cut[(size_t)i + 1]) { win[(size_t)i] = MtWindow(); return 0; }   // nothing to own
            hipStream_t st = m->stream[(size_t)i];
            const long long a_lo = cut[(size_t)i] - pl.margin, a_hi = cut[(size_t)i + 1] + pl.margin;
            MtWindow &w = win[(size_t)i];
            w = mt_window(pl, pl.pos + 4 * (a_lo < 0 ? 0 : a_lo), pl.pos + 4 * (a_hi > pl.attempts ? pl.attempts : a_hi) + MTN);
            if (int r = mt_ensure(c, w)) return r;
            if (int r = mt_generate(c, w, st)) return r;
            if (int r = mt_count(c, pl, w, w.at_lo, cut[(size_t)i], cut[(size_t)i + 1], st)) return r;
            HIP_TRY(hipMemcpyAsync(&cnt[(size_t)i * 2], c->mt.d_counters, sizeof(unsigned long long) * 2, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            return 0;
        });
        if (rc) { const std::string e = g_err; drain_streams(m); return fail(rc, "%s", e.c_str()); }
        for (int i = 0; i < n && sharded; ++i)   // a window must reach from its cut's margin to the next cut (it does unless the tape ended early)
            if (win[(size_t)i].S > 0 && (win[(size_t)i].at_lo > cut[(size_t)i] || win[(size_t)i].at_hi < cut[(size_t)i + 1])) sharded = false;
        if (sharded) {
            // ---- the exchange: n counts -> item index in front of every stretch, the round's total, what the round keeps
            std::vector<unsigned long long> before((size_t)n + 1, 0ull);
            for (int i = 0; i < n; ++i) before[(size_t)i + 1] = before[(size_t)i] + cnt[(size_t)i * 2 + 1];
            const unsigned long long total = before[(size_t)n];
            limit = total < need ? total : need;
            limit -= limit % (unsigned long long)N;
            // ---- phase B: emit, find the end
            std::vector<unsigned long long> tot((size_t)n * 2, 0ull);
            std::vector<long long> endt((size_t)n * 2, 0ll);
            rc = for_each_shard(m, [&](int i) -> int {
                ldpc_hip_ctx *c = m->shard[(size_t)i];
                const MtWindow &w = win[(size_t)i];
                if (w.S == 0) return 0;
                if (int r = set_device(c)) return r;
                hipStream_t st = m->stream[(size_t)i];
                const unsigned long long base0 = before[(size_t)i] - cnt[(size_t)i * 2];   // minus the accepted attempts of the left margin
                if (int r = mt_emit(c, pl, w, proto[(size_t)i], base0, st)) return r;
                if (int r = mt_finish(c, pl, w, proto[(size_t)i], base0, (long long)limit, st)) return r;
                HIP_TRY(hipMemcpyAsync(&tot[(size_t)i * 2], c->mt.d_total, sizeof(unsigned long long) * 2, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipMemcpyAsync(&endt[(size_t)i * 2], c->mt.d_end_t, sizeof(long long) * 2, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                return 0;
            });
            if (rc) { const std::string e = g_err; drain_streams(m); return fail(rc, "%s", e.c_str()); }
            // ---- every shard's items inside its window?  who holds the end?
            int owner = -1;
            for (int i = 0; i < n; ++i) {
                const MtWindow &w = win[(size_t)i];
                if (w.S > 0 && endt[(size_t)i * 2 + 1] == 1 && owner < 0) owner = i;
                const unsigned long long lo_item = (unsigned long long)part[(size_t)i] * (unsigned long long)N;
                unsigned long long hi_item = (unsigned long long)part[(size_t)i + 1] * (unsigned long long)N;
                if (hi_item > limit) hi_item = limit;
                if (!rows || lo_item >= hi_item) continue;
                const unsigned long long base0 = before[(size_t)i] - cnt[(size_t)i * 2];
                if (w.S == 0 || base0 > lo_item || base0 + tot[(size_t)i * 2] < hi_item) sharded = false;
            }
            if (owner < 0) sharded = false;
            if (sharded) {   // the state the round ends in: from the shard that found it to all
                uint32_t st_words[MTN];
                ldpc_hip_ctx *co = m->shard[(size_t)owner];
                HIP_TRY(hipSetDevice(co->device));
                HIP_TRY(hipMemcpy(st_words, co->mt.d_state_next, sizeof st_words, hipMemcpyDeviceToHost));
                for (int i = 0; i < n; ++i) {
                    ldpc_hip_ctx *c = m->shard[(size_t)i];
                    HIP_TRY(hipSetDevice(c->device));
                    HIP_TRY(hipMemcpy(c->mt.d_state, st_words, sizeof st_words, hipMemcpyHostToDevice));
                    c->mt.pos = 0;
                }
            }
        }
        if (!sharded) ++m->mt_fallbacks;
        else ++m->mt_sharded_rounds;
    }
    if (!sharded) {   // every shard generates the whole tape and emits its rows
        std::vector<unsigned long long> emitted((size_t)n, 0ull);
        const int rc = for_each_shard(m, [&](int i) -> int {
            ldpc_hip_ctx *c = m->shard[(size_t)i];
            if (int r = set_device(c)) return r;
            if (int r = shard_proto(i)) return r;
            return mt_round(c, need, proto[(size_t)i], &emitted[(size_t)i], m->stream[(size_t)i]);
        });
        if (rc) return rc;
        limit = emitted[0];
        for (int i = 1; i < n; ++i) if (emitted[(size_t)i] != limit) return fail(LDPC_HIP_EHIP, "exact replay: shards 0 and %d disagree on the round (%llu vs %llu items)", i, limit, emitted[(size_t)i]);
    }
    *fdone = (long long)(limit / (unsigned long long)N);
    for (ldpc_hip_ctx *c : m->shard) c->mt.frames_taken += *fdone;
    return 0;
}

// the next B frames of the stream: generated (sharded), and -- decode == true -- decoded and counted on the shard that holds their rows;
// records in global frame order
int multi_mt_frames(ldpc_hip_multi *m, double snr_db, int modulation_type, int punctured_blocks, int maxiter, double alpha, long long B, bool decode,
                    int32_t *frame_info, int32_t *iters) {
    const int n = (int)m->shard.size();
    ldpc_hip_ctx *c0 = m->shard[0];
    for (ldpc_hip_ctx *c : m->shard) {
        if (!c->mt.set) return fail(LDPC_HIP_EINVAL, "the generator has no state: call ldpc_hip_mt_set_state_multi first");
        if (c->mt.pos != c0->mt.pos || c->mt.frames_taken != c0->mt.frames_taken) return fail(LDPC_HIP_EINVAL, "the shards' generators are out of step");
    }
    // BP_DEC with the frame chain on is sequential by definition (decoders.cpp:1742-1762): shard 0 decodes every frame
    const bool chained = decode && c0->decoder_id == LDPC_HIP_BP_DEC && c0->bp_chain;
    long long per_round = mt_frames_per_round(c0);
    if (per_round > (1 << 16)) per_round = 1 << 16;
    struct Seg { long long first, count, off; };
    std::vector<std::vector<Seg>> segs((size_t)n);
    std::vector<long long> used((size_t)n, 0);   // record slots taken per shard
    if (decode) {
        const int rc = for_each_shard(m, [&](int i) -> int {   // record buffers: at most ceil(B / n) + one round's share per shard ... B bounds it
            ldpc_hip_ctx *c = m->shard[(size_t)i];
            if (int r = set_device(c)) return r;
            const long long cap = chained ? (i == 0 ? B : 0) : (B / n + per_round / n + 2 * ((B + per_round - 1) / per_round) + 2);
            ldpc_mt::DeviceState &ms = c->mt;
            if (cap > ms.cap_rec) {
                if (ms.d_info) (void)hipFree(ms.d_info);
                if (ms.d_iters) (void)hipFree(ms.d_iters);
                ms.d_info = nullptr; ms.d_iters = nullptr; ms.cap_rec = 0;
                HIP_TRY(hipMalloc(&ms.d_info, sizeof(int32_t) * (size_t)cap));
                HIP_TRY(hipMalloc(&ms.d_iters, sizeof(int32_t) * (size_t)cap));
                ms.cap_rec = cap;
            }
            HIP_TRY(hipMemsetAsync(c->w_counters, 0, sizeof(unsigned long long) * 8, m->stream[(size_t)i]));
            return 0;
        });
        if (rc) return rc;
    }
    long long done = 0;
    int stalled = 0;
    std::vector<long long> part((size_t)n + 1);
    while (done < B) {
        const long long fr = B - done < per_round ? B - done : per_round;
        for (int i = 0; i <= n; ++i) part[(size_t)i] = chained ? (i == 0 ? 0 : fr) : fr * i / n;
        if (decode) {
            const int rc = for_each_shard(m, [&](int i) -> int {
                const long long mine = part[(size_t)i + 1] - part[(size_t)i];
                if (mine <= 0) return 0;
                if (int r = set_device(m->shard[(size_t)i])) return r;
                return ensure_workspace(m->shard[(size_t)i], mine, false);
            });
            if (rc) return rc;
        }
        const long long first = c0->mt.frames_taken;
        long long fdone = 0;
        if (int rc = multi_mt_round(m, snr_db, modulation_type, punctured_blocks, fr, part, decode, &fdone)) return rc;
        if (decode && fdone > 0) {
            const int rc = for_each_shard(m, [&](int i) -> int {
                ldpc_hip_ctx *c = m->shard[(size_t)i];
                const long long r_lo = part[(size_t)i], r_hi = part[(size_t)i + 1] < fdone ? part[(size_t)i + 1] : fdone;
                if (r_lo >= r_hi) return 0;
                if (int r = set_device(c)) return r;
                hipStream_t st = m->stream[(size_t)i];
                const long long off = used[(size_t)i], rows = r_hi - r_lo;
                if (off + rows > c->mt.cap_rec) return fail(LDPC_HIP_EHIP, "exact replay: record buffer of shard %d too small", i);
                int32_t *it = c->mt.d_iters + off;
                if (int r = ldpc_hip_decode_dev(c, c->w_llr, rows, maxiter, alpha, c->w_hard, it, nullptr, st)) return r;
                if (int r = ldpc_hip_count_errors_cw_dev(c, c->w_hard, it, first + r_lo, rows, c->mt.d_info + off, c->w_counters, st)) return r;
                segs[(size_t)i].push_back(Seg{done + r_lo, rows, off});
                used[(size_t)i] += rows;   // (the next round's emit pass follows on the same stream: the workspace rows are safe)
                return 0;
            });
            if (rc) return rc;
        }
        done += fdone;
        stalled = fdone == 0 ? stalled + 1 : 0;
        if (stalled >= 2) return fail(LDPC_HIP_EHIP, "the exact-replay generator made no progress (frame of %d samples)", c0->N);
    }
    if (decode) {
        const int rc = for_each_shard(m, [&](int i) -> int {
            ldpc_hip_ctx *c = m->shard[(size_t)i];
            if (int r = set_device(c)) return r;
            HIP_TRY(hipStreamSynchronize(m->stream[(size_t)i]));   // the shard's streams do not synchronise with the copies below
            for (const Seg &sg : segs[(size_t)i]) {
                HIP_TRY(hipMemcpy(frame_info + sg.first, c->mt.d_info + sg.off, sizeof(int32_t) * (size_t)sg.count, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(iters + sg.first, c->mt.d_iters + sg.off, sizeof(int32_t) * (size_t)sg.count, hipMemcpyDeviceToHost));
            }
            return 0;
        });
        if (rc) return rc;
    }
    return 0;
}

}  // namespace

extern "C" {

int ldpc_hip_mt_set_state_multi(ldpc_hip_multi *m, const uint32_t state[624], int pos) {
    if (!m) return fail(LDPC_HIP_EINVAL, "null multi context");
    for (ldpc_hip_ctx *c : m->shard) if (int rc = ldpc_hip_mt_set_state(c, state, pos)) return rc;
    return 0;
}

int ldpc_hip_mt_get_state_multi(ldpc_hip_multi *m, uint32_t state[624], int *pos) {
    if (!m) return fail(LDPC_HIP_EINVAL, "null multi context");
    return ldpc_hip_mt_get_state(m->shard[0], state, pos);
}

int ldpc_hip_mt_set_frame_index_multi(ldpc_hip_multi *m, long long frames_taken) {
    if (!m) return fail(LDPC_HIP_EINVAL, "null multi context");
    for (ldpc_hip_ctx *c : m->shard) if (int rc = ldpc_hip_mt_set_frame_index(c, frames_taken)) return rc;
    return 0;
}

int ldpc_hip_mt_advance_multi(ldpc_hip_multi *m, double snr_db, int modulation_type, int punctured_blocks, long long B) {
    if (!m || B < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_advance_multi: bad argument");
    return multi_mt_frames(m, snr_db, modulation_type, punctured_blocks, 1, 0.8, B, false, nullptr, nullptr);
}

int ldpc_hip_mt_frames_multi(ldpc_hip_multi *m, double snr_db, int modulation_type, int punctured_blocks, int maxiter, double alpha,
                             long long B, int32_t *frame_info, int32_t *iters) {
    if (!m || B < 0 || !frame_info || !iters) return fail(LDPC_HIP_EINVAL, "ldpc_hip_mt_frames_multi: bad argument");
    return multi_mt_frames(m, snr_db, modulation_type, punctured_blocks, maxiter, alpha, B, true, frame_info, iters);
}

/* rounds that ran sharded / that fell back to every shard generating the whole tape, since the multi context was opened */
void ldpc_hip_multi_mt_stats(const ldpc_hip_multi *m, long long *sharded_rounds, long long *fallback_rounds) {
    if (sharded_rounds) *sharded_rounds = m ? m->mt_sharded_rounds : 0;
    if (fallback_rounds) *fallback_rounds = m ? m->mt_fallbacks : 0;
}

}  // extern "C"
