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

// ---- exact replay over the shards: every shard runs the SAME generator over the whole batch (the stream is sequential by nature)
// and decodes its contiguous slice of the frames; no exchange at all.
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
