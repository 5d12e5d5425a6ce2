// ldpc_jit.hpp -- just-in-time instances of the code-specialised min-sum kernel (ldpc_spec.hpp) via hiprtc.
//
// ldpc_hip_open() of an M = 64 min-sum code that is not the ahead-of-time instance writes the constexpr `Code`
// tables of the opened base matrix as C++ source, compiles ldpc_spec.hpp against them for gfx950 and keeps the
// code object in a per-process cache keyed by (device, base matrix).  hiprtc is loaded lazily with dlopen; when it
// (or the header next to the library) is unavailable the caller keeps using the table-driven kernel
// (ldpc_ms_fast.hpp) -- still a HIP kernel, never a CPU path.
#pragma once

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <fstream>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace ldpc_jit {

struct Kernel {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
};

// minimal hiprtc surface (resolved with dlsym so that libldpc_hip.so has no link-time dependency on it)
typedef struct _hiprtcProgram *hiprtcProgram;
struct Rtc {
    void *lib = nullptr;
    int (*CreateProgram)(hiprtcProgram *, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*CompileProgram)(hiprtcProgram, int, const char **) = nullptr;
    int (*GetCodeSize)(hiprtcProgram, size_t *) = nullptr;
    int (*GetCode)(hiprtcProgram, char *) = nullptr;
    int (*GetProgramLogSize)(hiprtcProgram, size_t *) = nullptr;
    int (*GetProgramLog)(hiprtcProgram, char *) = nullptr;
    int (*DestroyProgram)(hiprtcProgram *) = nullptr;
};

inline Rtc *rtc(std::string &err) {
    static Rtc r;
    static bool tried = false;
    static std::mutex rtc_mu;
    std::lock_guard<std::mutex> rtc_lk(rtc_mu);
    if (tried) { if (!r.lib) err = "hiprtc not available"; return r.lib ? &r : nullptr; }
    tried = true;
    std::vector<std::string> names;
    if (const char *p = getenv("LDPC_HIP_HIPRTC_PATH")) names.push_back(p);
    names.push_back("libhiprtc.so");
    names.push_back("libhiprtc.so.7");
    names.push_back("/opt/rocm/lib/libhiprtc.so");
    for (const auto &n : names) {
        r.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) { err = "dlopen(libhiprtc.so) failed"; return nullptr; }
#define LDPC_RTC_SYM(f)                                                         \
    r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.lib, "hiprtc" #f));           \
    if (!r.f) { err = "hiprtc" #f " missing"; r.lib = nullptr; return nullptr; }
    LDPC_RTC_SYM(CreateProgram) LDPC_RTC_SYM(CompileProgram) LDPC_RTC_SYM(GetCodeSize) LDPC_RTC_SYM(GetCode)
    LDPC_RTC_SYM(GetProgramLogSize) LDPC_RTC_SYM(GetProgramLog) LDPC_RTC_SYM(DestroyProgram)
#undef LDPC_RTC_SYM
    return &r;
}

// `struct Code` for ms_m64_body<>: rows[j] = list of (block column, shift) in ascending column order
inline std::string code_struct(const std::vector<std::vector<std::pair<int, int>>> &rows, int nh, int M) {
    const int rh = (int)rows.size();
    size_t wmax = 1;
    for (auto &r : rows) wmax = r.size() > wmax ? r.size() : wmax;
    std::vector<char> seen(nh, 0);
    std::ostringstream col, sh, fi, rw;
    for (int j = 0; j < rh; ++j) {
        col << (j ? ", {" : "{"); sh << (j ? ", {" : "{"); fi << (j ? ", {" : "{");
        rw << (j ? ", " : "") << rows[j].size();
        for (size_t s = 0; s < wmax; ++s) {
            const bool has = s < rows[j].size();
            const int k = has ? rows[j][s].first : 0, c = has ? rows[j][s].second : 0;
            int first = 0;
            if (has && !seen[k]) { first = 1; seen[k] = 1; }
            col << (s ? ", " : "") << k; sh << (s ? ", " : "") << c; fi << (s ? ", " : "") << first;
        }
        col << "}"; sh << "}"; fi << "}";
    }
    std::ostringstream o;
    o << "struct Code {\n"
      << "    static constexpr int RH = " << rh << ", NH = " << nh << ", M = " << M << ", WMAX = " << wmax << ";\n"
      << "    static constexpr int RW[" << rh << "] = {" << rw.str() << "};\n"
      << "    static constexpr int COL[" << rh << "][" << wmax << "] = {" << col.str() << "};\n"
      << "    static constexpr int SH[" << rh << "][" << wmax << "] = {" << sh.str() << "};\n"
      << "    static constexpr int FIRST[" << rh << "][" << wmax << "] = {" << fi.str() << "};\n"
      << "};\n";
    return o.str();
}

inline std::string this_library_dir() {
    Dl_info info;
    if (!dladdr(reinterpret_cast<void *>(&this_library_dir), &info) || !info.dli_fname) return ".";
    std::string p = info.dli_fname;
    const size_t k = p.find_last_of('/');
    return k == std::string::npos ? "." : p.substr(0, k);
}

// Returns the cached or freshly compiled kernel for (device, decoder body, rows, M); nullptr + err on failure.
// body: "ms_m64_body" (needs M == 64, 64 threads per frame) or "lms_body" (ceil(M/64)*64 threads per frame).
// compile == false: answer from the process cache or the on-disk cache only (nullptr, err empty when neither has it).
inline const Kernel *get(int device, const char *body, const std::vector<std::vector<std::pair<int, int>>> &rows, int nh,
                         int M, std::string &err, bool compile = true) {
    static std::mutex mu;
    static std::map<std::string, Kernel> &cache = *new std::map<std::string, Kernel>();   // never destroyed: a background compile may outlive main()
    const std::string code = code_struct(rows, nh, M);
    const bool eight_waves = std::string(body) == "sp_body" || std::string(body) == "asp_body" || std::string(body) == "bp_body";  // 8 waves per frame, 2 frames per CU
    const bool tasp = std::string(body) == "tasp_body";   // two lanes per check; two waves per SIMD while the Z halves + addresses + ~85 temporaries fit 256 registers
    int tasp_regs = 85;
    for (const auto &r : rows) tasp_regs += 2 * (((int)r.size() + 1) / 2) + (((int)r.size() + 1) / 2 + 1) / 2;
    const int threads = eight_waves ? 512 : std::string(body) == "ms_chunk_body" ? 64 : tasp ? ((2 * M + 63) / 64) * 64 : ((M + 63) / 64) * 64;
    const std::string key = std::to_string(device) + "|" + body + "|" + code;
    // the lock covers the process cache only: a compile takes seconds and must not hold up other contexts
    auto lookup = [&]() -> const Kernel * {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        return it != cache.end() ? &it->second : nullptr;
    };
    auto publish = [&](const Kernel &k) -> const Kernel * {
        std::lock_guard<std::mutex> lk(mu);
        auto ins = cache.emplace(key, k);   // a concurrent compile of the same instance: the first one stays ...
        if (!ins.second && k.mod && k.mod != ins.first->second.mod) (void)hipModuleUnload(k.mod);   // ... and the loser's module is released
        return &ins.first->second;
    };
    if (const Kernel *k = lookup()) return k;

    const std::string hdr_path = this_library_dir() + "/csrc/ldpc_spec.hpp";
    std::ifstream hf(hdr_path);
    if (!hf) { err = "cannot read " + hdr_path; return nullptr; }
    std::stringstream hs;
    hs << hf.rdbuf();
    const std::string hdr = hs.str();
    const std::string src = "#include \"ldpc_spec.hpp\"\nnamespace {\n" + code +
                            "}\nextern \"C\" __global__ void __launch_bounds__(" + std::to_string(threads) + (eight_waves ? ", 4" : ((tasp && tasp_regs > 256) || std::string(body) == "ms_chunk_body") ? ", 1" : ", 2") + ") spec_jit(const ldpc_spec::SpecArgs a) {\n"
                            "    ldpc_spec::" + body + "<Code>(a);\n}\n";
    // On-disk cache of compiled code objects: LDPC_HIP_CACHE_DIR, default $XDG_CACHE_HOME/ldpc_hip or $HOME/.cache/ldpc_hip,
    // "off" disables it.  The file name is a hash of everything the object depends on -- the generated source, this header's
    // text and the target -- so a stale entry cannot be picked up; a later process loads the object in milliseconds instead of
    // compiling for seconds.
    std::string cache_file;
    std::string cache_dir;
    if (const char *dir = getenv("LDPC_HIP_CACHE_DIR")) cache_dir = dir;
    else if (const char *x = getenv("XDG_CACHE_HOME")) cache_dir = std::string(x) + "/ldpc_hip";
    else if (const char *h = getenv("HOME")) cache_dir = std::string(h) + "/.cache/ldpc_hip";
    if (cache_dir == "off" || cache_dir == "0") cache_dir.clear();
    if (!cache_dir.empty()) {
        for (size_t i = 1; i <= cache_dir.size(); ++i)     // mkdir -p, best effort
            if (i == cache_dir.size() || cache_dir[i] == '/') (void)mkdir(cache_dir.substr(0, i).c_str(), 0755);
        const char *dir = cache_dir.c_str();
        unsigned long long h = 1469598103934665603ull;   // FNV-1a 64
        auto mix = [&](const std::string &t) { for (unsigned char ch : t) { h ^= ch; h *= 1099511628211ull; } };
        mix(src); mix(hdr); mix("gfx950 -O3 -ffp-contract=off");
        // ... and on who compiled it for what: the runtime / compiler version and the device's own architecture string, so that an
        // upgrade of ROCm (or another GPU in the box) never picks up an object made by an older compiler
        int rt_version = 0;
        (void)hipRuntimeGetVersion(&rt_version);
        mix("rt" + std::to_string(rt_version));
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) mix(prop.gcnArchName);
        char name[64];
        snprintf(name, sizeof name, "/ldpc_spec_%016llx.hsaco", h);
        cache_file = std::string(dir) + name;
        std::ifstream cf(cache_file, std::ios::binary);
        if (cf) {
            std::vector<char> bin((std::istreambuf_iterator<char>(cf)), std::istreambuf_iterator<char>());
            Kernel k;
            if (!bin.empty() && hipSetDevice(device) == hipSuccess && hipModuleLoadData(&k.mod, bin.data()) == hipSuccess &&
                hipModuleGetFunction(&k.fn, k.mod, "spec_jit") == hipSuccess)
                return publish(k);
            (void)hipGetLastError();   // unreadable / stale file: compile again below and overwrite it
        }
    }
    if (!compile) return nullptr;
    Rtc *r = rtc(err);
    if (!r) return nullptr;
    hiprtcProgram prog = nullptr;
    const char *hdr_src[] = {hdr.c_str()};
    const char *hdr_name[] = {"ldpc_spec.hpp"};
    if (r->CreateProgram(&prog, src.c_str(), "ldpc_spec_jit.hip", 1, hdr_src, hdr_name) != 0) { err = "hiprtcCreateProgram failed"; return nullptr; }
    // -ffp-contract=off is part of the numerics contract (two roundings in y + s*alpha)
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"};
    const int rc = r->CompileProgram(prog, 4, opts);
    if (rc != 0) {
        size_t n = 0;
        r->GetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) r->GetProgramLog(prog, &log[0]);
        err = "hiprtcCompileProgram failed: " + log.substr(0, 400);
        r->DestroyProgram(&prog);
        return nullptr;
    }
    size_t n = 0;
    r->GetCodeSize(prog, &n);
    std::vector<char> bin(n);
    r->GetCode(prog, bin.data());
    r->DestroyProgram(&prog);
    Kernel k;
    if (hipSetDevice(device) != hipSuccess || hipModuleLoadData(&k.mod, bin.data()) != hipSuccess ||
        hipModuleGetFunction(&k.fn, k.mod, "spec_jit") != hipSuccess) {
        err = std::string("loading the JIT code object failed: ") + hipGetErrorString(hipGetLastError());
        return nullptr;
    }
    if (!cache_file.empty()) {   // best effort, atomically: a concurrent process either sees the whole file or none
        static std::atomic<unsigned> tmp_serial{0};   // the foreground open and the background worker may write the same key at once
        const std::string tmp = cache_file + ".tmp" + std::to_string((long long)getpid()) + "." + std::to_string(tmp_serial.fetch_add(1));
        std::ofstream of(tmp, std::ios::binary);
        if (of && of.write(bin.data(), (std::streamsize)bin.size()) && (of.close(), true)) {
            if (rename(tmp.c_str(), cache_file.c_str()) != 0) (void)remove(tmp.c_str());
        }
    }
    return publish(k);
}

// ---- background specialisation ------------------------------------------------------------------------------------------
// A context whose decoder has another HIP tier (table-driven or shape-unlimited kernel) need not wait seconds for hiprtc: it
// starts on that tier, a single worker thread compiles the instance, and the context moves over at its next launch after the
// instance is ready (every tier gives identical bits).  A code search opens thousands of short-lived contexts: jobs whose context
// was closed before their turn are dropped; at exit the process waits for the one compile in flight, not for the queue.
struct Job {
    int device = 0, nh = 0, M = 0;
    std::string body;
    std::vector<std::vector<std::pair<int, int>>> rows;
    std::atomic<int> state{0};          // 0 queued / compiling, 1 ready, 2 failed or dropped
    std::atomic<bool> cancelled{false};
    const Kernel *kernel = nullptr;
    std::string err;
};

class Worker {
public:
    static Worker &instance() {
        static Worker *w = new Worker();   // never destroyed; stop() runs from atexit
        return *w;
    }
    void submit(const std::shared_ptr<Job> &j) {
        std::lock_guard<std::mutex> lk(mu_);
        if (!started_) {
            started_ = true;
            // hiprtc dlopen()s its compiler (comgr) at its first compile, and exit() runs handlers and library destructors newest
            // first: compile something trivial NOW, so that the compiler's destructors are registered BEFORE the handler below --
            // the handler then runs first and lets a compile that is in flight at exit finish while its compiler is still alive
            warm_up();
            th_ = std::thread([this] { run(); });
            atexit([] { Worker::instance().stop(); });
        }
        q_.push_back(j);
        cv_.notify_one();
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
            cv_.notify_one();
        }
        if (th_.joinable()) th_.join();
    }

private:
    static void warm_up() {
        std::string err;
        Rtc *r = rtc(err);
        if (!r) return;
        hiprtcProgram prog = nullptr;
        if (r->CreateProgram(&prog, "extern \"C\" __global__ void ldpc_jit_warm_up() {}\n", "warm_up.hip", 0, nullptr, nullptr) != 0) return;
        const char *opts[] = {"--offload-arch=gfx950", "-O1"};
        (void)r->CompileProgram(prog, 2, opts);
        r->DestroyProgram(&prog);
    }
    void run() {
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                if (stop_) return;
                j = q_.back();   // newest first: its context is the one most likely still open
                q_.pop_back();
            }
            if (j->cancelled.load()) { j->state.store(2); continue; }
            j->kernel = get(j->device, j->body.c_str(), j->rows, j->nh, j->M, j->err, true);
            j->state.store(j->kernel ? 1 : 2, std::memory_order_release);
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<Job>> q_;
    std::thread th_;
    bool started_ = false, stop_ = false;
};

}  // namespace ldpc_jit
