// ldpc_hip.hip -- C-ABI (include/ldpc_hip.h) over the gfx950 kernels.  Host side: context, code tables,
// launch geometry, workspaces, HIP-event timing.  No CPU decode path exists in this library: every entry
// point either runs the HIP kernels or fails with a message.
#include "../../include/ldpc_hip.h"
#include "../../include/ldpc/interleaver.h"
#include "../../include/ldpc/encoder.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

#include "ldpc_frontend.hpp"
#include "ldpc_jit.hpp"
#include "ldpc_kernels.hpp"
#include "ldpc_global.hpp"
#include "ldpc_ms_fast.hpp"
#include "ldpc_aot.hpp"   // ldpc_spec.hpp, code_appendix_c_m64.hpp + the declarations of the aot/*.hip kernels
#include "ldpc_sumprod.hpp"
#include "ldpc_mt.hpp"
#include "ldpc_encode.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(LDPC_HIP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

constexpr int kRHM = 16;   // block rows held in VGPRs by the min-sum kernels
constexpr int kNHM = 32;   // block columns whose channel LLRs the flooding kernel keeps in VGPRs
constexpr int kRWM = 16;   // max circulants per block row (edge-sign bits per row word)

}  // namespace

// ---- code-specialised instances (ldpc_spec.hpp): the ahead-of-time ones are defined in aot/*.hip and declared in ldpc_aot.hpp;
// everything else is compiled at ldpc_hip_open() with hiprtc from the same header (ldpc_jit.hpp).
namespace {

// the opened base matrix as edge lists (row-major = upstream's j-then-k loop order)
struct CodeTables {
    int rh = 0, nh = 0, M = 0, ne = 0, max_rw = 0, min_rw = 1 << 30, max_cw = 0;
    bool all_cols_used = true;
    std::vector<int32_t> row_start, col_start;
    std::vector<uint32_t> edges;      // (block column << 16) | shift
    std::vector<uint32_t> col_edges;  // (block row << 16) | shift, column-major
    std::vector<uint32_t> col_slot;   // row-major edge id of the column-major entries
    std::vector<uint32_t> edge_row;   // block row of each row-major edge

    CodeTables(int rh_, int nh_, int M_, const int16_t *hd) : rh(rh_), nh(nh_), M(M_), row_start(rh_ + 1), col_start(nh_ + 1) {
        for (int j = 0; j < rh; ++j) {
            row_start[j] = (int32_t)edges.size();
            for (int k = 0; k < nh; ++k) {
                int v = hd[j * nh + k];
                if (v == -1) continue;
                while (v < 0) v += M;   // rotate() reduces the shift like this (decoders.cpp:335-339)
                while (v >= M) v -= M;
                edges.push_back(((uint32_t)k << 16) | (uint32_t)v);
                edge_row.push_back((uint32_t)j);
            }
            const int rw = (int)edges.size() - row_start[j];
            max_rw = rw > max_rw ? rw : max_rw;
            min_rw = rw < min_rw ? rw : min_rw;
        }
        row_start[rh] = (int32_t)edges.size();
        ne = (int)edges.size();
        for (int k = 0; k < nh; ++k) {
            col_start[k] = (int32_t)col_edges.size();
            for (int e = 0; e < ne; ++e)
                if ((int)(edges[e] >> 16) == k) {     // row-major order == rows ascending inside a column
                    col_edges.push_back((edge_row[e] << 16) | (edges[e] & 0xffffu));
                    col_slot.push_back((uint32_t)e);
                }
            const int cw = (int)col_edges.size() - col_start[k];
            max_cw = cw > max_cw ? cw : max_cw;
            all_cols_used = all_cols_used && cw > 0;
        }
        col_start[nh] = (int32_t)col_edges.size();
    }
    std::vector<std::vector<std::pair<int, int>>> rows() const {
        std::vector<std::vector<std::pair<int, int>>> r(rh);
        for (int e = 0; e < ne; ++e) r[edge_row[e]].emplace_back((int)(edges[e] >> 16), (int)(edges[e] & 0xffffu));
        return r;
    }
    template <class FC>
    bool is() const {  // does the opened matrix equal the constexpr tables FC?
        bool same = rh == FC::RH && nh == FC::NH && M == FC::M;
        for (int j = 0; same && j < rh; ++j) {
            same = (row_start[j + 1] - row_start[j]) == FC::RW[j];
            for (int e = row_start[j]; same && e < row_start[j + 1]; ++e)
                same = (int)(edges[e] >> 16) == FC::COL[j][e - row_start[j]] && (int)(edges[e] & 0xffffu) == FC::SH[j][e - row_start[j]];
        }
        return same;
    }
};

struct AotInstance {
    int decoder;
    const void *fn;
    int threads;
    const char *name;
    bool (CodeTables::*matches)() const;
};
const AotInstance kAot[] = {
    {LDPC_HIP_MS_DEC, (const void *)ms_spec_appendix_c_m64_kernel, 64, "ms_spec_appendix_c_m64_kernel", &CodeTables::is<ldpc_spec::CodeAppendixCM64>},
    {LDPC_HIP_MS_DEC, (const void *)ms_spec_appendix_c_m126_kernel, 128, "ms_spec_appendix_c_m126_kernel", &CodeTables::is<ldpc_spec::CodeAppendixCM126>},
    {LDPC_HIP_MS_DEC, (const void *)ms_chunk_appendix_c_m126_kernel, 64, "ms_chunk_appendix_c_m126_kernel", &CodeTables::is<ldpc_spec::CodeAppendixCM126>},
    {LDPC_HIP_MS_DEC, (const void *)ms_spec_appendix_c_m512_kernel, 512, "ms_spec_appendix_c_m512_kernel", &CodeTables::is<ldpc_spec::CodeAppendixCM512>},
    {LDPC_HIP_IMS_DEC, (const void *)ims_spec_appendix_c_m64_kernel, 64, "ims_spec_appendix_c_m64_kernel", &CodeTables::is<ldpc_spec::CodeAppendixCM64>},
    {LDPC_HIP_IMS_DEC, (const void *)ims_spec_appendix_c_m126_kernel, 128, "ims_spec_appendix_c_m126_kernel", &CodeTables::is<ldpc_spec::CodeAppendixCM126>},
    {LDPC_HIP_LMS_DEC, (const void *)lms_spec_appendix_c_m64_kernel, 64, "lms_spec_appendix_c_m64_kernel", &CodeTables::is<ldpc_spec::CodeAppendixCM64>},
    {LDPC_HIP_LMS_DEC, (const void *)lms_spec_appendix_c_m512_kernel, 512, "lms_spec_appendix_c_m512_kernel", &CodeTables::is<ldpc_spec::CodeAppendixCM512>},
    {LDPC_HIP_SP_DEC, (const void *)sp_spec_appendix_c_m64_kernel, 64 * ldpc_spec::kSpBodyWaves, "sp_spec_appendix_c_m64_kernel", &CodeTables::is<ldpc_spec::CodeAppendixCM64>},
    {LDPC_HIP_BP_DEC, (const void *)bp_spec_appendix_c_m64_kernel, 512, "bp_spec_appendix_c_m64_kernel", &CodeTables::is<ldpc_spec::CodeAppendixCM64>},
    {LDPC_HIP_ASP_DEC, (const void *)asp_spec_appendix_c_m64_kernel, 512, "asp_spec_appendix_c_m64_kernel", &CodeTables::is<ldpc_spec::CodeAppendixCM64>},
    {LDPC_HIP_TASP_DEC, (const void *)tasp_spec_appendix_c_m64_kernel, 128, "tasp_spec_appendix_c_m64_kernel", &CodeTables::is<ldpc_spec::CodeAppendixCM64>},
    {LDPC_HIP_TASP_DEC, (const void *)tasp_spec_appendix_c_m126_kernel, 256, "tasp_spec_appendix_c_m126_kernel", &CodeTables::is<ldpc_spec::CodeAppendixCM126>},
};

}  // namespace

// What the probability-domain decoders leave in their input array (decoders.cpp:2611-2618): P(bit = 1) of the channel,
// with the decoder's own exp (glibc's algorithm on the device).
__global__ void __launch_bounds__(256) channel_prior_kernel(double *x, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double h = x[i] * 0.5;
        const double y = h < 20.0 ? (h < -20.0 ? -20.0 : h) : 20.0;
        const double e0 = ldpc_spec::exp_glibc(y), e1 = ldpc_spec::exp_glibc(-y);
        x[i] = e1 / (e0 + e1);
    }
}

struct ldpc_hip_ctx {
    int decoder_id = 0, device = 0;
    int rh = 0, nh = 0, M = 0, N = 0, R = 0, ne = 0, hard_words = 0;
    // generic (table-driven) kernel geometry
    int F = 1;         // frames per workgroup (M <= 64: floor(64/M))
    int threads = 64;  // workgroup size
    bool multiwave = false;
    bool have_generic = false;   // a table-driven kernel serves this decoder / code shape
    size_t lds_bytes = 0;
    double ims_thr = 1.4;  // MS_THR, MS_QBITS, MS_DBITS (decoders.h:46-48), see ldpc_hip_set_ims_params
    int ims_qbits = 6, ims_dbits = 8;
    // shape-unlimited tier (ldpc_global.hpp): message state in a workspace in global memory
    bool asp_cw2 = false;    // ASP_DEC on a code whose block columns all hold two circulants: upstream's own branch, on this tier only
    bool global_tier = false;
    char *d_glob_ws = nullptr;
    size_t glob_stride = 0;
    int glob_grid = 0;
    // table-driven M = 64 min-sum kernel (ldpc_ms_fast.hpp)
    bool fast_m64 = false;
    int variant = 2;  // LDPC_HIP_MS_VARIANT: 2 code-specialised (AOT / hiprtc) [default], 0 table kernel with LDS fp64 atomics,
                      // 1 table kernel read-add-write, -1 generic kernels only
    ldpc::FastTab fast_tab;
    // code-specialised kernel (ldpc_spec.hpp): one frame per workgroup of spec_threads threads
    const void *spec_aot = nullptr;
    const ldpc_jit::Kernel *spec_jit = nullptr;
    int spec_threads = 64;
    int spec_frames_per_block = 1;
    size_t spec_lds = 0;
    std::shared_ptr<ldpc_jit::Job> jit_job;   // LDPC_HIP_JIT=async: the instance is being compiled in the background; until it is
    std::string jit_name;                     // ready the table-driven / shape-unlimited tier runs (identical bits)
    bool global_is_fallback = false;
    // persistent launch of the one-wave-per-frame min-sum body (SpecArgs::queue)
    unsigned *d_queue = nullptr;
    int persist_grid = 0;     // resident waves of that kernel on this device (0: not determined yet)
    bool persist_jit = false; // ... determined for the hiprtc instance (a context may move from one to the other)
    std::string kernel_name;  // what this context launches (ldpc_hip_kernel_name)
    std::string generic_name; // the table-driven kernel of this decoder, if one serves this code shape
    const char *last_launch = "";  // ldpc_hip_last_launch
    // BP_DEC: upstream's input check sees the syndrome the previous frame left in DEC_STATE::syndr (decoders.cpp:1742-1762)
    bool bp_chain = true;                 // ldpc_hip_set_bp_chain
    std::vector<uint32_t> bp_carry;       // [R/32] syndrome left behind by the last frame decoded on this context
    uint32_t *d_bp_stale = nullptr, *d_bp_synd = nullptr;
    int32_t *d_bp_idx = nullptr;
    long long bp_frames = 0;
    // IMS_DEC: per-frame quantiser scale for the code-specialised kernel (ims_coef_kernel)
    double *d_ims_coef = nullptr;
    long long ims_coef_frames = 0;
    // device tables of the generic kernels
    int32_t *d_row_start = nullptr, *d_col_start = nullptr;
    uint32_t *d_edges = nullptr, *d_col_edges = nullptr, *d_col_slot = nullptr, *d_edge_row = nullptr;
    // workspace for ldpc_hip_simulate / decode_host
    double *w_llr = nullptr;
    uint32_t *w_hard = nullptr;
    int32_t *w_iters = nullptr;
    double *w_soft = nullptr;
    unsigned long long *w_counters = nullptr;
    long long w_frames = 0;
    bool w_has_soft = false;
    // simulation chain around the decoder (ldpc_hip_set_interleaver / ldpc_hip_set_codewords): default = upstream's shipped
    // wiring, all-zero codeword and permutation_type 0
    int perm_type = 0, perm_block = 128, perm_inter = 1;
    std::vector<int> hd_int;              // the base matrix as opened (the interleaver builder reads it)
    std::vector<uint8_t> codewords;       // [ncw][N] 0/1, decoder order (ldpc_hip_set_codewords)
    uint8_t *d_cw_bytes = nullptr;        // [ncw][N] the same table when it was produced on the device (ldpc_hip_set_random_codewords)
    int *d_hd_enc = nullptr;              // [rh][nh] base matrix for the device encoder
    int ncw = 0;
    int chain_mod = -1;                   // modulation_type the device-side chain tables below were built for (-1: stale)
    int chain_ntx = 0;
    uint8_t *d_tx = nullptr;              // [ncw][ntx] bits in channel order, zero padded to whole symbols
    uint32_t *d_cw_packed = nullptr;      // [ncw][hard_words]
    int32_t *d_scatter = nullptr;         // [N] decoder index of channel bit j
    // exact replay of upstream's noise on the device (ldpc_mt.hpp, ldpc_hip_mt_*)
    ldpc_mt::DeviceState mt;
    MtShardRound mt_round;                // a round of the shared-out generator whose exchange the caller does (ldpc_hip_mt_shard_*)
    // HIP-event timing of decode launches
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    double prof_ms = 0;
    long long prof_launches = 0;
};

namespace {

std::atomic<int> g_jit_mode{1};   // ldpc_hip_set_jit_mode: 0 never, 1 compile inside ldpc_hip_open, 2 compile in the background
thread_local int t_jit_mode = -1;  // ldpc_hip_set_jit_mode_thread: this thread's override (-1 none)

int jit_mode_effective() {
    if (const char *e = getenv("LDPC_HIP_JIT")) {   // the environment wins
        const std::string v(e);
        if (v == "async" || v == "2") return 2;
        if (v == "sync") return 1;
        return atoi(e) != 0 ? 1 : 0;
    }
    if (t_jit_mode >= 0) return t_jit_mode;
    return g_jit_mode.load();
}

// moves a context onto its code-specialised instance once the background compile has delivered it
void adopt_jit(ldpc_hip_ctx *c) {
    if (!c->jit_job) return;
    const int st = c->jit_job->state.load(std::memory_order_acquire);
    if (st == 0) return;
    if (st == 1) {
        c->spec_jit = c->jit_job->kernel;
        c->kernel_name = c->jit_name;
        if (c->global_is_fallback) c->global_tier = false;
    }
    c->jit_job.reset();
}

int set_device(const ldpc_hip_ctx *c) {
    HIP_TRY(hipSetDevice(c->device));
    return 0;
}

void free_workspace(ldpc_hip_ctx *c) {
    if (c->w_llr) (void)hipFree(c->w_llr);
    if (c->w_hard) (void)hipFree(c->w_hard);
    if (c->w_iters) (void)hipFree(c->w_iters);
    if (c->w_soft) (void)hipFree(c->w_soft);
    c->w_llr = nullptr; c->w_hard = nullptr; c->w_iters = nullptr; c->w_soft = nullptr;
    c->w_frames = 0; c->w_has_soft = false;
}

int ensure_workspace(ldpc_hip_ctx *c, long long B, bool need_soft) {
    if (B <= c->w_frames && (!need_soft || c->w_has_soft)) return 0;
    const long long nb = B > c->w_frames ? B : c->w_frames;
    free_workspace(c);
    HIP_TRY(hipMalloc(&c->w_llr, sizeof(double) * (size_t)nb * c->N));
    HIP_TRY(hipMalloc(&c->w_hard, sizeof(uint32_t) * (size_t)nb * c->hard_words));
    HIP_TRY(hipMalloc(&c->w_iters, sizeof(int32_t) * (size_t)nb));
    if (need_soft) HIP_TRY(hipMalloc(&c->w_soft, sizeof(double) * (size_t)nb * c->N));
    c->w_has_soft = need_soft;
    c->w_frames = nb;
    return 0;
}

int set_lds_limit(const void *kernel, size_t bytes) {
    if (bytes > 48 * 1024) HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

// Which code-specialised body serves this decoder / code shape, if any.  `required`: no generic kernel exists.
struct SpecPlan { const char *body = nullptr; int threads = 0; size_t lds = 0; bool required = false; int frames_per_block = 1; };

constexpr size_t kLdsBudget = 160 * 1024 - 64;   // the bodies' dynamic LDS image; the rest is the frame queue's ticket word

SpecPlan plan_spec(int decoder_id, const CodeTables &t) {
    SpecPlan p;
    const int M = t.M, N = t.nh * t.M, W = (M + 63) / 64;
    const size_t soft_lds = sizeof(double) * (size_t)N + 16;  // a-posteriori values + vote flag
    if (!t.all_cols_used || t.max_rw > 16 || t.rh > 64 || t.nh > 64) return p;
    switch (decoder_id) {
    case LDPC_HIP_MS_DEC:
        if (M == 64) { p.body = "ms_m64_body"; p.threads = 64; p.lds = sizeof(double) * (size_t)N; }
        else if (M <= 32) {   // several frames per wavefront
            p.body = "ms_small_body"; p.threads = 64; p.frames_per_block = 64 / M; p.lds = sizeof(double) * (size_t)N * (size_t)(64 / M);
        }
        else if (M > 64 && M <= 128 && !(getenv("LDPC_HIP_MS_CHUNK") && atoi(getenv("LDPC_HIP_MS_CHUNK")) == 0)) {
            p.body = "ms_chunk_body"; p.threads = 64; p.lds = sizeof(double) * (size_t)N;   // one wave, two 64-lane chunks, no barriers
        }
        else if (M >= 33 && M <= 512 && soft_lds <= kLdsBudget) { p.body = "ms_body"; p.threads = 64 * W; p.lds = soft_lds; }
        break;
    case LDPC_HIP_LMS_DEC:
        if (M <= 32) {   // several frames per wavefront
            p.body = "lms_small_body"; p.threads = 64; p.frames_per_block = 64 / M; p.lds = sizeof(double) * (size_t)N * (size_t)(64 / M);
        } else if (M >= 33 && M <= 512 && soft_lds <= kLdsBudget) { p.body = "lms_body"; p.threads = 64 * W; p.lds = soft_lds; }
        break;
    case LDPC_HIP_IMS_DEC: {   // int8 messages: MS_DBITS <= 8 and ialpha <= 16, checked per launch (the generic kernel takes the rest)
        const size_t lds = (((size_t)2 * N + 15) & ~(size_t)15) + (size_t)t.rh * 2 * LDPC_IMS_MSG_COPIES * M * 4 + 16;
        if (M <= 32 && t.max_rw <= 8) {   // several frames per wavefront
            const size_t Fr = 64 / M;
            p.body = "ims_small_body"; p.threads = 64; p.frames_per_block = (int)Fr;
            p.lds = ((Fr * 2 * N + 15) & ~(size_t)15) + Fr * (size_t)t.rh * 2 * LDPC_IMS_MSG_COPIES * M * 4 + 16;
        } else if (M >= 33 && M <= 512 && t.max_rw <= 8 && lds <= kLdsBudget) { p.body = "ims_body"; p.threads = 64 * W; p.lds = lds; }
        break;
    }
    case LDPC_HIP_SP_DEC: {
        const size_t lds = ldpc::sp_lds_bytes(t.ne, M, t.rh * M, N);
        if (M >= 33 && lds <= kLdsBudget) { p.body = "sp_body"; p.threads = 64 * ldpc_spec::kSpBodyWaves; p.lds = lds; }
        break;
    }
    case LDPC_HIP_BP_DEC: {
        p.required = true;  // code-specialised instances only
        const size_t lds = ((((sizeof(double) + 1) * ((size_t)t.ne * M + (size_t)t.rh * M) + (size_t)N) + 15) & ~(size_t)15) + 16;
        const size_t with_tables = lds + (size_t)ldpc_spec::kBpTabWords * 8;   // exp / log tables in LDS when they fit (bp_body: TAB_LDS)
        if (lds <= kLdsBudget) { p.body = "bp_body"; p.threads = 512; p.lds = with_tables <= kLdsBudget ? with_tables : lds; }
        break;
    }
    case LDPC_HIP_ASP_DEC: {
        p.required = true;  // code-specialised instances only; upstream's all-columns-of-weight-2 branch (decoders.cpp:2431-2480) runs on the shape-unlimited tier
        const size_t lds = sizeof(double) * (size_t)t.ne * M + (((size_t)N + 15) & ~(size_t)15) + 16;
        bool all_cw2 = true;
        for (int k = 0; k < t.nh; ++k) all_cw2 = all_cw2 && (t.col_start[k + 1] - t.col_start[k] == 2);
        if (t.min_rw >= 2 && !all_cw2 && lds <= kLdsBudget) { p.body = "asp_body"; p.threads = 512; p.lds = lds; }
        break;
    }
    case LDPC_HIP_TASP_DEC:
        p.required = true;  // per-edge state lives in VGPRs of the two lanes of a check: code-specialised instances only
        {   // 2 M threads per frame; LDS: a-posteriori probabilities + one spare slot per thread + flag words
            const int th = 64 * ((2 * M + 63) / 64);
            const size_t lds = sizeof(double) * ((size_t)N + (size_t)th) + 16;
            // registers of a lane: its half of every row's Z (2 each), the packed LDS addresses, ~85 temporaries (ldpc_jit.hpp picks one
            // or two waves per SIMD from the same estimate); rounds 1-2 (the whole row in one lane) stopped at 144 circulants
            int regs = 85;
            for (int j = 0; j < t.rh; ++j) { const int L = (t.row_start[j + 1] - t.row_start[j] + 1) / 2; regs += 2 * L + (L + 1) / 2; }
            if (M <= 256 && t.min_rw >= 2 && t.rh <= 64 && regs <= 480 && lds <= kLdsBudget) { p.body = "tasp_body"; p.threads = th; p.lds = lds; }
        }
        break;
    default: break;
    }
    return p;
}

// BP_DEC frame chain (upstream's uncleared syndrome array, decoders.cpp:1742-1762).  Frames are decoded as if one after the
// other on ONE upstream DEC_STATE: frame b's input check sees the syndrome frame b-1 left behind, frame 0 the one the previous
// call left in this context.  A frame's final syndrome is non-zero exactly when it failed, and the stale syndrome only matters
// to the input check, so: pass 1 with zero stale syndromes (frame 0: the carry); then re-decode only the frames that follow a
// failed frame, with stale[b] = synd[b-1] (the kernel leaves at once when nothing can change), and repeat for successors of
// frames whose outcome flipped.  Synchronises the stream (it reads the iteration counts back).
// launch(frames, io) enqueues one pass of the decoder's kernel (resident or global tier) over `frames` workgroup slots.
struct ChainIo { const uint32_t *stale = nullptr; uint32_t *synd_out = nullptr; const int *frame_idx = nullptr; int32_t *iters = nullptr; };
template <class Launch>
int run_bp_chain(ldpc_hip_ctx *c, long long B, int32_t *d_iters, hipStream_t stream, Launch launch) {
    const size_t sw = (size_t)c->rh * ((c->M + 63) / 64) * 2;   // u32 words per frame: one u64 per block row and 64-lane chunk
    if (B > c->bp_frames) {
        if (c->d_bp_stale) (void)hipFree(c->d_bp_stale);
        if (c->d_bp_synd) (void)hipFree(c->d_bp_synd);
        if (c->d_bp_idx) (void)hipFree(c->d_bp_idx);
        c->d_bp_stale = nullptr; c->d_bp_synd = nullptr; c->d_bp_idx = nullptr; c->bp_frames = 0;
        HIP_TRY(hipMalloc(&c->d_bp_stale, sizeof(uint32_t) * sw * (size_t)B));
        HIP_TRY(hipMalloc(&c->d_bp_synd, sizeof(uint32_t) * sw * (size_t)B));
        HIP_TRY(hipMalloc(&c->d_bp_idx, sizeof(int32_t) * (size_t)B));
        c->bp_frames = B;
    }
    int32_t *own_iters = nullptr;   // the chain needs the iteration counts even if the caller does not
    if (!d_iters) HIP_TRY(hipMalloc(&own_iters, sizeof(int32_t) * (size_t)B));
    std::unique_ptr<int32_t, void (*)(int32_t *)> own_guard(own_iters, [](int32_t *p) { if (p) (void)hipFree(p); });
    ChainIo io;
    io.iters = d_iters ? d_iters : own_iters;
    HIP_TRY(hipMemsetAsync(c->d_bp_stale, 0, sizeof(uint32_t) * sw * (size_t)B, stream));
    HIP_TRY(hipMemcpyAsync(c->d_bp_stale, c->bp_carry.data(), sizeof(uint32_t) * sw, hipMemcpyHostToDevice, stream));
    io.stale = c->d_bp_stale; io.synd_out = c->d_bp_synd; io.frame_idx = nullptr;
    if (int rc = launch(B, io)) return rc;
    std::vector<int32_t> it_old((size_t)B), it_new((size_t)B), todo;
    HIP_TRY(hipMemcpyAsync(it_old.data(), io.iters, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (long long b = 1; b < B; ++b) if (it_old[b - 1] < 0) todo.push_back((int32_t)b);
    while (!todo.empty()) {
        if (B > 1) HIP_TRY(hipMemcpyAsync(c->d_bp_stale + sw, c->d_bp_synd, sizeof(uint32_t) * sw * (size_t)(B - 1), hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipMemcpyAsync(c->d_bp_idx, todo.data(), sizeof(int32_t) * todo.size(), hipMemcpyHostToDevice, stream));
        io.frame_idx = c->d_bp_idx;
        if (int rc = launch((long long)todo.size(), io)) return rc;
        HIP_TRY(hipMemcpyAsync(it_new.data(), io.iters, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        std::vector<int32_t> next;
        for (int32_t b : todo)
            if ((it_old[b] < 0) != (it_new[b] < 0) && b + 1 < B) next.push_back(b + 1);  // its successor saw the wrong stale syndrome
        it_old = it_new;
        todo.swap(next);
    }
    HIP_TRY(hipMemcpy(c->bp_carry.data(), c->d_bp_synd + sw * (size_t)(B - 1), sizeof(uint32_t) * sw, hipMemcpyDeviceToHost));
    return 0;
}

}  // namespace

extern "C" {

int ldpc_hip_abi_version(void) { return LDPC_HIP_ABI_VERSION; }

const char *ldpc_hip_last_error(void) { return g_err.c_str(); }

int ldpc_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ldpc_hip_open(int decoder_id, int rh, int nh, int M, const int16_t *hd, int device, ldpc_hip_ctx **out) {
    if (out) *out = nullptr;
    if (!out || !hd || rh <= 0 || nh <= 0 || M <= 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_open: bad argument");
    if (decoder_id != LDPC_HIP_MS_DEC && decoder_id != LDPC_HIP_LMS_DEC && decoder_id != LDPC_HIP_SP_DEC && decoder_id != LDPC_HIP_IMS_DEC &&
        decoder_id != LDPC_HIP_TASP_DEC && decoder_id != LDPC_HIP_ASP_DEC && decoder_id != LDPC_HIP_BP_DEC)
        return fail(LDPC_HIP_EUNSUPPORTED, "ldpc_hip_open: decoder id %d is not built (built: BP=0, SP=1, ASP=2, MS=3, IMS=4, TASP=7, LMS=8)", decoder_id);
    if (M >= 65536 || nh >= 65536 || rh >= 65536) return fail(LDPC_HIP_EUNSUPPORTED, "M, rh and nh must be < 65536");
    if ((long long)nh * M >= (1LL << 28)) return fail(LDPC_HIP_EUNSUPPORTED, "code length nh * M = %lld: at most 2^28 - 1 is supported (32-bit indices)", (long long)nh * M);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(LDPC_HIP_EINVAL, "ldpc_hip_open: device %d of %d", device, ndev);

    const CodeTables t(rh, nh, M, hd);
    std::unique_ptr<ldpc_hip_ctx, void (*)(ldpc_hip_ctx *)> c(new ldpc_hip_ctx(), ldpc_hip_close);
    c->decoder_id = decoder_id; c->device = device;
    c->rh = rh; c->nh = nh; c->M = M; c->N = nh * M; c->R = rh * M; c->ne = t.ne;
    c->hard_words = (c->N + 31) / 32;
    c->hd_int.assign(hd, hd + (size_t)rh * nh);
    c->bp_carry.assign((size_t)rh * ((M + 63) / 64) * 2, 0u);
    const char *venv = getenv("LDPC_HIP_MS_VARIANT");
    c->variant = venv ? atoi(venv) : 2;

    // ---- generic (table-driven) kernel for this decoder, where one exists
    bool have_generic = false;
    if (decoder_id == LDPC_HIP_MS_DEC || decoder_id == LDPC_HIP_LMS_DEC || decoder_id == LDPC_HIP_IMS_DEC) {
        const bool needs_nh = decoder_id != LDPC_HIP_LMS_DEC;  // the flooding kernels keep the channel LLRs of kNHM block columns in VGPRs
        c->multiwave = M > 64;
        c->F = c->multiwave ? 1 : 64 / M;
        c->threads = c->multiwave ? ((M + 63) / 64) * 64 : 64;
        c->lds_bytes = sizeof(double) * (size_t)c->N * c->F + 16;
        have_generic = rh <= kRHM && t.max_rw <= kRWM && (!needs_nh || nh <= kNHM) && M <= 512 && c->lds_bytes <= 160 * 1024;
        const char *base = decoder_id == LDPC_HIP_MS_DEC ? "ms_flood_kernel" : decoder_id == LDPC_HIP_IMS_DEC ? "ims_flood_kernel" : "lms_layered_kernel";
        c->kernel_name = std::string(base) + (c->multiwave ? "<multiwave>" : "");
        if (have_generic && decoder_id == LDPC_HIP_MS_DEC && M == 64 && rh <= ldpc::kFastRows && nh <= ldpc::kFastCols && t.all_cols_used &&
            t.max_rw <= ldpc::kFastSlots && c->variant >= 0) {
            // table-driven M = 64 kernel: 64 dwords of packed descriptors that stay in SGPRs
            c->fast_m64 = true;
            c->kernel_name = c->variant == 1 ? "ms_flood_m64_kernel<rmw>" : "ms_flood_m64_kernel<atomic>";
            std::memset(&c->fast_tab, 0, sizeof c->fast_tab);
            std::vector<char> seen(nh, 0);
            for (int e = 0; e < t.ne; ++e) {
                const uint32_t j = t.edge_row[e], k = t.edges[e] >> 16, sh = t.edges[e] & 0xffffu;
                const uint32_t first = seen[k] ? 0u : 1u;  // rows ascend: the first hit is the column's first edge
                seen[k] = 1;
                const int slot = e - t.row_start[j];
                c->fast_tab.pk[j][slot >> 1] |= ldpc::fast_desc(first, k, sh) << ((slot & 1) * 16);
            }
            c->lds_bytes = sizeof(double) * 2048;
        }
    } else if (decoder_id == LDPC_HIP_SP_DEC) {
        c->multiwave = true; c->F = 1; c->threads = ldpc::kSpThreads;
        c->lds_bytes = ldpc::sp_lds_bytes(t.ne, M, c->R, c->N);
        c->kernel_name = "sp_flood_kernel";
        have_generic = c->N <= ldpc::kSpNVM * c->threads && c->lds_bytes <= 160 * 1024;
    }

    c->have_generic = have_generic;
    if (have_generic) c->generic_name = c->kernel_name;

    bool all_cw2 = true;   // decoder 2 has its own branch for codes whose block columns all hold two circulants (decoders.cpp:1027-1044,
                           // :2431-2480): the shape-unlimited tier runs it (asp_global_kernel), the resident asp_body is the general branch
    for (int k = 0; k < nh; ++k) all_cw2 = all_cw2 && (t.col_start[k + 1] - t.col_start[k] == 2);
    c->asp_cw2 = decoder_id == LDPC_HIP_ASP_DEC && all_cw2;
    const bool can_global = decoder_id == LDPC_HIP_BP_DEC || decoder_id == LDPC_HIP_MS_DEC || decoder_id == LDPC_HIP_LMS_DEC || decoder_id == LDPC_HIP_SP_DEC || decoder_id == LDPC_HIP_IMS_DEC ||
                            (decoder_id == LDPC_HIP_TASP_DEC && t.min_rw >= 2) || (decoder_id == LDPC_HIP_ASP_DEC && t.min_rw >= 2);

    // ---- code-specialised instance: ahead of time for the shipped example code, hiprtc for anything else
    const SpecPlan plan = plan_spec(decoder_id, t);
    std::string why_not = plan.body ? "" : "this code shape has no code-specialised kernel";
    if (plan.body && (c->variant >= 2 || plan.required)) {
        c->spec_threads = plan.threads; c->spec_lds = plan.lds; c->spec_frames_per_block = plan.frames_per_block;
        for (const AotInstance &inst : kAot)
            if (inst.decoder == decoder_id && inst.threads == plan.threads && (t.*inst.matches)()) {
                c->spec_aot = inst.fn;
                c->kernel_name = std::string(inst.name) + " (ahead of time)";
                break;
            }
        const int jit_mode = jit_mode_effective();
        const char *fg = getenv("LDPC_HIP_FORCE_GLOBAL");
        const bool forced_global = fg && atoi(fg) != 0;
        if (!c->spec_aot && jit_mode != 0 && !forced_global) {
            // background compile only where another HIP tier can serve the first launches
            const bool has_fallback = have_generic || can_global;
            if (jit_mode == 2 && has_fallback) {
                c->spec_jit = ldpc_jit::get(device, plan.body, t.rows(), nh, M, why_not, /*compile=*/false);   // process or disk cache
                if (!c->spec_jit) {
                    auto job = std::make_shared<ldpc_jit::Job>();
                    job->device = device; job->nh = nh; job->M = M; job->body = plan.body; job->rows = t.rows();
                    ldpc_jit::Worker::instance().submit(job);
                    c->jit_job = job;
                    c->jit_name = std::string(plan.body) + " instance (hiprtc)";
                    why_not = "its hiprtc instance is being compiled in the background";
                }
            } else {
                c->spec_jit = ldpc_jit::get(device, plan.body, t.rows(), nh, M, why_not);
            }
            if (c->spec_jit) c->kernel_name = std::string(plan.body) + " instance (hiprtc)";
        } else if (!c->spec_aot) {
            why_not = forced_global ? "LDPC_HIP_FORCE_GLOBAL" : "LDPC_HIP_JIT=0";
        }
    }
    const char *genv = getenv("LDPC_HIP_FORCE_GLOBAL");   // tests: run the shape-unlimited tier on shapes the resident kernels take
    if (can_global && ((genv && atoi(genv) != 0) || (!c->spec_aot && !c->spec_jit && !have_generic))) {
        c->global_tier = true;
        c->global_is_fallback = c->jit_job != nullptr;   // until the background instance arrives
        c->spec_aot = nullptr; c->spec_jit = nullptr;
        c->kernel_name = decoder_id == LDPC_HIP_MS_DEC ? "ms_global_kernel" : decoder_id == LDPC_HIP_LMS_DEC ? "lms_global_kernel" : decoder_id == LDPC_HIP_IMS_DEC ? "ims_global_kernel" : decoder_id == LDPC_HIP_ASP_DEC ? "asp_global_kernel" : decoder_id == LDPC_HIP_BP_DEC ? "bp_global_kernel" :
                         decoder_id == LDPC_HIP_SP_DEC ? "sp_global_kernel" : "tasp_global_kernel";
        c->glob_stride = ldpc::glob_ws_bytes(c->N, c->R, t.ne, M, (decoder_id == LDPC_HIP_MS_DEC || decoder_id == LDPC_HIP_IMS_DEC) ? 0 : decoder_id == LDPC_HIP_TASP_DEC ? 4 : decoder_id == LDPC_HIP_ASP_DEC ? 3 : 1);
    }
    if (decoder_id == LDPC_HIP_IMS_DEC && !c->global_tier)   // parameters beyond int8 may send a launch to the global tier later
        c->glob_stride = ldpc::glob_ws_bytes(c->N, c->R, t.ne, M, 0);
    if (!c->spec_aot && !c->spec_jit && !c->global_tier) {
        if (!have_generic)
            return fail(LDPC_HIP_EUNSUPPORTED, "decoder %d, code %dx%d lifting %d: not supported by the generic kernel (limits: %d block rows, "
                        "%d block columns, row weight %d, M <= 512, 160 KiB LDS), no code-specialised instance: %s; the shape-unlimited "
                        "tier serves every built decoder; decoders 2 and 7 need row weights >= 2",
                        decoder_id, rh, nh, M, kRHM, kNHM, kRWM, why_not.c_str());
        if (plan.body && c->variant >= 2 && !c->jit_job)
            fprintf(stderr, "[ldpc_hip] code-specialised kernel unavailable (%s); using %s\n", why_not.c_str(), c->kernel_name.c_str());
    }

    HIP_TRY(hipSetDevice(device));
    auto upload = [&](auto **dst, const auto &src) -> hipError_t {
        using T = typename std::remove_reference<decltype(src)>::type::value_type;
        hipError_t e = hipMalloc(dst, sizeof(T) * (src.size() + 1));
        if (e == hipSuccess && !src.empty()) e = hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice);
        return e;
    };
    HIP_TRY(upload(&c->d_row_start, t.row_start));
    HIP_TRY(upload(&c->d_col_start, t.col_start));
    HIP_TRY(upload(&c->d_edges, t.edges));
    HIP_TRY(upload(&c->d_col_edges, t.col_edges));
    HIP_TRY(upload(&c->d_col_slot, t.col_slot));
    HIP_TRY(upload(&c->d_edge_row, t.edge_row));
    HIP_TRY(hipMalloc(&c->w_counters, sizeof(unsigned long long) * 8));
    *out = c.release();
    return 0;
}

void ldpc_hip_close(ldpc_hip_ctx *c) {
    if (!c) return;
    if (c->jit_job) c->jit_job->cancelled.store(true);   // not compiled yet: drop it
    (void)hipSetDevice(c->device);
    free_workspace(c);
    if (c->d_row_start) (void)hipFree(c->d_row_start);
    if (c->d_col_start) (void)hipFree(c->d_col_start);
    if (c->d_edges) (void)hipFree(c->d_edges);
    if (c->d_col_edges) (void)hipFree(c->d_col_edges);
    if (c->d_col_slot) (void)hipFree(c->d_col_slot);
    if (c->d_edge_row) (void)hipFree(c->d_edge_row);
    if (c->w_counters) (void)hipFree(c->w_counters);
    if (c->d_queue) (void)hipFree(c->d_queue);
    if (c->d_ims_coef) (void)hipFree(c->d_ims_coef);
    if (c->d_bp_stale) (void)hipFree(c->d_bp_stale);
    if (c->d_bp_synd) (void)hipFree(c->d_bp_synd);
    if (c->d_bp_idx) (void)hipFree(c->d_bp_idx);
    if (c->d_glob_ws) (void)hipFree(c->d_glob_ws);
    if (c->d_tx) (void)hipFree(c->d_tx);
    if (c->d_cw_packed) (void)hipFree(c->d_cw_packed);
    if (c->d_scatter) (void)hipFree(c->d_scatter);
    if (c->d_cw_bytes) (void)hipFree(c->d_cw_bytes);
    if (c->d_hd_enc) (void)hipFree(c->d_hd_enc);
    ldpc_mt::release(c->mt);
    for (auto &ev : c->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    delete c;
}

int ldpc_hip_n(const ldpc_hip_ctx *c) { return c ? c->N : 0; }
int ldpc_hip_r(const ldpc_hip_ctx *c) { return c ? c->R : 0; }
int ldpc_hip_edges(const ldpc_hip_ctx *c) { return c ? c->ne : 0; }
int ldpc_hip_hard_words(const ldpc_hip_ctx *c) { return c ? c->hard_words : 0; }
const char *ldpc_hip_kernel_name(const ldpc_hip_ctx *c) {
    if (!c) return "";
    adopt_jit(const_cast<ldpc_hip_ctx *>(c));
    return c->kernel_name.c_str();
}

int ldpc_hip_set_jit_mode(int mode) {
    if (mode < 0 || mode > 2) return fail(LDPC_HIP_EINVAL, "ldpc_hip_set_jit_mode: 0 (never), 1 (inside ldpc_hip_open) or 2 (in the background)");
    return g_jit_mode.exchange(mode);
}
int ldpc_hip_set_jit_mode_thread(int mode) {
    if (mode < -1 || mode > 2) return fail(LDPC_HIP_EINVAL, "ldpc_hip_set_jit_mode_thread: -1 (no override), 0, 1 or 2");
    const int before = t_jit_mode;
    t_jit_mode = mode;
    return before;
}
const char *ldpc_hip_last_launch(const ldpc_hip_ctx *c) { return c ? c->last_launch : ""; }

int ldpc_hip_decode_dev(ldpc_hip_ctx *c, const double *d_llr, long long B, int maxiter, double alpha,
                        uint32_t *d_hard, int32_t *d_iters, double *d_soft, void *stream_) {
    if (!c || B < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_decode_dev: bad argument");
    if (B == 0) return 0;  // empty batch: nothing to do (an empty device tensor has a null pointer)
    if (!d_llr) return fail(LDPC_HIP_EINVAL, "ldpc_hip_decode_dev: null llr");
    if (B > 0x7fffffffLL) return fail(LDPC_HIP_EINVAL, "batch too large");
    // upstream's decoders with maxiter <= 0 return on a syndrome computed into stale state and leave decword untouched
    // (decoders.cpp:4602-4625,4766): there is nothing meaningful to reproduce, so it is rejected instead
    if (maxiter < 1) return fail(LDPC_HIP_EINVAL, "ldpc_hip_decode_dev: maxiter must be >= 1 (got %d)", maxiter);
    if (int rc = set_device(c)) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    adopt_jit(c);

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (c->prof) {
        HIP_TRY(hipEventCreate(&ev0));
        HIP_TRY(hipEventCreate(&ev1));
        HIP_TRY(hipEventRecord(ev0, stream));
    }
    bool use_spec = c->spec_aot || c->spec_jit;
    bool use_global = c->global_tier;
    const int ims_ialpha = (int)(alpha * (1 << 4));   // decoders.cpp:5458, MS_ALPHA_FPP = 4
    if (use_spec && c->decoder_id == LDPC_HIP_IMS_DEC && (c->ims_dbits > 8 || c->ims_qbits > 8 || ims_ialpha < 0 || ims_ialpha > 16)) {
        use_spec = false;   // values beyond int8 (messages, or quantised inputs): table-driven int32 kernel, or the global tier's
        if (!c->have_generic) use_global = true;
    }
    c->last_launch = use_global ? (c->global_tier ? c->kernel_name.c_str() : "ims_global_kernel") : use_spec ? c->kernel_name.c_str() : c->generic_name.c_str();
    if (use_global) {
        // one workgroup per frame, capped: the workspace is per workgroup and does not grow with the batch
        int grid = (int)(B < 1024 ? B : 1024);
        while (grid > 16 && (size_t)grid * c->glob_stride > ((size_t)8 << 30)) grid /= 2;
        if (grid > c->glob_grid) {
            if (c->d_glob_ws) (void)hipFree(c->d_glob_ws);
            c->d_glob_ws = nullptr; c->glob_grid = 0;
            HIP_TRY(hipMalloc(&c->d_glob_ws, (size_t)grid * c->glob_stride));
            c->glob_grid = grid;
        }
        ldpc::GlobArgs ga{};
        ga.d.llr = d_llr; ga.d.hard = d_hard; ga.d.iters = d_iters; ga.d.soft_out = d_soft;
        ga.d.row_start = c->d_row_start; ga.d.edges = c->d_edges; ga.d.col_start = c->d_col_start; ga.d.col_edges = c->d_col_edges;
        ga.d.col_slot = c->d_col_slot; ga.d.edge_row = c->d_edge_row;
        ga.d.B = B; ga.d.rh = c->rh; ga.d.nh = c->nh; ga.d.M = c->M; ga.d.N = c->N; ga.d.F = 1; ga.d.maxiter = maxiter;
        ga.d.hard_words = c->hard_words; ga.d.alpha = alpha;
        ga.ws = c->d_glob_ws; ga.ws_stride = c->glob_stride; ga.ne = c->ne; ga.asp_cw2 = c->asp_cw2 ? 1 : 0;
        if (c->decoder_id == LDPC_HIP_IMS_DEC) {
            if (B > c->ims_coef_frames) {
                if (c->d_ims_coef) (void)hipFree(c->d_ims_coef);
                c->d_ims_coef = nullptr; c->ims_coef_frames = 0;
                HIP_TRY(hipMalloc(&c->d_ims_coef, sizeof(double) * (size_t)B));
                c->ims_coef_frames = B;
            }
            ldpc::ImsCoefArgs ca{d_llr, c->d_ims_coef, B, c->N};
            hipLaunchKernelGGL(ldpc::ims_coef_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, stream, ca);
            HIP_TRY(hipGetLastError());
            ga.ims_coef = c->d_ims_coef;
            ga.d.ims_thr = c->ims_thr; ga.d.ims_qbits = c->ims_qbits; ga.d.ims_dbits = c->ims_dbits;
        }
        auto glaunch = [&](long long slots, const ChainIo &io) -> int {
            ga.slots = slots; ga.stale = io.stale; ga.synd_out = io.synd_out; ga.frame_idx = io.frame_idx;
            if (io.iters) ga.d.iters = io.iters;
            const dim3 gg((unsigned)(slots < grid ? slots : grid)), bb((unsigned)ldpc::glob_threads(c->N));
            switch (c->decoder_id) {
            case LDPC_HIP_MS_DEC: hipLaunchKernelGGL(ldpc::ms_global_kernel, gg, bb, 0, stream, ga); break;
            case LDPC_HIP_LMS_DEC: hipLaunchKernelGGL(ldpc::lms_global_kernel, gg, bb, 0, stream, ga); break;
            case LDPC_HIP_IMS_DEC: hipLaunchKernelGGL(ldpc::ims_global_kernel, gg, bb, 0, stream, ga); break;
            case LDPC_HIP_ASP_DEC: hipLaunchKernelGGL(ldpc::asp_global_kernel, gg, bb, 0, stream, ga); break;
            case LDPC_HIP_SP_DEC: hipLaunchKernelGGL(ldpc::sp_global_kernel, gg, bb, 0, stream, ga); break;
            case LDPC_HIP_BP_DEC: hipLaunchKernelGGL(ldpc::bp_global_kernel, gg, bb, 0, stream, ga); break;
            default: hipLaunchKernelGGL(ldpc::tasp_global_kernel, gg, bb, 0, stream, ga); break;
            }
            HIP_TRY(hipGetLastError());
            return 0;
        };
        if (c->decoder_id == LDPC_HIP_BP_DEC && c->bp_chain) {
            if (int rc = run_bp_chain(c, B, d_iters, stream, glaunch)) return rc;
        } else {
            ChainIo io;
            io.iters = d_iters;
            if (int rc = glaunch(B, io)) return rc;
        }
    } else if (use_spec) {
        // code-specialised kernel: one frame per workgroup
        ldpc_spec::SpecArgs sa{};
        if (c->decoder_id == LDPC_HIP_IMS_DEC) {
            if (B > c->ims_coef_frames) {
                if (c->d_ims_coef) (void)hipFree(c->d_ims_coef);
                c->d_ims_coef = nullptr; c->ims_coef_frames = 0;
                HIP_TRY(hipMalloc(&c->d_ims_coef, sizeof(double) * (size_t)B));
                c->ims_coef_frames = B;
            }
            ldpc::ImsCoefArgs ca{d_llr, c->d_ims_coef, B, c->N};
            hipLaunchKernelGGL(ldpc::ims_coef_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, stream, ca);
            HIP_TRY(hipGetLastError());
            sa.ims_coef = c->d_ims_coef; sa.ims_thr = c->ims_thr;
            sa.ims_max_quant = (1 << (c->ims_qbits - 1)) - 1; sa.ims_max_data = (1 << (c->ims_dbits - 1)) - 1; sa.ims_ialpha = ims_ialpha;
        }
        sa.llr = d_llr; sa.hard = d_hard; sa.iters = d_iters; sa.soft_out = d_soft; sa.maxiter = maxiter; sa.alpha = alpha;
        sa.nframes = B;
        // One frame per workgroup, min-sum and layered min-sum bodies (the frame loop costs the integer, SP and ASP bodies spilled
        // registers and the TDMP body 7 % of its speed; BP has its frame chain, the small-lifting bodies several frames per
        // workgroup): PERSISTENT workgroups, one per
        // resident slot of the chip, that pull frames from a queue (SpecArgs::queue) -- frames converge after different numbers of
        // iterations, and the next frame should start the moment a slot is free, without a workgroup launch in between
        static const bool persist_on = !(getenv("LDPC_HIP_PERSISTENT") && atoi(getenv("LDPC_HIP_PERSISTENT")) == 0);
        const bool persistent = persist_on && c->spec_frames_per_block == 1 &&
                                (c->decoder_id == LDPC_HIP_MS_DEC || c->decoder_id == LDPC_HIP_LMS_DEC);
        if (persistent) {
            if (!c->d_queue) HIP_TRY(hipMalloc(&c->d_queue, 64));
            if (c->persist_grid == 0 || c->persist_jit != (c->spec_aot == nullptr)) {   // resident workgroups: occupancy x CUs (8 x 256 for the flagship)
                int per_cu = 0, cus = 0;
                if (c->spec_aot) { if (int rc = set_lds_limit(c->spec_aot, c->spec_lds)) return rc; }
                hipError_t e = c->spec_aot ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, c->spec_aot, c->spec_threads, c->spec_lds)
                                           : hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, c->spec_jit->fn, c->spec_threads, c->spec_lds);
                if (e != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 1; }
                HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
                c->persist_grid = per_cu * (cus > 0 ? cus : 256);
                c->persist_jit = c->spec_aot == nullptr;
            }
        }
        auto launch = [&](long long frames) -> int {
            long long blocks = (frames + c->spec_frames_per_block - 1) / c->spec_frames_per_block;
            if (persistent && blocks > c->persist_grid) {
                HIP_TRY(hipMemsetAsync(c->d_queue, 0, sizeof(unsigned), stream));
                sa.queue = c->d_queue;
                blocks = c->persist_grid;
            }
            void *kargs[] = {&sa};
            if (c->spec_aot) {
                if (int rc = set_lds_limit(c->spec_aot, c->spec_lds)) return rc;
                HIP_TRY(hipLaunchKernel(c->spec_aot, dim3((unsigned)blocks), dim3((unsigned)c->spec_threads), kargs, c->spec_lds, stream));
            } else {
                HIP_TRY(hipModuleLaunchKernel(c->spec_jit->fn, (unsigned)blocks, 1, 1, (unsigned)c->spec_threads, 1, 1, (unsigned)c->spec_lds,
                                              stream, kargs, nullptr));
            }
            return 0;
        };
        if (c->decoder_id == LDPC_HIP_BP_DEC && c->bp_chain) {
            if (int rc = run_bp_chain(c, B, d_iters, stream, [&](long long frames, const ChainIo &io) -> int {
                    sa.stale = io.stale; sa.synd_out = io.synd_out; sa.frame_idx = io.frame_idx; sa.iters = io.iters;
                    return launch(frames);
                }))
                return rc;
        } else {
            if (int rc = launch(B)) return rc;
        }
    } else {
        ldpc::DecArgs a{};
        a.llr = d_llr; a.hard = d_hard; a.iters = d_iters; a.soft_out = d_soft;
        a.row_start = c->d_row_start; a.edges = c->d_edges;
        a.col_start = c->d_col_start; a.col_edges = c->d_col_edges; a.col_slot = c->d_col_slot; a.edge_row = c->d_edge_row;
        a.B = B; a.rh = c->rh; a.nh = c->nh; a.M = c->M; a.N = c->N; a.F = c->F;
        a.maxiter = maxiter; a.hard_words = c->hard_words; a.alpha = alpha;
        a.ims_thr = c->ims_thr; a.ims_qbits = c->ims_qbits; a.ims_dbits = c->ims_dbits;
        const dim3 grid((unsigned)((B + c->F - 1) / c->F)), block((unsigned)c->threads);
        void *kargs[] = {&a};
        const void *k = nullptr;
        switch (c->decoder_id) {
        case LDPC_HIP_MS_DEC:
            if (c->fast_m64) {
                void *fargs[] = {&a, &c->fast_tab};
                k = c->variant == 1 ? (const void *)ldpc::ms_flood_m64_kernel<false> : (const void *)ldpc::ms_flood_m64_kernel<true>;
                HIP_TRY(hipLaunchKernel(k, grid, block, fargs, c->lds_bytes, stream));
                k = nullptr;
            } else {
                k = c->multiwave ? (const void *)ldpc::ms_flood_kernel<kRHM, kNHM, true> : (const void *)ldpc::ms_flood_kernel<kRHM, kNHM, false>;
            }
            break;
        case LDPC_HIP_LMS_DEC:
            k = c->multiwave ? (const void *)ldpc::lms_layered_kernel<kRHM, kRWM, true> : (const void *)ldpc::lms_layered_kernel<kRHM, kRWM, false>;
            break;
        case LDPC_HIP_IMS_DEC:
            k = c->multiwave ? (const void *)ldpc::ims_flood_kernel<kRHM, kNHM, true> : (const void *)ldpc::ims_flood_kernel<kRHM, kNHM, false>;
            break;
        case LDPC_HIP_SP_DEC:
            k = (const void *)ldpc::sp_flood_kernel;
            break;
        default:
            return fail(LDPC_HIP_EUNSUPPORTED, "decoder %d has no generic kernel", c->decoder_id);
        }
        if (k) {
            if (int rc = set_lds_limit(k, c->lds_bytes)) return rc;
            HIP_TRY(hipLaunchKernel(k, grid, block, kargs, c->lds_bytes, stream));
        }
    }
    HIP_TRY(hipGetLastError());
    if (c->prof) {
        HIP_TRY(hipEventRecord(ev1, stream));
        c->events.emplace_back(ev0, ev1);
    }
    return 0;
}

int ldpc_hip_decode_host(ldpc_hip_ctx *c, double *llr, long long B, int maxiter, int decision, double alpha,
                         double *decword, int32_t *iters, int clobber_sp_input) {
    if (!c || !llr || B < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_decode_host: bad argument");
    if (B == 0) return 0;
    if (int rc = set_device(c)) return rc;
    const bool sp = c->decoder_id == LDPC_HIP_SP_DEC || c->decoder_id == LDPC_HIP_BP_DEC;  // soft[] is the working array upstream
    const bool tasp = c->decoder_id == LDPC_HIP_TASP_DEC || c->decoder_id == LDPC_HIP_ASP_DEC;  // probability-domain decoders
    if (c->decoder_id == LDPC_HIP_TASP_DEC) decision = 0;  // upstream ignores `decision` for this decoder: the result is always hard (decoders.cpp:2737-2738)
    const bool need_soft = decision != 0 || (sp && clobber_sp_input);
    if (int rc = ensure_workspace(c, B, need_soft)) return rc;
    const size_t nllr = (size_t)B * c->N;
    HIP_TRY(hipMemcpy(c->w_llr, llr, sizeof(double) * nllr, hipMemcpyHostToDevice));
    if (int rc = ldpc_hip_decode_dev(c, c->w_llr, B, maxiter, alpha, c->w_hard, c->w_iters,
                                     need_soft ? c->w_soft : nullptr, nullptr))
        return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (iters) HIP_TRY(hipMemcpy(iters, c->w_iters, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost));
    std::vector<double> soft;
    if (need_soft) {
        soft.resize(nllr);
        HIP_TRY(hipMemcpy(soft.data(), c->w_soft, sizeof(double) * nllr, hipMemcpyDeviceToHost));
    }
    if (decword) {
        if (decision) {
            std::memcpy(decword, soft.data(), sizeof(double) * nllr);
        } else {
            std::vector<uint32_t> hard((size_t)B * c->hard_words);
            HIP_TRY(hipMemcpy(hard.data(), c->w_hard, sizeof(uint32_t) * hard.size(), hipMemcpyDeviceToHost));
            for (long long b = 0; b < B; ++b)
                for (int v = 0; v < c->N; ++v)
                    decword[(size_t)b * c->N + v] = (double)((hard[(size_t)b * c->hard_words + (v >> 5)] >> (v & 31)) & 1u);
        }
    }
    if (sp && clobber_sp_input) std::memcpy(llr, soft.data(), sizeof(double) * nllr);  // decoders.cpp:1950,2124
    if (tasp && clobber_sp_input) {  // decoders.cpp:2611-2618: soft[] is left holding P(bit = 1) of the channel -- same exp() as the decoder's
        long long blocks = ((long long)nllr + 255) / 256;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(channel_prior_kernel, dim3((unsigned)blocks), dim3(256), 0, nullptr, c->w_llr, (long long)nllr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(llr, c->w_llr, sizeof(double) * nllr, hipMemcpyDeviceToHost));
    }
    return 0;
}

int ldpc_hip_set_bp_chain(ldpc_hip_ctx *c, int on, int reset_carry) {
    if (!c) return fail(LDPC_HIP_EINVAL, "ldpc_hip_set_bp_chain: null context");
    c->bp_chain = on != 0;
    if (reset_carry) std::fill(c->bp_carry.begin(), c->bp_carry.end(), 0u);
    return 0;
}

static int awgn_sigma(const ldpc_hip_ctx *c, double snr_db, int modulation_type, int punctured_blocks, double *sigma) {
    const int b = c->rh, cc = c->nh;
    if (punctured_blocks < 0 || punctured_blocks >= cc) return fail(LDPC_HIP_EINVAL, "punctured_blocks=%d", punctured_blocks);
    const double bitrate = (double)(cc - b) / (cc - punctured_blocks);  // bp_simulation.cpp:444
    if (modulation_type == 0) {
        *sigma = std::sqrt(std::pow(10, -snr_db / 10) / 2 / bitrate);    // :445
    } else if (modulation_type >= 1 && modulation_type <= 4) {   // bp_simulation.cpp:402-411: QAM4, QAM16, QAM64, QAM256
        const int QAM = 1 << (2 * modulation_type), halfmlog = modulation_type == 1 ? 1 : modulation_type;
        const double norm_factor = 2.0 * (QAM - 1.0) / 3.0;              // :447
        *sigma = std::sqrt(std::pow(10., -snr_db / 10.) / (2 * bitrate * halfmlog * 2) * norm_factor);  // :449
    } else {
        return fail(LDPC_HIP_EUNSUPPORTED, "modulation_type %d", modulation_type);
    }
    return 0;
}

// Device tables of the transmit side for one modulation: the interleaver's scatter map and the codewords in channel order.
static int prepare_chain(ldpc_hip_ctx *c, int modulation_type) {
    if (c->chain_mod == modulation_type) return 0;
    const int N = c->N, halfmlog = modulation_type <= 1 ? 1 : modulation_type;   // bp_simulation.cpp:402-411
    const int m = modulation_type <= 1 ? 2 : 2 * modulation_type;
    const int ntx = modulation_type <= 1 ? N : ((N + m - 1) / m) * m;           // extra_bits of :575, zero
    if (c->d_tx) (void)hipFree(c->d_tx);
    if (c->d_cw_packed) (void)hipFree(c->d_cw_packed);
    if (c->d_scatter) (void)hipFree(c->d_scatter);
    c->d_tx = nullptr; c->d_cw_packed = nullptr; c->d_scatter = nullptr; c->chain_mod = -1;
    std::vector<int32_t> direct, inverse;
    if (c->perm_type != 0) {
        ldpc::Interleaver il;
        std::string err;
        if (!ldpc::build_interleaver(c->rh, c->nh, c->M, halfmlog, c->perm_type, c->perm_block, c->perm_inter, c->hd_int.data(), il, err))
            return fail(LDPC_HIP_EUNSUPPORTED, "%s", err.c_str());
        direct.assign(il.direct.begin(), il.direct.end());
        inverse.assign(il.inverse.begin(), il.inverse.end());
        std::vector<int32_t> scatter((size_t)N, -1);
        for (int i = 0; i < N; ++i) {   // y[i] = buffer[inverse[i]] (bp_simulation.cpp:684)  <=>  y[scatter[j]] = buffer[j]
            if (inverse[(size_t)i] < 0 || inverse[(size_t)i] >= N || scatter[(size_t)inverse[(size_t)i]] != -1)
                return fail(LDPC_HIP_EUNSUPPORTED, "interleaver mode %d: the inverse map is not a permutation", c->perm_type);
            scatter[(size_t)inverse[(size_t)i]] = i;
        }
        HIP_TRY(hipMalloc(&c->d_scatter, sizeof(int32_t) * (size_t)N));
        HIP_TRY(hipMemcpy(c->d_scatter, scatter.data(), sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice));
    }
    if (c->ncw > 0 && c->d_cw_bytes) {   // the table lives on the device: transmit order and packed words by kernels
        int32_t *d_direct = nullptr;
        if (!direct.empty()) {
            HIP_TRY(hipMalloc(&d_direct, sizeof(int32_t) * (size_t)N));
            HIP_TRY(hipMemcpy(d_direct, direct.data(), sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice));
        }
        HIP_TRY(hipMalloc(&c->d_tx, (size_t)c->ncw * ntx));
        HIP_TRY(hipMalloc(&c->d_cw_packed, sizeof(uint32_t) * (size_t)c->ncw * c->hard_words));
        ldpc::CwOrderArgs oa{c->d_cw_bytes, c->d_tx, d_direct, c->ncw, N, ntx};
        ldpc::CwPackArgs pa{c->d_cw_bytes, c->d_cw_packed, c->ncw, N, c->hard_words};
        hipLaunchKernelGGL(ldpc::cw_channel_order_kernel, dim3(1024), dim3(256), 0, nullptr, oa);
        hipLaunchKernelGGL(ldpc::cw_pack_kernel, dim3(1024), dim3(256), 0, nullptr, pa);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        if (d_direct) (void)hipFree(d_direct);
    } else if (c->ncw > 0) {
        std::vector<uint8_t> tx((size_t)c->ncw * ntx, 0);
        std::vector<uint32_t> packed((size_t)c->ncw * c->hard_words, 0u);
        for (int w = 0; w < c->ncw; ++w) {
            const uint8_t *cw = c->codewords.data() + (size_t)w * N;
            for (int j = 0; j < N; ++j) tx[(size_t)w * ntx + j] = cw[direct.empty() ? j : direct[(size_t)j]] & 1;   // Permutation(.., 0, ..) :570
            for (int v = 0; v < N; ++v) packed[(size_t)w * c->hard_words + (v >> 5)] |= (uint32_t)(cw[v] & 1) << (v & 31);
        }
        HIP_TRY(hipMalloc(&c->d_tx, tx.size()));
        HIP_TRY(hipMemcpy(c->d_tx, tx.data(), tx.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&c->d_cw_packed, sizeof(uint32_t) * packed.size()));
        HIP_TRY(hipMemcpy(c->d_cw_packed, packed.data(), sizeof(uint32_t) * packed.size(), hipMemcpyHostToDevice));
    }
    c->chain_ntx = ntx;
    c->chain_mod = modulation_type;
    return 0;
}

int ldpc_hip_set_interleaver(ldpc_hip_ctx *c, int permutation_type, int permutation_block, int permutation_inter) {
    if (!c) return fail(LDPC_HIP_EINVAL, "ldpc_hip_set_interleaver: null context");
    if (permutation_type < 0 || permutation_type > 4) return fail(LDPC_HIP_EINVAL, "permutation_type %d (0..4)", permutation_type);
    if (int rc = set_device(c)) return rc;
    c->perm_type = permutation_type; c->perm_block = permutation_block; c->perm_inter = permutation_inter;
    c->chain_mod = -1;
    if (permutation_type != 0) {   // fail now, not at the first batch, when this mode does not accept the code shape
        const int keep = c->ncw;
        c->ncw = 0;
        const int rc = prepare_chain(c, 0);
        c->ncw = keep; c->chain_mod = -1;
        if (rc) { c->perm_type = 0; return rc; }
    }
    return 0;
}

int ldpc_hip_set_codewords(ldpc_hip_ctx *c, const uint8_t *codewords, int ncw) {
    if (!c || ncw < 0 || (ncw > 0 && !codewords)) return fail(LDPC_HIP_EINVAL, "ldpc_hip_set_codewords: bad argument");
    c->codewords.assign(codewords, codewords + (size_t)ncw * c->N);
    if (c->d_cw_bytes) { (void)hipSetDevice(c->device); (void)hipFree(c->d_cw_bytes); c->d_cw_bytes = nullptr; }
    c->ncw = ncw;
    c->chain_mod = -1;
    return 0;
}

// the device encoder's view of the code: one dual-diagonal block (encoder.h encode() with brk = {0, b}), validated by encoding
// one word on the host first
static int encoder_setup(ldpc_hip_ctx *c, ldpc::EncodeArgs &ea) {
    const int b = c->rh, cc = c->nh, M = c->M;
    const int *mx = c->hd_int.data();
    auto at = [&](int i, int j) { return mx[i * cc + j]; };
    std::vector<int> brk(1, 0);         // block boundaries of bp_simulation.cpp:143-156, as include/ldpc/encoder.h finds them
    for (int i = 1; i + 1 < b; ++i) {
        const bool bi = at(i, i) >= 0 && at(i + 1, i) >= 0 && at(i, i - 1) < 0;
        const bool uni = i > 1 && at(i, i) >= 0 && at(i + 1, i) < 0 && at(i, i - 1) < 0 && at(i - 1, i - 1) >= 0 && at(i - 1, i - 2) >= 0;
        if (bi || uni) brk.push_back(i);
    }
    brk.push_back(b);
    if ((int)brk.size() - 1 > ldpc::kEncMaxBlocks)
        return fail(LDPC_HIP_EUNSUPPORTED, "the device encoder takes up to %d dual-diagonal blocks; this base matrix has %d", ldpc::kEncMaxBlocks, (int)brk.size() - 1);
    if (cc <= b) return fail(LDPC_HIP_EUNSUPPORTED, "no information part to encode");
    const size_t lds = (size_t)c->N + (size_t)c->R + (size_t)M;
    if (lds > 150 * 1024) return fail(LDPC_HIP_EUNSUPPORTED, "code length %d is beyond the device encoder's LDS image", c->N);
    std::vector<uint8_t> probe((size_t)(cc - b) * M);
    for (size_t i = 0; i < probe.size(); ++i) probe[i] = (uint8_t)((i * 2654435761u >> 7) & 1u);
    ldpc::BitVec cw;
    const int rc = ldpc::encode(mx, b, cc, M, probe.data(), cw);
    if (rc != 0) return fail(LDPC_HIP_EUNSUPPORTED, "this base matrix is not encodable by the dual-diagonal encoder (code %d)", rc);
    ea.b = b; ea.c = cc; ea.M = M;
    ea.nblk = (int)brk.size() - 1;
    for (int q = 0; q < ea.nblk; ++q) {
        const int off = brk[(size_t)q], hi = brk[(size_t)q + 1], rb = hi - off;
        ea.off[q] = off; ea.hi[q] = hi;
        ea.single[q] = rb > 1 ? (at(off + 1, off) < 0) : 1;               // bp_simulation.cpp:33 on the block's own sub-matrix
        int p = 0;
        while (p < rb && at(off + p, hi - 1) <= 0) ++p;                     // :36-39
        ea.p[q] = p < rb ? p : 0;
    }
    if (!c->d_hd_enc) {
        HIP_TRY(hipMalloc(&c->d_hd_enc, sizeof(int) * (size_t)b * cc));
        HIP_TRY(hipMemcpy(c->d_hd_enc, mx, sizeof(int) * (size_t)b * cc, hipMemcpyHostToDevice));
    }
    ea.hd = c->d_hd_enc;
    return 0;
}

static int encode_launch(ldpc_hip_ctx *c, ldpc::EncodeArgs ea, const uint8_t *d_info, long long B, uint8_t *d_cw, hipStream_t st) {
    ea.info = d_info; ea.cw = d_cw; ea.B = B;
    const size_t lds = (size_t)c->N + (size_t)c->R + (size_t)c->M;
    if (int rc = set_lds_limit(reinterpret_cast<const void *>(ldpc::qc_encode_kernel), lds)) return rc;
    const long long grid = B < 256 * 8 ? B : 256 * 8;
    hipLaunchKernelGGL(ldpc::qc_encode_kernel, dim3((unsigned)grid), dim3(256), lds, st, ea);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ldpc_hip_encode_dev(ldpc_hip_ctx *c, const uint8_t *d_info_bits, long long B, uint8_t *d_codewords, void *stream_) {
    if (!c || B < 0 || (B > 0 && (!d_info_bits || !d_codewords))) return fail(LDPC_HIP_EINVAL, "ldpc_hip_encode_dev: bad argument");
    if (B == 0) return 0;
    if (int rc = set_device(c)) return rc;
    ldpc::EncodeArgs ea{};
    if (int rc = encoder_setup(c, ea)) return rc;
    return encode_launch(c, ea, d_info_bits, B, d_codewords, (hipStream_t)stream_);
}

int ldpc_hip_set_random_codewords(ldpc_hip_ctx *c, uint64_t seed, int ncw) {
    if (!c || ncw < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_set_random_codewords: bad argument");
    if (int rc = set_device(c)) return rc;
    if (ncw == 0) return ldpc_hip_set_codewords(c, nullptr, 0);
    ldpc::EncodeArgs ea{};
    if (int rc = encoder_setup(c, ea)) return rc;
    const int K = c->N - c->R;
    uint8_t *d_info = nullptr, *d_cw = nullptr;
    HIP_TRY(hipMalloc(&d_info, (size_t)ncw * K));
    if (hipMalloc(&d_cw, (size_t)ncw * c->N) != hipSuccess) { (void)hipFree(d_info); return fail(LDPC_HIP_ENOMEM, "ldpc_hip_set_random_codewords: out of device memory"); }
    ldpc::RandomInfoArgs ra{d_info, ncw, 0, K, seed};
    hipLaunchKernelGGL(ldpc::random_info_kernel, dim3(1024), dim3(256), 0, nullptr, ra);
    int rc = encode_launch(c, ea, d_info, ncw, d_cw, nullptr);
    if (rc == 0 && hipDeviceSynchronize() != hipSuccess) rc = fail(LDPC_HIP_EHIP, "ldpc_hip_set_random_codewords: %s", hipGetErrorString(hipGetLastError()));
    (void)hipFree(d_info);
    if (rc) { (void)hipFree(d_cw); return rc; }
    if (c->d_cw_bytes) (void)hipFree(c->d_cw_bytes);
    c->d_cw_bytes = d_cw;
    c->codewords.clear();
    c->ncw = ncw;
    c->chain_mod = -1;
    return 0;
}

int ldpc_hip_channel_llr_dev(ldpc_hip_ctx *c, double snr_db, int modulation_type, int punctured_blocks, double T, uint64_t seed,
                             long long first_frame, long long B, double *d_llr, void *stream_) {
    if (!c || !d_llr || B < 0 || first_frame < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_channel_llr_dev: bad argument");
    if (modulation_type < 0 || modulation_type > 4)
        return fail(LDPC_HIP_EUNSUPPORTED, "modulation_type %d (0 BPSK, 1 QAM4, 2 QAM16, 3 QAM64, 4 QAM256)", modulation_type);
    if (B == 0) return 0;
    if (int rc = set_device(c)) return rc;
    ldpc::ChannelArgs a{};
    if (int rc = awgn_sigma(c, snr_db, modulation_type, punctured_blocks, &a.sigma)) return rc;
    if (int rc = prepare_chain(c, modulation_type)) return rc;
    a.llr = d_llr; a.B = B; a.first_frame = first_frame; a.N = c->N; a.T = T; a.seed = seed;
    a.tx = c->ncw > 0 ? c->d_tx : nullptr; a.ncw = c->ncw > 0 ? c->ncw : 1; a.ntx = c->chain_ntx;
    a.scatter = c->d_scatter;
    a.punct_start = c->N - c->M * punctured_blocks;
    a.punct_val = (c->decoder_id == LDPC_HIP_SP_DEC || c->decoder_id == LDPC_HIP_TASP_DEC || c->decoder_id == LDPC_HIP_ASP_DEC) ? 0.0 : 0.5;  // :700 (sic), out_type :451-466
    const int m = modulation_type <= 1 ? 2 : 2 * modulation_type;
    const long long total = B * (long long)((c->N + m - 1) / m);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t stream = (hipStream_t)stream_;
    switch (modulation_type) {
    case 0: case 1: hipLaunchKernelGGL(ldpc::channel_llr_kernel<0>, grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL(ldpc::channel_llr_kernel<2>, grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL(ldpc::channel_llr_kernel<3>, grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL(ldpc::channel_llr_kernel<4>, grid, block, 0, stream, a); break;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int ldpc_hip_awgn_llr_dev(ldpc_hip_ctx *c, double snr_db, int modulation_type, int punctured_blocks, uint64_t seed,
                          long long first_frame, long long B, double *d_llr, void *stream_) {
    if (modulation_type != 0 && modulation_type != 1) return fail(LDPC_HIP_EUNSUPPORTED, "modulation_type %d (0 BPSK, 1 QAM4)", modulation_type);
    return ldpc_hip_channel_llr_dev(c, snr_db, modulation_type, punctured_blocks, 26.0, seed, first_frame, B, d_llr, stream_);
}

int ldpc_hip_awgn_qam_llr_dev(ldpc_hip_ctx *c, int modulation_type, double snr_db, double T, uint64_t seed, long long first_frame,
                              long long B, double *d_llr, void *stream_) {
    if (modulation_type < 2 || modulation_type > 4)
        return fail(LDPC_HIP_EUNSUPPORTED, "ldpc_hip_awgn_qam_llr_dev: modulation_type %d (2 QAM16, 3 QAM64, 4 QAM256)", modulation_type);
    return ldpc_hip_channel_llr_dev(c, snr_db, modulation_type, 0, T, seed, first_frame, B, d_llr, stream_);
}

int ldpc_hip_awgn_qam16_llr_dev(ldpc_hip_ctx *c, double snr_db, double T, uint64_t seed, long long first_frame,
                                long long B, double *d_llr, void *stream_) {
    return ldpc_hip_awgn_qam_llr_dev(c, 2, snr_db, T, seed, first_frame, B, d_llr, stream_);
}

int ldpc_hip_qam_modulate_dev(int Q, const uint8_t *d_bits, long long ns, double *d_x, int device, void *stream_) {
    if (!d_bits || !d_x || ns < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_qam_modulate_dev: bad argument");
    if (Q != 4 && Q != 16 && Q != 64 && Q != 256) return fail(LDPC_HIP_EUNSUPPORTED, "QAM-%d mapper not built (4, 16, 64, 256)", Q);
    if (ns == 0) return 0;
    HIP_TRY(hipSetDevice(device));
    ldpc::ModArgs a{d_bits, d_x, ns, Q == 4 ? 1 : Q == 16 ? 2 : Q == 64 ? 3 : 4};
    long long blocks = (ns + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(ldpc::qam_modulate_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ldpc_hip_qam_demod_dev(int Q, double T, double sigma, const double *d_x, long long ns, double *d_out,
                           int out_type, int device, void *stream_) {
    if (!d_x || !d_out || ns < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_qam_demod_dev: bad argument");
    if (Q != 4 && Q != 16 && Q != 64 && Q != 256) return fail(LDPC_HIP_EUNSUPPORTED, "QAM-%d demapper not built (4, 16, 64, 256)", Q);
    if (Q == 4 && out_type != 0) return fail(LDPC_HIP_EUNSUPPORTED, "QAM-4 probability output not built");
    if (ns == 0) return 0;
    HIP_TRY(hipSetDevice(device));
    ldpc::DemodArgs a{};
    a.x = d_x; a.out = d_out; a.ns = ns; a.Q = Q; a.out_type = out_type; a.T = T; a.sigma = sigma;
    long long blocks = (ns + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(ldpc::qam_demod_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ldpc_hip_encode_host(int rh, int nh, int M, const int16_t *hd, const uint8_t *info_bits, uint8_t *codeword) {
    if (!hd || !info_bits || !codeword || rh <= 0 || nh <= rh || M <= 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_encode_host: bad argument");
    std::vector<int> h((size_t)rh * nh);
    for (size_t i = 0; i < h.size(); ++i) h[i] = hd[i];
    ldpc::BitVec cw;
    const int rc = ldpc::encode(h.data(), rh, nh, M, info_bits, cw);
    if (rc != 0)
        return fail(LDPC_HIP_EUNSUPPORTED, "ldpc_hip_encode_host: this base matrix is not encodable by the dual-diagonal encoder (code %d: %s)", rc,
                    rc < 0 ? "no positive shift in the special parity column" : "result is not a codeword");
    std::memcpy(codeword, cw.data(), cw.size());
    return 0;
}

int ldpc_hip_interleaver_build(int b, int c, int M, int halfmlog, int mode, int block_size, int step_size, const int16_t *hd,
                               int32_t *direct, int32_t *inverse) {
    if (!hd || !direct || !inverse) return fail(LDPC_HIP_EINVAL, "ldpc_hip_interleaver_build: null argument");
    if (b <= 0 || c <= 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_interleaver_build: bad code shape");
    std::vector<int> h((size_t)b * c);
    for (size_t i = 0; i < h.size(); ++i) h[i] = hd[i];
    ldpc::Interleaver il;
    std::string err;
    if (!ldpc::build_interleaver(b, c, M, halfmlog, mode, block_size, step_size, h.data(), il, err)) return fail(LDPC_HIP_EUNSUPPORTED, "%s", err.c_str());
    std::memcpy(direct, il.direct.data(), sizeof(int32_t) * il.direct.size());
    std::memcpy(inverse, il.inverse.data(), sizeof(int32_t) * il.inverse.size());
    return 0;
}

int ldpc_hip_permute_dev(const double *d_in, double *d_out, long long B, int N, const int32_t *d_map, int device, void *stream_) {
    if (!d_in || !d_out || !d_map || B < 0 || N <= 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_permute_dev: bad argument");
    if (d_in == d_out) return fail(LDPC_HIP_EINVAL, "ldpc_hip_permute_dev: in-place permutation is not supported");
    if (B == 0) return 0;
    HIP_TRY(hipSetDevice(device));
    ldpc::PermuteArgs a{d_in, d_out, d_map, B, N};
    long long blocks = (B * (long long)N + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(ldpc::permute_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ldpc_hip_count_errors_cw_dev(ldpc_hip_ctx *c, const uint32_t *d_hard, const int32_t *d_iters, long long first_frame, long long B,
                                 int32_t *d_frame_info, unsigned long long *d_counters, void *stream_) {
    if (!c || !d_hard || !d_iters || !d_counters || B < 0 || first_frame < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_count_errors_dev: bad argument");
    if (B == 0) return 0;
    if (int rc = set_device(c)) return rc;
    if (c->ncw > 0 && c->chain_mod < 0) { if (int rc = prepare_chain(c, 0)) return rc; }   // the packed codewords do not depend on the modulation
    ldpc::CountArgs a{};
    a.hard = d_hard; a.iters = d_iters; a.frame_info = d_frame_info; a.counters = d_counters;
    a.B = B; a.hard_words = c->hard_words; a.R = c->R;
    a.cw = c->ncw > 0 ? c->d_cw_packed : nullptr; a.ncw = c->ncw > 0 ? c->ncw : 1; a.first_frame = first_frame;
    long long blocks = (B + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(ldpc::count_errors_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ldpc_hip_count_errors_dev(ldpc_hip_ctx *c, const uint32_t *d_hard, const int32_t *d_iters, long long B,
                              int32_t *d_frame_info, unsigned long long *d_counters, void *stream_) {
    if (c && c->ncw > 1) return fail(LDPC_HIP_EINVAL, "ldpc_hip_count_errors_dev: several codewords are set, the frame index is needed: use ldpc_hip_count_errors_cw_dev");
    return ldpc_hip_count_errors_cw_dev(c, d_hard, d_iters, 0, B, d_frame_info, d_counters, stream_);
}

}  // extern "C"

namespace {

// One Monte-Carlo pass over frames [first_frame, first_frame + B) on `stream`: channel -> decode -> count, in chunks of the
// workspace.  Counters accumulate into c->w_counters (caller zeroes); d_frame_info / d_iters_out (device, [B]) receive the
// ordered per-frame records when not null.  Asynchronous except for BP_DEC with the chain on.
int sim_enqueue(ldpc_hip_ctx *c, double snr_db, int modulation_type, int punctured_blocks, int maxiter, double alpha, uint64_t seed,
                long long first_frame, long long B, int32_t *d_frame_info, int32_t *d_iters_out, hipStream_t stream) {
    // frames per pass: at most 65536, and at most 8 GiB of LLRs (very long codes on the shape-unlimited tier)
    long long chunk_max = ((long long)8 << 30) / ((long long)sizeof(double) * c->N);
    chunk_max = chunk_max > (1 << 16) ? (1 << 16) : (chunk_max < 64 ? 64 : chunk_max);
    const long long chunk = B < chunk_max ? B : chunk_max;
    if (chunk > 0) { if (int rc = ensure_workspace(c, chunk, false)) return rc; }
    for (long long done = 0; done < B; done += chunk) {
        const long long nb = (B - done) < chunk ? (B - done) : chunk;
        int rc = ldpc_hip_channel_llr_dev(c, snr_db, modulation_type, punctured_blocks, 26.0, seed, first_frame + done, nb, c->w_llr, stream);
        if (rc) return rc;
        int32_t *it = d_iters_out ? d_iters_out + done : c->w_iters;
        if ((rc = ldpc_hip_decode_dev(c, c->w_llr, nb, maxiter, alpha, c->w_hard, it, nullptr, stream))) return rc;
        if ((rc = ldpc_hip_count_errors_cw_dev(c, c->w_hard, it, first_frame + done, nb, d_frame_info ? d_frame_info + done : nullptr,
                                               c->w_counters, stream)))
            return rc;
    }
    return 0;
}

}  // namespace

extern "C" {

int ldpc_hip_simulate(ldpc_hip_ctx *c, double snr_db, int modulation_type, int punctured_blocks, int maxiter,
                      double alpha, uint64_t seed, long long first_frame, long long B, unsigned long long counters[4],
                      unsigned long long *sum_abs_iters) {
    if (!c || !counters || B < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_simulate: bad argument");
    if (int rc = set_device(c)) return rc;
    HIP_TRY(hipMemsetAsync(c->w_counters, 0, sizeof(unsigned long long) * 8, nullptr));
    if (int rc = sim_enqueue(c, snr_db, modulation_type, punctured_blocks, maxiter, alpha, seed, first_frame, B, nullptr, nullptr, nullptr)) return rc;
    unsigned long long h[8];
    HIP_TRY(hipMemcpy(h, c->w_counters, sizeof h, hipMemcpyDeviceToHost));
    counters[0] = h[0]; counters[1] = h[1]; counters[2] = h[2]; counters[3] = h[3];
    if (sum_abs_iters) *sum_abs_iters = h[4];
    return 0;
}

int ldpc_hip_set_ims_params(ldpc_hip_ctx *c, double thr, int qbits, int dbits) {
    if (!c || !(thr > 0) || qbits < 2 || qbits > 15 || dbits < 2 || dbits > 15)
        return fail(LDPC_HIP_EINVAL, "ldpc_hip_set_ims_params: bad argument");
    c->ims_thr = thr; c->ims_qbits = qbits; c->ims_dbits = dbits;
    return 0;
}

int ldpc_hip_profile_enable(ldpc_hip_ctx *c, int enable) {
    if (!c) return fail(LDPC_HIP_EINVAL, "null ctx");
    c->prof = enable != 0;
    return 0;
}

int ldpc_hip_profile_read(ldpc_hip_ctx *c, double *total_ms, long long *launches, int reset) {
    if (!c) return fail(LDPC_HIP_EINVAL, "null ctx");
    if (int rc = set_device(c)) return rc;
    for (auto &ev : c->events) {
        HIP_TRY(hipEventSynchronize(ev.second));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
        c->prof_ms += ms;
        c->prof_launches += 1;
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    c->events.clear();
    if (total_ms) *total_ms = c->prof_ms;
    if (launches) *launches = c->prof_launches;
    if (reset) { c->prof_ms = 0; c->prof_launches = 0; }
    return 0;
}

}  // extern "C"

#include "ldpc_mt_api.hpp"
#include "ldpc_multi.hpp"
