// ldpc_hip.hip -- C-ABI (include/ldpc_hip.h) over the gfx950 kernels.  Host side: context, code tables,
// launch geometry, workspaces, HIP-event timing.  No CPU decode path exists in this library: every entry
// point either runs the HIP kernels or fails with a message.
#include "../../include/ldpc_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ldpc_frontend.hpp"
#include "ldpc_jit.hpp"
#include "ldpc_kernels.hpp"
#include "ldpc_ms_fast.hpp"
#include "ldpc_ms_spec.hpp"
#include "code_appendix_c_m64.hpp"
#include "ldpc_sumprod.hpp"

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

// ahead-of-time instance of the code-specialised kernel for the shipped example code
__global__ void __launch_bounds__(64, 2) ms_spec_appendix_c_m64_kernel(const ldpc_spec::SpecArgs a) {
    ldpc_spec::ms_m64_body<ldpc_spec::CodeAppendixCM64>(a);
}

__global__ void __launch_bounds__(128, 2) ms_spec_appendix_c_m126_kernel(const ldpc_spec::SpecArgs a) {
    ldpc_spec::ms_body<ldpc_spec::CodeAppendixCM126>(a);
}
__global__ void __launch_bounds__(512, 2) ms_spec_appendix_c_m512_kernel(const ldpc_spec::SpecArgs a) {
    ldpc_spec::ms_body<ldpc_spec::CodeAppendixCM512>(a);
}
__global__ void __launch_bounds__(64, 2) lms_spec_appendix_c_m64_kernel(const ldpc_spec::SpecArgs a) {
    ldpc_spec::lms_body<ldpc_spec::CodeAppendixCM64>(a);
}
__global__ void __launch_bounds__(512, 2) lms_spec_appendix_c_m512_kernel(const ldpc_spec::SpecArgs a) {
    ldpc_spec::lms_body<ldpc_spec::CodeAppendixCM512>(a);
}

__global__ void __launch_bounds__(512, 4) sp_spec_appendix_c_m64_kernel(const ldpc_spec::SpecArgs a) {
    ldpc_spec::sp_body<ldpc_spec::CodeAppendixCM64>(a);
}
__global__ void __launch_bounds__(512, 2) sp_spec_appendix_c_m64_occ2_kernel(const ldpc_spec::SpecArgs a) {
    ldpc_spec::sp_body<ldpc_spec::CodeAppendixCM64>(a);
}

__global__ void __launch_bounds__(64, 1) tasp_spec_appendix_c_m64_kernel(const ldpc_spec::SpecArgs a) {
    ldpc_spec::tasp_body<ldpc_spec::CodeAppendixCM64>(a);
}
__global__ void __launch_bounds__(128, 1) tasp_spec_appendix_c_m126_kernel(const ldpc_spec::SpecArgs a) {
    ldpc_spec::tasp_body<ldpc_spec::CodeAppendixCM126>(a);
}

template <class FC>
static bool same_code(int rh, int nh, int M, const std::vector<int32_t> &row_start, const std::vector<uint32_t> &edges) {
    bool same = rh == FC::RH && nh == FC::NH && M == FC::M;
    for (int j = 0; same && j < rh; ++j) {
        same = (row_start[j + 1] - row_start[j]) == FC::RW[j];
        for (int e = row_start[j]; same && e < row_start[j + 1]; ++e)
            same = (int)(edges[e] >> 16) == FC::COL[j][e - row_start[j]] && (int)(edges[e] & 0xffffu) == FC::SH[j][e - row_start[j]];
    }
    return same;
}

struct ldpc_hip_ctx {
    int decoder_id = 0, device = 0;
    int rh = 0, nh = 0, M = 0, N = 0, R = 0, ne = 0, hard_words = 0;
    int max_rw = 0, max_cw = 0;
    int F = 1;        // frames per workgroup (M <= 64: floor(64/M))
    int threads = 64; // workgroup size of the decode kernel
    bool multiwave = false;
    size_t lds_bytes = 0;
    double ims_thr = 1.4;      // MS_THR, MS_QBITS, MS_DBITS (decoders.h:46-48), see ldpc_hip_set_ims_params
    int ims_qbits = 6, ims_dbits = 8;
    bool fast_m64 = false;     // flagship path: min-sum, M == 64, table in the kernel-argument segment
    int fast_variant = 2;      // LDPC_HIP_MS_VARIANT: 2 = code-specialised (AOT/JIT) [default], 0 = table kernel with LDS
                               // fp64 atomics, 1 = table kernel read-add-write, -1 = generic kernel
    bool spec_aot = false;     // the opened matrix is the shipped example code: use the ahead-of-time instance
    int spec_threads = 64;     // workgroup size of the code-specialised kernel (one frame per workgroup)
    const ldpc_jit::Kernel *jit = nullptr;  // code-specialised instance compiled at open() for any other matrix
    std::string kernel_name;   // which decode kernel this context launches (ldpc_hip_kernel_name)
    ldpc::FastTab fast_tab;
    // device tables
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
    // HIP-event timing of decode launches
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    double prof_ms = 0;
    long long prof_launches = 0;
};

namespace {

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

template <typename K>
int set_lds_limit(K kernel, size_t bytes) {
    if (bytes > 48 * 1024) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)bytes));
    }
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
        decoder_id != LDPC_HIP_TASP_DEC)
        return fail(LDPC_HIP_EUNSUPPORTED, "ldpc_hip_open: decoder id %d is not built (built: SP=1, MS=3, IMS=4, TASP=7, LMS=8)", decoder_id);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(LDPC_HIP_EINVAL, "ldpc_hip_open: device %d of %d", device, ndev);

    ldpc_hip_ctx *c = new ldpc_hip_ctx();
    c->decoder_id = decoder_id; c->device = device;
    c->rh = rh; c->nh = nh; c->M = M; c->N = nh * M; c->R = rh * M;
    c->hard_words = (c->N + 31) / 32;

    std::vector<int32_t> row_start(rh + 1), col_start(nh + 1);
    std::vector<uint32_t> edges, col_edges, col_slot;
    std::vector<int> slot_of;  // per row-major edge: index inside its row
    for (int j = 0; j < rh; ++j) {
        row_start[j] = (int32_t)edges.size();
        for (int k = 0; k < nh; ++k) {
            int v = hd[j * nh + k];
            if (v == -1) continue;
            while (v < 0) v += M;   // rotate() reduces the shift like this (decoders.cpp:335-339)
            while (v >= M) v -= M;
            slot_of.push_back((int)edges.size() - row_start[j]);
            edges.push_back(((uint32_t)k << 16) | (uint32_t)v);
        }
        const int rw = (int)edges.size() - row_start[j];
        if (rw > c->max_rw) c->max_rw = rw;
    }
    row_start[rh] = (int32_t)edges.size();
    c->ne = (int)edges.size();
    for (int k = 0; k < nh; ++k) {
        col_start[k] = (int32_t)col_edges.size();
        for (int j = 0; j < rh; ++j)
            for (int e = row_start[j]; e < row_start[j + 1]; ++e)
                if ((int)(edges[e] >> 16) == k) {
                    col_edges.push_back(((uint32_t)j << 16) | (edges[e] & 0xffffu));
                    col_slot.push_back((uint32_t)e);  // global edge id (row-major)
                }
        const int cw = (int)col_edges.size() - col_start[k];
        if (cw > c->max_cw) c->max_cw = cw;
    }
    col_start[nh] = (int32_t)col_edges.size();

    // what the kernels were instantiated for
    if (M >= 65536 || nh >= 65536) { delete c; return fail(LDPC_HIP_EUNSUPPORTED, "M and nh must be < 65536"); }
    if (decoder_id == LDPC_HIP_TASP_DEC) {
        // TDMP sum-product exists as code-specialised instances only (per-edge state in VGPRs of the check lane)
        int min_rw = 1 << 30, regs = 0;
        for (int j = 0; j < rh; ++j) { const int rw = row_start[j + 1] - row_start[j]; min_rw = rw < min_rw ? rw : min_rw; regs += 2 * rw; }
        if (M > 256 || min_rw < 2 || regs > 288 || sizeof(double) * (size_t)c->N + 16 > 64 * 1024) {
            delete c;
            return fail(LDPC_HIP_EUNSUPPORTED, "TASP: needs M <= 256 (got %d), every block row of weight >= 2 (min %d), <= 144 circulants (got %d) and "
                        "N <= 8190 (got %d)", M, min_rw, regs / 2, nh * M);
        }
        c->F = 1; c->multiwave = M > 64;
        c->spec_threads = c->threads = ((M + 63) / 64) * 64;
        c->lds_bytes = sizeof(double) * (size_t)c->N + 16;
        if (same_code<ldpc_spec::CodeAppendixCM64>(rh, nh, M, row_start, edges) ||
            same_code<ldpc_spec::CodeAppendixCM126>(rh, nh, M, row_start, edges)) {
            c->spec_aot = true;
            c->kernel_name = M == 64 ? "tasp_spec_appendix_c_m64_kernel (ahead of time)" : "tasp_spec_appendix_c_m126_kernel (ahead of time)";
        } else {
            std::vector<std::vector<std::pair<int, int>>> rows(rh);
            for (int j = 0; j < rh; ++j)
                for (int e2 = row_start[j]; e2 < row_start[j + 1]; ++e2)
                    rows[j].emplace_back((int)(edges[e2] >> 16), (int)(edges[e2] & 0xffffu));
            std::string jerr;
            c->jit = ldpc_jit::get(device, "tasp_body", rows, nh, M, jerr);
            if (!c->jit) { delete c; return fail(LDPC_HIP_EUNSUPPORTED, "TASP needs a hiprtc instance for this base matrix: %s", jerr.c_str()); }
            c->kernel_name = "tasp_spec_jit (hiprtc)";
        }
    } else if (decoder_id == LDPC_HIP_MS_DEC || decoder_id == LDPC_HIP_LMS_DEC || decoder_id == LDPC_HIP_IMS_DEC) {
        if (rh > kRHM) { delete c; return fail(LDPC_HIP_EUNSUPPORTED, "rh=%d > %d block rows", rh, kRHM); }
        if (c->max_rw > kRWM) { delete c; return fail(LDPC_HIP_EUNSUPPORTED, "row weight %d > %d", c->max_rw, kRWM); }
        if ((decoder_id == LDPC_HIP_MS_DEC || decoder_id == LDPC_HIP_IMS_DEC) && nh > kNHM) { delete c; return fail(LDPC_HIP_EUNSUPPORTED, "nh=%d > %d block columns", nh, kNHM); }
        if (M > 512) { delete c; return fail(LDPC_HIP_EUNSUPPORTED, "M=%d > 512", M); }
        c->multiwave = M > 64;
        c->F = c->multiwave ? 1 : 64 / M;
        c->threads = c->multiwave ? ((M + 63) / 64) * 64 : 64;
        c->lds_bytes = sizeof(double) * (size_t)c->N * c->F + 16;
        bool all_cols_used = true;
        for (int k = 0; k < nh; ++k) all_cols_used = all_cols_used && (col_start[k + 1] > col_start[k]);
        const char *venv = getenv("LDPC_HIP_MS_VARIANT");
        c->fast_variant = venv ? atoi(venv) : 2;
        c->kernel_name = decoder_id == LDPC_HIP_MS_DEC ? (c->multiwave ? "ms_flood_kernel<multiwave>" : "ms_flood_kernel")
                       : decoder_id == LDPC_HIP_IMS_DEC ? (c->multiwave ? "ims_flood_kernel<multiwave>" : "ims_flood_kernel")
                                                        : (c->multiwave ? "lms_layered_kernel<multiwave>" : "lms_layered_kernel");
        const bool m64 = decoder_id == LDPC_HIP_MS_DEC && M == 64 && rh <= ldpc::kFastRows && nh <= ldpc::kFastCols && all_cols_used;
        if (m64 && c->max_rw <= ldpc::kFastSlots && c->fast_variant >= 0) {
            // table-driven M = 64 kernel (always available)
            c->fast_m64 = true;
            c->kernel_name = c->fast_variant == 1 ? "ms_flood_m64_kernel<rmw>" : "ms_flood_m64_kernel<atomic>";
            std::memset(&c->fast_tab, 0, sizeof c->fast_tab);
            std::vector<char> seen(nh, 0);
            for (int j = 0; j < rh; ++j)
                for (int e = row_start[j]; e < row_start[j + 1]; ++e) {
                    const uint32_t k = edges[e] >> 16, sh = edges[e] & 0xffffu;
                    const uint32_t first = seen[k] ? 0u : 1u;  // rows ascend: the first hit is the column's first edge
                    seen[k] = 1;
                    const int slot = e - row_start[j];
                    c->fast_tab.pk[j][slot >> 1] |= ldpc::fast_desc(first, k, sh) << ((slot & 1) * 16);
                }
            c->lds_bytes = sizeof(double) * 2048;
        }
        if (m64 && c->max_rw <= 16 && c->fast_variant == 2) {
            // code-specialised kernel: ahead-of-time instance for the shipped example code, hiprtc for anything else
            const bool same = same_code<ldpc_spec::CodeAppendixCM64>(rh, nh, M, row_start, edges);
            const char *jenv = getenv("LDPC_HIP_JIT");
            if (same) {
                c->spec_aot = true;
                c->kernel_name = "ms_spec_appendix_c_m64_kernel (ahead of time)";
                c->lds_bytes = sizeof(double) * (size_t)c->N;
            } else if (!jenv || atoi(jenv) != 0) {
                std::vector<std::vector<std::pair<int, int>>> rows(rh);
                for (int j = 0; j < rh; ++j)
                    for (int e = row_start[j]; e < row_start[j + 1]; ++e)
                        rows[j].emplace_back((int)(edges[e] >> 16), (int)(edges[e] & 0xffffu));
                std::string jerr;
                c->jit = ldpc_jit::get(device, "ms_m64_body", rows, nh, M, jerr);
                if (c->jit) {
                    c->kernel_name = "ms_spec_jit (hiprtc)";
                    c->lds_bytes = sizeof(double) * (size_t)c->N;
                } else {
                    fprintf(stderr, "[ldpc_hip] code-specialised kernel unavailable (%s); using %s\n", jerr.c_str(),
                            c->kernel_name.c_str());
                }
            }
        }
        if (decoder_id == LDPC_HIP_MS_DEC && M != 64 && M >= 48 && all_cols_used && c->fast_variant == 2 && nh <= 64 && c->max_rw <= 16 &&
            sizeof(double) * (size_t)c->N + 16 <= 160 * 1024) {
            // code-specialised flooding min-sum for liftings other than 64: one frame per workgroup of ceil(M/64) waves
            const char *jenv = getenv("LDPC_HIP_JIT");
            c->spec_threads = ((M + 63) / 64) * 64;
            if (same_code<ldpc_spec::CodeAppendixCM126>(rh, nh, M, row_start, edges) ||
                same_code<ldpc_spec::CodeAppendixCM512>(rh, nh, M, row_start, edges)) {
                c->spec_aot = true;
                c->kernel_name = M == 126 ? "ms_spec_appendix_c_m126_kernel (ahead of time)" : "ms_spec_appendix_c_m512_kernel (ahead of time)";
            } else if (!jenv || atoi(jenv) != 0) {
                std::vector<std::vector<std::pair<int, int>>> rows(rh);
                for (int j = 0; j < rh; ++j)
                    for (int e = row_start[j]; e < row_start[j + 1]; ++e)
                        rows[j].emplace_back((int)(edges[e] >> 16), (int)(edges[e] & 0xffffu));
                std::string jerr;
                c->jit = ldpc_jit::get(device, "ms_body", rows, nh, M, jerr);
                if (c->jit) c->kernel_name = "ms_spec_jit (hiprtc, multi-wave)";
                else fprintf(stderr, "[ldpc_hip] code-specialised kernel unavailable (%s); using %s\n", jerr.c_str(), c->kernel_name.c_str());
            }
            if (c->spec_aot || c->jit) c->lds_bytes = sizeof(double) * (size_t)c->N + 16;
        }
        if (decoder_id == LDPC_HIP_LMS_DEC && M >= 48 && all_cols_used && c->fast_variant == 2 &&
            sizeof(double) * (size_t)c->N + 16 <= 160 * 1024) {
            // code-specialised layered min-sum: one frame per workgroup of ceil(M/64) waves
            const char *jenv = getenv("LDPC_HIP_JIT");
            c->spec_threads = ((M + 63) / 64) * 64;
            if (same_code<ldpc_spec::CodeAppendixCM64>(rh, nh, M, row_start, edges) ||
                same_code<ldpc_spec::CodeAppendixCM512>(rh, nh, M, row_start, edges)) {
                c->spec_aot = true;
                c->kernel_name = M == 64 ? "lms_spec_appendix_c_m64_kernel (ahead of time)" : "lms_spec_appendix_c_m512_kernel (ahead of time)";
            } else if (!jenv || atoi(jenv) != 0) {
                std::vector<std::vector<std::pair<int, int>>> rows(rh);
                for (int j = 0; j < rh; ++j)
                    for (int e = row_start[j]; e < row_start[j + 1]; ++e)
                        rows[j].emplace_back((int)(edges[e] >> 16), (int)(edges[e] & 0xffffu));
                std::string jerr;
                c->jit = ldpc_jit::get(device, "lms_body", rows, nh, M, jerr);
                if (c->jit) c->kernel_name = "lms_spec_jit (hiprtc)";
                else fprintf(stderr, "[ldpc_hip] code-specialised kernel unavailable (%s); using %s\n", jerr.c_str(), c->kernel_name.c_str());
            }
            if (c->spec_aot || c->jit) c->lds_bytes = sizeof(double) * (size_t)c->N + 16;
        }
    } else {
        c->multiwave = true;
        c->F = 1;
        c->threads = ldpc::kSpThreads;
        c->lds_bytes = ldpc::sp_lds_bytes(c->ne, M, c->R, c->N);
        c->kernel_name = "sp_flood_kernel";
        if (c->N > ldpc::kSpNVM * c->threads) { delete c; return fail(LDPC_HIP_EUNSUPPORTED, "sum-product: N=%d > %d", c->N, ldpc::kSpNVM * c->threads); }
        {   // code-specialised sum-product: liftings that are a multiple of 64 and fit the LDS
            const char *venv = getenv("LDPC_HIP_MS_VARIANT");
            c->fast_variant = venv ? atoi(venv) : 2;
            const char *jenv = getenv("LDPC_HIP_JIT");
            bool all_cols_used = true;
            for (int k = 0; k < nh; ++k) all_cols_used = all_cols_used && (col_start[k + 1] > col_start[k]);
            if (c->fast_variant >= 2 && M % 64 == 0 && all_cols_used && rh <= 64 && c->lds_bytes <= 160 * 1024) {
                c->spec_threads = 512;
                if (same_code<ldpc_spec::CodeAppendixCM64>(rh, nh, M, row_start, edges)) {
                    c->spec_aot = true;
                    // two frames per CU (<= 128 VGPRs, a few spills) beats one frame with 243 VGPRs: 4.18 vs 3.70 M frames/s at 2 dB
                    c->kernel_name = c->fast_variant == 3 ? "sp_spec_appendix_c_m64_occ2_kernel (ahead of time)" : "sp_spec_appendix_c_m64_kernel (ahead of time)";
                } else if (!jenv || atoi(jenv) != 0) {
                    std::vector<std::vector<std::pair<int, int>>> rows(rh);
                    for (int j = 0; j < rh; ++j)
                        for (int e2 = row_start[j]; e2 < row_start[j + 1]; ++e2)
                            rows[j].emplace_back((int)(edges[e2] >> 16), (int)(edges[e2] & 0xffffu));
                    std::string jerr;
                    c->jit = ldpc_jit::get(device, "sp_body", rows, nh, M, jerr);
                    if (c->jit) c->kernel_name = "sp_spec_jit (hiprtc)";
                    else fprintf(stderr, "[ldpc_hip] code-specialised kernel unavailable (%s); using %s\n", jerr.c_str(), c->kernel_name.c_str());
                }
            }
        }
    }
    if (c->lds_bytes > 160 * 1024) {
        const size_t need = c->lds_bytes;
        delete c;
        return fail(LDPC_HIP_EUNSUPPORTED, "code needs %zu B of LDS per workgroup (> 160 KiB)", need);
    }

    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc(&c->d_row_start, sizeof(int32_t) * (rh + 1));
    if (e == hipSuccess) e = hipMalloc(&c->d_col_start, sizeof(int32_t) * (nh + 1));
    if (e == hipSuccess) e = hipMalloc(&c->d_edges, sizeof(uint32_t) * (c->ne + 1));
    if (e == hipSuccess) e = hipMalloc(&c->d_col_edges, sizeof(uint32_t) * (c->ne + 1));
    if (e == hipSuccess) e = hipMalloc(&c->d_col_slot, sizeof(uint32_t) * (c->ne + 1));
    if (e == hipSuccess) e = hipMalloc(&c->d_edge_row, sizeof(uint32_t) * (c->ne + 1));
    std::vector<uint32_t> edge_row(c->ne + 1, 0);
    for (int j = 0; j < rh; ++j) for (int q = row_start[j]; q < row_start[j + 1]; ++q) edge_row[q] = (uint32_t)j;
    if (e == hipSuccess && c->ne) e = hipMemcpy(c->d_edge_row, edge_row.data(), sizeof(uint32_t) * c->ne, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&c->w_counters, sizeof(unsigned long long) * 8);
    if (e == hipSuccess) e = hipMemcpy(c->d_row_start, row_start.data(), sizeof(int32_t) * (rh + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->d_col_start, col_start.data(), sizeof(int32_t) * (nh + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess && c->ne) e = hipMemcpy(c->d_edges, edges.data(), sizeof(uint32_t) * c->ne, hipMemcpyHostToDevice);
    if (e == hipSuccess && c->ne) e = hipMemcpy(c->d_col_edges, col_edges.data(), sizeof(uint32_t) * c->ne, hipMemcpyHostToDevice);
    if (e == hipSuccess && c->ne) e = hipMemcpy(c->d_col_slot, col_slot.data(), sizeof(uint32_t) * c->ne, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        ldpc_hip_close(c);
        return fail(LDPC_HIP_EHIP, "ldpc_hip_open: %s", hipGetErrorString(e));
    }
    *out = c;
    return 0;
}

void ldpc_hip_close(ldpc_hip_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    free_workspace(c);
    if (c->d_row_start) (void)hipFree(c->d_row_start);
    if (c->d_col_start) (void)hipFree(c->d_col_start);
    if (c->d_edges) (void)hipFree(c->d_edges);
    if (c->d_col_edges) (void)hipFree(c->d_col_edges);
    if (c->d_col_slot) (void)hipFree(c->d_col_slot);
    if (c->d_edge_row) (void)hipFree(c->d_edge_row);
    if (c->w_counters) (void)hipFree(c->w_counters);
    for (auto &ev : c->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    delete c;
}

int ldpc_hip_n(const ldpc_hip_ctx *c) { return c ? c->N : 0; }
int ldpc_hip_r(const ldpc_hip_ctx *c) { return c ? c->R : 0; }
int ldpc_hip_edges(const ldpc_hip_ctx *c) { return c ? c->ne : 0; }
int ldpc_hip_hard_words(const ldpc_hip_ctx *c) { return c ? c->hard_words : 0; }
const char *ldpc_hip_kernel_name(const ldpc_hip_ctx *c) { return c ? c->kernel_name.c_str() : ""; }

int ldpc_hip_decode_dev(ldpc_hip_ctx *c, const double *d_llr, long long B, int maxiter, double alpha,
                        uint32_t *d_hard, int32_t *d_iters, double *d_soft, void *stream_) {
    if (!c || B < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_decode_dev: bad argument");
    if (B == 0) return 0;  // empty batch: nothing to do (an empty device tensor has a null pointer)
    if (!d_llr) return fail(LDPC_HIP_EINVAL, "ldpc_hip_decode_dev: null llr");
    if (int rc = set_device(c)) return rc;
    hipStream_t stream = (hipStream_t)stream_;

    ldpc::DecArgs a{};
    a.llr = d_llr; a.hard = d_hard; a.iters = d_iters; a.soft_out = d_soft;
    a.row_start = c->d_row_start; a.edges = c->d_edges;
    a.col_start = c->d_col_start; a.col_edges = c->d_col_edges; a.col_slot = c->d_col_slot; a.edge_row = c->d_edge_row;
    a.B = B; a.rh = c->rh; a.nh = c->nh; a.M = c->M; a.N = c->N; a.F = c->F;
    a.maxiter = maxiter; a.hard_words = c->hard_words; a.alpha = alpha;
    a.ims_thr = c->ims_thr; a.ims_qbits = c->ims_qbits; a.ims_dbits = c->ims_dbits;

    const long long blocks = (B + c->F - 1) / c->F;
    if (blocks > 0x7fffffffLL) return fail(LDPC_HIP_EINVAL, "batch too large");
    const dim3 grid((unsigned)blocks), block((unsigned)c->threads);

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (c->prof) {
        HIP_TRY(hipEventCreate(&ev0));
        HIP_TRY(hipEventCreate(&ev1));
        HIP_TRY(hipEventRecord(ev0, stream));
    }
    switch (c->decoder_id) {
    case LDPC_HIP_MS_DEC:
        if (c->spec_aot || c->jit) {
            ldpc_spec::SpecArgs sa{};
            sa.llr = d_llr; sa.hard = d_hard; sa.iters = d_iters; sa.soft_out = d_soft; sa.maxiter = maxiter; sa.alpha = alpha;
            if (c->spec_aot && c->M == 64) {
                hipLaunchKernelGGL(ms_spec_appendix_c_m64_kernel, dim3((unsigned)B), dim3(64), c->lds_bytes, stream, sa);
            } else if (c->spec_aot && c->M == 126) {
                hipLaunchKernelGGL(ms_spec_appendix_c_m126_kernel, dim3((unsigned)B), dim3(128), c->lds_bytes, stream, sa);
            } else if (c->spec_aot) {
                if (int rc = set_lds_limit(ms_spec_appendix_c_m512_kernel, c->lds_bytes)) return rc;
                hipLaunchKernelGGL(ms_spec_appendix_c_m512_kernel, dim3((unsigned)B), dim3(512), c->lds_bytes, stream, sa);
            } else {
                void *kargs[] = {&sa};
                HIP_TRY(hipModuleLaunchKernel(c->jit->fn, (unsigned)B, 1, 1, (unsigned)c->spec_threads, 1, 1, (unsigned)c->lds_bytes, stream,
                                              kargs, nullptr));
            }
        } else if (c->fast_m64) {
            if (c->fast_variant == 1) hipLaunchKernelGGL(ldpc::ms_flood_m64_kernel<false>, grid, block, c->lds_bytes, stream, a, c->fast_tab);
            else hipLaunchKernelGGL(ldpc::ms_flood_m64_kernel<true>, grid, block, c->lds_bytes, stream, a, c->fast_tab);
        } else if (c->multiwave) {
            auto k = ldpc::ms_flood_kernel<kRHM, kNHM, true>;
            if (int rc = set_lds_limit(k, c->lds_bytes)) return rc;
            hipLaunchKernelGGL(k, grid, block, c->lds_bytes, stream, a);
        } else {
            auto k = ldpc::ms_flood_kernel<kRHM, kNHM, false>;
            if (int rc = set_lds_limit(k, c->lds_bytes)) return rc;
            hipLaunchKernelGGL(k, grid, block, c->lds_bytes, stream, a);
        }
        break;
    case LDPC_HIP_LMS_DEC:
        if (c->spec_aot || c->jit) {
            ldpc_spec::SpecArgs sa{};
            sa.llr = d_llr; sa.hard = d_hard; sa.iters = d_iters; sa.soft_out = d_soft; sa.maxiter = maxiter; sa.alpha = alpha;
            if (B > 0x7fffffffLL) return fail(LDPC_HIP_EINVAL, "batch too large");
            if (c->spec_aot && c->M == 64) {
                hipLaunchKernelGGL(lms_spec_appendix_c_m64_kernel, dim3((unsigned)B), dim3(64), c->lds_bytes, stream, sa);
            } else if (c->spec_aot) {
                if (int rc = set_lds_limit(lms_spec_appendix_c_m512_kernel, c->lds_bytes)) return rc;
                hipLaunchKernelGGL(lms_spec_appendix_c_m512_kernel, dim3((unsigned)B), dim3(512), c->lds_bytes, stream, sa);
            } else {
                void *kargs[] = {&sa};
                HIP_TRY(hipModuleLaunchKernel(c->jit->fn, (unsigned)B, 1, 1, (unsigned)c->spec_threads, 1, 1, (unsigned)c->lds_bytes, stream,
                                              kargs, nullptr));
            }
        } else if (c->multiwave) {
            auto k = ldpc::lms_layered_kernel<kRHM, kRWM, true>;
            if (int rc = set_lds_limit(k, c->lds_bytes)) return rc;
            hipLaunchKernelGGL(k, grid, block, c->lds_bytes, stream, a);
        } else {
            auto k = ldpc::lms_layered_kernel<kRHM, kRWM, false>;
            if (int rc = set_lds_limit(k, c->lds_bytes)) return rc;
            hipLaunchKernelGGL(k, grid, block, c->lds_bytes, stream, a);
        }
        break;
    case LDPC_HIP_TASP_DEC: {
        ldpc_spec::SpecArgs sa{};
        sa.llr = d_llr; sa.hard = d_hard; sa.iters = d_iters; sa.soft_out = d_soft; sa.maxiter = maxiter; sa.alpha = alpha;
        if (B > 0x7fffffffLL) return fail(LDPC_HIP_EINVAL, "batch too large");
        if (c->spec_aot && c->M == 64) hipLaunchKernelGGL(tasp_spec_appendix_c_m64_kernel, dim3((unsigned)B), dim3(64), c->lds_bytes, stream, sa);
        else if (c->spec_aot) hipLaunchKernelGGL(tasp_spec_appendix_c_m126_kernel, dim3((unsigned)B), dim3(128), c->lds_bytes, stream, sa);
        else {
            void *kargs[] = {&sa};
            HIP_TRY(hipModuleLaunchKernel(c->jit->fn, (unsigned)B, 1, 1, (unsigned)c->spec_threads, 1, 1, (unsigned)c->lds_bytes, stream, kargs, nullptr));
        }
        break;
    }
    case LDPC_HIP_IMS_DEC:
        if (c->multiwave) {
            auto k = ldpc::ims_flood_kernel<kRHM, kNHM, true>;
            if (int rc = set_lds_limit(k, c->lds_bytes)) return rc;
            hipLaunchKernelGGL(k, grid, block, c->lds_bytes, stream, a);
        } else {
            auto k = ldpc::ims_flood_kernel<kRHM, kNHM, false>;
            hipLaunchKernelGGL(k, grid, block, c->lds_bytes, stream, a);
        }
        break;
    case LDPC_HIP_SP_DEC:
        if (c->spec_aot || c->jit) {
            ldpc_spec::SpecArgs sa{};
            sa.llr = d_llr; sa.hard = d_hard; sa.iters = d_iters; sa.soft_out = d_soft; sa.maxiter = maxiter; sa.alpha = alpha;
            if (B > 0x7fffffffLL) return fail(LDPC_HIP_EINVAL, "batch too large");
            if (c->spec_aot && c->fast_variant == 3) {
                if (int rc = set_lds_limit(sp_spec_appendix_c_m64_occ2_kernel, c->lds_bytes)) return rc;
                hipLaunchKernelGGL(sp_spec_appendix_c_m64_occ2_kernel, dim3((unsigned)B), dim3(512), c->lds_bytes, stream, sa);
            } else if (c->spec_aot) {
                if (int rc = set_lds_limit(sp_spec_appendix_c_m64_kernel, c->lds_bytes)) return rc;
                hipLaunchKernelGGL(sp_spec_appendix_c_m64_kernel, dim3((unsigned)B), dim3(512), c->lds_bytes, stream, sa);
            } else {
                void *kargs[] = {&sa};
                HIP_TRY(hipModuleLaunchKernel(c->jit->fn, (unsigned)B, 1, 1, 512, 1, 1, (unsigned)c->lds_bytes, stream, kargs, nullptr));
            }
            break;
        }
        {
        auto k = ldpc::sp_flood_kernel;
        if (int rc = set_lds_limit(k, c->lds_bytes)) return rc;
        hipLaunchKernelGGL(k, grid, block, c->lds_bytes, stream, a);
        break;
    }
    default:
        return fail(LDPC_HIP_EUNSUPPORTED, "decoder %d", c->decoder_id);
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
    const bool sp = c->decoder_id == LDPC_HIP_SP_DEC;
    const bool tasp = c->decoder_id == LDPC_HIP_TASP_DEC;
    if (tasp) decision = 0;  // upstream ignores `decision` for this decoder: the result is always hard (decoders.cpp:2737-2738)
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
    if (tasp && clobber_sp_input) {  // decoders.cpp:2611-2618: soft[] is left holding P(bit = 1) of the channel
        for (size_t i = 0; i < nllr; ++i) {
            const double x = llr[i] * 0.5;
            const double y = x < 20.0 ? (x < -20.0 ? -20.0 : x) : 20.0;
            const double e0 = std::exp(y), e1 = std::exp(-y);
            llr[i] = e1 / (e0 + e1);
        }
    }
    return 0;
}

static int awgn_sigma(const ldpc_hip_ctx *c, double snr_db, int modulation_type, int punctured_blocks, double *sigma) {
    const int b = c->rh, cc = c->nh;
    if (punctured_blocks < 0 || punctured_blocks >= cc) return fail(LDPC_HIP_EINVAL, "punctured_blocks=%d", punctured_blocks);
    const double bitrate = (double)(cc - b) / (cc - punctured_blocks);  // bp_simulation.cpp:444
    if (modulation_type == 0) {
        *sigma = std::sqrt(std::pow(10, -snr_db / 10) / 2 / bitrate);    // :445
    } else if (modulation_type == 1 || modulation_type == 2) {
        const int QAM = modulation_type == 1 ? 4 : 16, halfmlog = modulation_type == 1 ? 1 : 2;
        const double norm_factor = 2.0 * (QAM - 1.0) / 3.0;              // :447
        *sigma = std::sqrt(std::pow(10., -snr_db / 10.) / (2 * bitrate * halfmlog * 2) * norm_factor);  // :449
    } else {
        return fail(LDPC_HIP_EUNSUPPORTED, "modulation_type %d", modulation_type);
    }
    return 0;
}

int ldpc_hip_awgn_llr_dev(ldpc_hip_ctx *c, double snr_db, int modulation_type, int punctured_blocks, uint64_t seed,
                          long long first_frame, long long B, double *d_llr, void *stream_) {
    if (!c || !d_llr || B < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_awgn_llr_dev: bad argument");
    if (modulation_type != 0 && modulation_type != 1) return fail(LDPC_HIP_EUNSUPPORTED, "modulation_type %d (0 BPSK, 1 QAM4)", modulation_type);
    if (B == 0) return 0;
    if (int rc = set_device(c)) return rc;
    ldpc::AwgnArgs a{};
    if (int rc = awgn_sigma(c, snr_db, modulation_type, punctured_blocks, &a.sigma)) return rc;
    a.llr = d_llr; a.B = B; a.first_frame = first_frame; a.N = c->N;
    a.punct_start = c->N - c->M * punctured_blocks;
    a.punct_val = (c->decoder_id == LDPC_HIP_SP_DEC || c->decoder_id == LDPC_HIP_TASP_DEC) ? 0.0 : 0.5;  // :700 (sic), out_type :451-466
    a.seed = seed;
    const long long total = B * (long long)((c->N + 1) / 2);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(ldpc::awgn_llr_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ldpc_hip_awgn_qam16_llr_dev(ldpc_hip_ctx *c, double snr_db, double T, uint64_t seed, long long first_frame,
                                long long B, double *d_llr, void *stream_) {
    if (!c || !d_llr || B < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_awgn_qam16_llr_dev: bad argument");
    if (c->N % 4) return fail(LDPC_HIP_EUNSUPPORTED, "16-QAM needs N %% 4 == 0 (N=%d)", c->N);
    if (B == 0) return 0;
    if (int rc = set_device(c)) return rc;
    ldpc::Qam16Args a{};
    if (int rc = awgn_sigma(c, snr_db, 2, 0, &a.sigma)) return rc;
    a.llr = d_llr; a.B = B; a.first_frame = first_frame; a.N = c->N; a.T = T; a.seed = seed;
    const long long total = B * (long long)(c->N / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(ldpc::awgn_qam16_llr_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ldpc_hip_qam_demod_dev(int Q, double T, double sigma, const double *d_x, long long ns, double *d_out,
                           int out_type, int device, void *stream_) {
    if (!d_x || !d_out || ns < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_qam_demod_dev: bad argument");
    if (Q != 4 && Q != 16) return fail(LDPC_HIP_EUNSUPPORTED, "QAM-%d demapper not built (4, 16)", Q);
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

int ldpc_hip_count_errors_dev(ldpc_hip_ctx *c, const uint32_t *d_hard, const int32_t *d_iters, long long B,
                              int32_t *d_frame_info, unsigned long long *d_counters, void *stream_) {
    if (!c || !d_hard || !d_iters || !d_counters || B < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_count_errors_dev: bad argument");
    if (B == 0) return 0;
    if (int rc = set_device(c)) return rc;
    ldpc::CountArgs a{};
    a.hard = d_hard; a.iters = d_iters; a.frame_info = d_frame_info; a.counters = d_counters;
    a.B = B; a.hard_words = c->hard_words; a.R = c->R;
    long long blocks = (B + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(ldpc::count_errors_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ldpc_hip_simulate(ldpc_hip_ctx *c, double snr_db, int modulation_type, int punctured_blocks, int maxiter,
                      double alpha, uint64_t seed, long long first_frame, long long B, unsigned long long counters[4],
                      unsigned long long *sum_abs_iters) {
    if (!c || !counters || B < 0) return fail(LDPC_HIP_EINVAL, "ldpc_hip_simulate: bad argument");
    if (int rc = set_device(c)) return rc;
    const long long chunk_max = 1 << 16;
    const long long chunk = B < chunk_max ? B : chunk_max;
    if (chunk > 0) { if (int rc = ensure_workspace(c, chunk, false)) return rc; }
    HIP_TRY(hipMemsetAsync(c->w_counters, 0, sizeof(unsigned long long) * 8, nullptr));
    for (long long done = 0; done < B; done += chunk) {
        const long long nb = (B - done) < chunk ? (B - done) : chunk;
        int rc;
        if (modulation_type == 2) rc = ldpc_hip_awgn_qam16_llr_dev(c, snr_db, 26.0, seed, first_frame + done, nb, c->w_llr, nullptr);
        else rc = ldpc_hip_awgn_llr_dev(c, snr_db, modulation_type, punctured_blocks, seed, first_frame + done, nb, c->w_llr, nullptr);
        if (rc) return rc;
        if ((rc = ldpc_hip_decode_dev(c, c->w_llr, nb, maxiter, alpha, c->w_hard, c->w_iters, nullptr, nullptr))) return rc;
        if ((rc = ldpc_hip_count_errors_dev(c, c->w_hard, c->w_iters, nb, nullptr, c->w_counters, nullptr))) return rc;
    }
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
