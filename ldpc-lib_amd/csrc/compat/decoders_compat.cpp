// decoders_compat.cpp -- upstream's decoder call surface (decoders.h:293-308) on top of the C-ABI in ldpc_hip.h.
//
// Compiles against EITHER include/ldpc/decoders.h (standalone) or upstream's own decoders.h (drop-in: put the
// upstream tree first on the include path and build with -DLDPC_COMPAT_UPSTREAM_HEADERS).  Only members both
// headers share are touched; the private state lives in a side table keyed by the DEC_STATE pointer.
//
// Ownership and error convention follow decoders.cpp:348-791 (open), :1009-1207 (init), :1210-1376 (close):
// decod_open allocates codeword/y/decword/hd, the caller fills hd between open and init (bp_simulation.cpp:353-382),
// decod_init uploads the base matrix (ldpc_hip_open), decod_close frees everything including the struct.
#ifdef LDPC_COMPAT_UPSTREAM_HEADERS
#include "decoders.h"
#else
#include "ldpc/decoders.h"
#endif

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "ldpc_hip.h"

char const *const DEC_FULL_NAME[] = {  // decoders.cpp:18-30: the display names upstream's drivers print, by DEC_ID
    "Belief Propagation", "Sum-Product", "Advanced Sum-Product", "Min-Sum", "Integer Min-Sum",
    "Integer Advanced Sum-Product", "FHT Sum-Product", "TDMP Advanced Sum-Product", "Layered Min-Sum",
    "Low complexity-high efficiency",
};

namespace {

struct Impl {
    ldpc_hip_ctx *ctx = nullptr;
    int device = 0;
};
std::mutex g_mu;
std::unordered_map<const void *, Impl> g_impl;

Impl *impl_of(const void *st) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_impl.find(st);
    return it == g_impl.end() ? nullptr : &it->second;
}

[[noreturn]] void not_built(const char *what) {
    fprintf(stderr, "%s is not built in ldpc-lib_amd (built decoders: BP_DEC=0, SP_DEC=1, ASP_DEC=2, MS_DEC=3, IMS_DEC=4, TASP_DEC=7, LMS_DEC=8)\n", what);
    exit(1);  // upstream's die() convention (commons_portable.cpp:181-189)
}

short **alloc2d_short(int rows, int cols) {  // one contiguous block + row pointers, like decoders.cpp:245-262
    short **p = (short **)calloc(rows, sizeof(short *));
    short *d = (short *)calloc((size_t)rows * cols, sizeof(short));
    if (!p || !d) { free(p); free(d); return nullptr; }
    for (int i = 0; i < rows; ++i) p[i] = d + (size_t)i * cols;
    return p;
}

int decode_common(DEC_STATE *st, int expect_id, double *soft, double *decword, int *iters, long long B, int maxiter,
                  int decision, double alpha) {
    Impl *im = impl_of(st);
    if (!st || !im || !im->ctx) { fprintf(stderr, "decoder called on a state that was not decod_init()ed\n"); exit(1); }
    if (st->codec_id != expect_id) { fprintf(stderr, "decoder %d called on a state opened for decoder %d\n", expect_id, st->codec_id); exit(1); }
    std::vector<int32_t> it((size_t)B);
    const int rc = ldpc_hip_decode_host(im->ctx, soft, B, maxiter, decision, alpha, decword, it.data(), /*clobber_sp_input=*/1);
    if (rc != 0) { fprintf(stderr, "ldpc_hip: %s\n", ldpc_hip_last_error()); exit(1); }
    if (iters) for (long long b = 0; b < B; ++b) iters[b] = it[(size_t)b];
    return it.empty() ? 0 : it[0];
}

}  // namespace

DEC_STATE *decod_open(int codec_id, int q_bits, int mh, int nh, int M) {
    if (codec_id != SP_DEC && codec_id != MS_DEC && codec_id != LMS_DEC && codec_id != IMS_DEC && codec_id != TASP_DEC && codec_id != ASP_DEC && codec_id != BP_DEC) {
        fprintf(stderr, "decod_open: decoder id %d is not built in ldpc-lib_amd (built: BP_DEC=0, SP_DEC=1, ASP_DEC=2, MS_DEC=3, IMS_DEC=4, TASP_DEC=7, LMS_DEC=8)\n", codec_id);
        return NULL;  // decoders.cpp:786: unknown id -> NULL
    }
    if (mh <= 0 || nh <= 0 || M <= 0) return NULL;
    DEC_STATE *st = (DEC_STATE *)calloc(1, sizeof(DEC_STATE));
    if (!st) return NULL;
    const int N = nh * M;
    st->q_bits = q_bits; st->nh = nh; st->rh = mh; st->m = M; st->n = N;
    st->codec_id = codec_id; st->bin_codec = 1; st->q = 1;
    st->codeword = (int *)malloc(sizeof(int) * N);
    st->y = (double *)calloc(N, sizeof(double));
    st->decword = (double *)calloc(N, sizeof(double));
    st->syndr = (short *)calloc((size_t)mh * M, sizeof(short));
    st->hd = alloc2d_short(mh, nh);
    if (!st->codeword || !st->y || !st->decword || !st->syndr || !st->hd) { decod_close(st); return NULL; }
    for (int i = 0; i < mh; ++i) for (int j = 0; j < nh; ++j) st->hd[i][j] = -1;
    std::lock_guard<std::mutex> lk(g_mu);
    g_impl[st] = Impl();
    const char *dev = getenv("LDPC_HIP_DEVICE");
    g_impl[st].device = dev ? atoi(dev) : 0;
    return st;
}

int decod_init(void *state) {
    if (!state) return 1;  // sic: decoders.cpp:1014-1015 reports success for a NULL state
    DEC_STATE *st = (DEC_STATE *)state;
    Impl *im = impl_of(st);
    if (!im) return 0;
    if (im->ctx) { ldpc_hip_close(im->ctx); im->ctx = nullptr; }  // re-init after the caller changed hd
    std::vector<int16_t> hd((size_t)st->rh * st->nh);
    for (int i = 0; i < st->rh; ++i) for (int j = 0; j < st->nh; ++j) hd[(size_t)i * st->nh + j] = st->hd[i][j];
    const int rc = ldpc_hip_open(st->codec_id, st->rh, st->nh, st->m, hd.data(), im->device, &im->ctx);
    if (rc != 0) { fprintf(stderr, "decod_init: %s\n", ldpc_hip_last_error()); return 0; }
    return 1;
}

void decod_close(DEC_STATE *st) {
    if (!st) return;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_impl.find(st);
        if (it != g_impl.end()) {
            if (it->second.ctx) ldpc_hip_close(it->second.ctx);
            g_impl.erase(it);
        }
    }
    if (st->hd) { free(st->hd[0]); free(st->hd); }
    free(st->codeword); free(st->y); free(st->decword); free(st->syndr);
    free(st);
}

struct ldpc_hip_ctx *ldpc_decod_ctx(DEC_STATE *st) {
    Impl *im = impl_of(st);
    return im ? im->ctx : nullptr;
}

int min_sum_decod_qc_lm(DEC_STATE *st, double y[], double decword[], int maxsteps, int decision, double alpha) {
    return decode_common(st, MS_DEC, y, decword, nullptr, 1, maxsteps, decision, alpha);
}

int lmin_sum_decod_qc_lm(DEC_STATE *st, double y[], double decword[], int maxsteps, int decision, double, double) {
    return decode_common(st, LMS_DEC, y, decword, nullptr, 1, maxsteps, decision, 0.0);  // alpha, beta dead upstream
}

int sum_prod_decod_qc_lm(DEC_STATE *st, double soft[], double decword[], int maxiter, int decision) {
    return decode_common(st, SP_DEC, soft, decword, nullptr, 1, maxiter, decision, 0.0);  // soft[] is clobbered
}

int ldpc_decod_batch(DEC_STATE *st, double *soft, double *decword, int *iters, long long B, int maxiter, int decision) {
    return decode_common(st, st ? st->codec_id : -1, soft, decword, iters, B, maxiter, decision, MS_ALPHA);
}

int bp_decod_qc_lm(DEC_STATE *st, double soft[], double decword[], int maxiter, int decision) {
    // soft[] is upstream's working array: it comes back holding the a-posteriori LLRs.  The context keeps the syndrome
    // the previous call left behind, like upstream's uncleared st->syndr (decoders.cpp:1742-1762).
    return decode_common(st, BP_DEC, soft, decword, nullptr, 1, maxiter, decision, 0.0);
}
int sum_prod_gf2_decod_qc_lm(DEC_STATE *st, double soft[], double decword[], int maxsteps, int decision) {
    return decode_common(st, ASP_DEC, soft, decword, nullptr, 1, maxsteps, decision, 0.0);  // soft[] is clobbered with P(bit=1)
}
int imin_sum_decod_qc_lm(DEC_STATE *st, double y[], double decword[], int maxsteps, int decision, double alpha, double thr,
                         int qbits, int dbits) {
    Impl *im = impl_of(st);
    if (!im || !im->ctx || ldpc_hip_set_ims_params(im->ctx, thr, qbits, dbits) != 0) {
        fprintf(stderr, "imin_sum_decod_qc_lm: %s\n", im && im->ctx ? ldpc_hip_last_error() : "state not initialised");
        exit(1);
    }
    return decode_common(st, IMS_DEC, y, decword, nullptr, 1, maxsteps, decision, alpha);
}
int isum_prod_gf2_decod_qc_lm(DEC_STATE *, double[], double[], int, int) { not_built("isum_prod_gf2_decod_qc_lm (IASP_DEC)"); }
int sum_prod_gfq_decod_lm(DEC_STATE *, double *[], short *, double *[], int, double) { not_built("sum_prod_gfq_decod_lm (FHT_DEC)"); }
int tdmp_sum_prod_gf2_decod_qc_lm(DEC_STATE *st, double soft[], double decword[], int maxsteps, int decision) {
    return decode_common(st, TASP_DEC, soft, decword, nullptr, 1, maxsteps, decision, 0.0);  // soft[] is clobbered, result always hard
}
int lche_decod(DEC_STATE *, double[], double[], int, int) { not_built("lche_decod (LCHE_DEC)"); }
int encode_NBQCLDPC(DEC_STATE *, int *) { not_built("encode_NBQCLDPC (GF(q) encoder)"); }
void left2right(short **, int, int) { not_built("left2right (GF(q) only)"); }
