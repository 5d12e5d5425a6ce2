// ref_driver.cpp -- thin extern "C" driver around the UPSTREAM reference decoders.
//
// TEST INFRASTRUCTURE ONLY.  This file is our own code; it is compiled together with the reference's
// own sources *where they lie* (/root/reference/{decoders,QAM_modulator,QAM_demodulator,direct_inverse_perm}.cpp, flags
// -O3 -DSKIP_MEX as in the reference Makefile:20-21) into oracle/_ref/libldpc_ref.so by oracle/Makefile.
// Nothing from the reference is copied into this repository and the .so never ships in git
// (oracle/_ref/ is git-ignored).  It exists only in the build container; tests that need it skip
// when it is absent (e.g. on the GPU box when it was not prebuilt).
//
// The call sequence mirrors bp_simulation.cpp:353-382 (open, fill hd, init) and the decoder
// dispatch at bp_simulation.cpp:716-729.
#include <cstring>

#include "decoders.h"    // from -I/root/reference
#include "modulation.h"  // from -I/root/reference

extern "C" {

void *ref_open(int dec_id, int rh, int nh, int M, const short *hd) {
    DEC_STATE *st = decod_open(dec_id, /*q_bits=binlog(2)*/ 1, rh, nh, M);
    if (!st) return nullptr;
    for (int i = 0; i < rh; i++)
        for (int j = 0; j < nh; j++) st->hd[i][j] = hd[i * nh + j];
    if (!decod_init(st)) { decod_close(st); return nullptr; }
    return st;
}

void ref_close(void *h) {
    if (h) decod_close((DEC_STATE *)h);
}

// y_inout: N doubles, copied into st->y first (SP clobbers its input; the clobbered array is copied back
// so the caller can inspect it).  decword: N doubles out.  Returns the decoder's signed iteration count.
int ref_decode(void *h, int dec_id, double *y_inout, double *decword, int maxiter, int decision) {
    DEC_STATE *st = (DEC_STATE *)h;
    const int N = st->n;
    int iter = -12345;
    std::memcpy(st->y, y_inout, sizeof(double) * N);
    switch (dec_id) {
    case BP_DEC:   iter = bp_decod_qc_lm(st, st->y, st->decword, maxiter, decision); break;
    case SP_DEC:   iter = sum_prod_decod_qc_lm(st, st->y, st->decword, maxiter, decision); break;
    case ASP_DEC:  iter = sum_prod_gf2_decod_qc_lm(st, st->y, st->decword, maxiter, decision); break;
    case MS_DEC:   iter = min_sum_decod_qc_lm(st, st->y, st->decword, maxiter, decision, MS_ALPHA); break;
    case IMS_DEC:  iter = imin_sum_decod_qc_lm(st, st->y, st->decword, maxiter, decision, MS_ALPHA, MS_THR, MS_QBITS, MS_DBITS); break;
    case TASP_DEC: iter = tdmp_sum_prod_gf2_decod_qc_lm(st, st->y, st->decword, maxiter, decision); break;
    case LMS_DEC:  iter = lmin_sum_decod_qc_lm(st, st->y, st->decword, maxiter, decision, MS_ALPHA, MS_BETA); break;
    default: return -12345;
    }
    std::memcpy(decword, st->decword, sizeof(double) * N);
    std::memcpy(y_inout, st->y, sizeof(double) * N);
    return iter;
}

// QAM mapper: in = nbits doubles (0/1), out = 2*ns doubles, returns ns.
int ref_qam_modulate(int Q, const double *bits, int nbits, double *out) {
    int m = 0; while ((1 << m) < Q) m++;
    QAM_MODULATOR_STATE *st = QAM_modulator_open(Q, nbits, m);
    if (!st) return -1;
    int ns = st->ns;
    double *in = new double[st->Lfact]();
    std::memcpy(in, bits, sizeof(double) * nbits);
    QAM_modulator(st, in, out);
    delete[] in;
    QAM_modulator_close(st);
    return ns;
}

// Soft demapper, called the way its own MEX wrapper does (m = log2(Q)), not the way bp_simulation.cpp:471
// does (which passes the lifting as m -- SURVEY Appendix B Q5).
void ref_qam_demodulate(int Q, double T, double sigma, const double *x, int ns, double *out, int out_type) {
    int m = 0; while ((1 << m) < Q) m++;
    QAM_DEMODULATOR_STATE *st = QAM_demodulator_open(T, sigma, (short)Q, ns * m, m, ns, out_type);
    if (!st) return;
    Demodulate(st, const_cast<double *>(x), out);
    QAM_demodulator_close(st);
}

// Interleaver (direct_inverse_perm.cpp): the gather maps of both directions, read off by permuting the ramp 0..N-1.
// hd row-major b x c.  Returns 0, or -1 when upstream refuses to open.
int ref_perm_maps(int b, int c, int M, int QAM, int halfmlog, int mode, int block_size, int step_size, const short *hd,
                  int *direct, int *inverse) {
    PERMSTATE *st = Permutations_Open(b, c, M, QAM, halfmlog, mode, block_size, step_size);
    if (!st) return -1;
    short **rows = new short *[b];
    for (int i = 0; i < b; i++) rows[i] = const_cast<short *>(hd) + (size_t)i * c;
    Permutation_Init(st, rows);
    const int N = c * M;
    double *in = new double[N + N], *out = in + N;
    for (int i = 0; i < N; i++) in[i] = i;
    for (int dir = 0; dir < 2; dir++) {
        for (int i = 0; i < N; i++) out[i] = -1;
        Permutation(st, dir, in, out);
        for (int i = 0; i < N; i++) (dir ? inverse : direct)[i] = (int)out[i];
    }
    delete[] in;
    delete[] rows;
    Permutations_Close(st);
    return 0;
}

}  // extern "C"
