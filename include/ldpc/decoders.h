/*
 * ldpc/decoders.h -- source-compatible stand-in for upstream's decoders.h (eovs/ldpc-lib decoders.h:1-312) for
 * programs built WITHOUT the upstream tree.  Same enum, constants, function names, argument meaning, return
 * values and error behaviour for the decoders on the accelerated path; DEC_STATE exposes exactly the members that
 * upstream callers touch (bp_simulation.cpp:361-388,554,670,684,702-709,718-727,736,748).
 *
 * When you build INSIDE the upstream tree, keep upstream's own decoders.h on the include path instead: the
 * implementation (csrc/compat/decoders_compat.cpp) only uses members both headers share and keeps its private
 * state (the ldpc_hip context) in a side table keyed by the DEC_STATE pointer, so it works with either layout.
 * See INTEGRATION.md.
 */
#ifndef LDPC_COMPAT_DECODERS_H_
#define LDPC_COMPAT_DECODERS_H_

#define DEC_DECISION 0  /* 0 - hard decision, 1 - soft decision        (decoders.h:7)  */

enum DEC_ID {           /* decoders.h:16-28 */
    BP_DEC, SP_DEC, ASP_DEC, MS_DEC, IMS_DEC, IASP_DEC, FHT_DEC, TASP_DEC, LMS_DEC, LCHE_DEC
};

extern char const *const DEC_FULL_NAME[];  /* decoders.cpp:18-30 */

#define SKIP (-1)
#define MS_ALPHA 0.8    /* decoders.h:43 */
#define MS_BETA 0.4     /* decoders.h:44 (dead: lmin_sum hard-codes 0.4, decoders.cpp:5163) */
#define MS_THR 1.4
#define MS_QBITS 6
#define MS_DBITS (MS_QBITS + 2)

typedef struct {
    int q_bits, q;
    int nh, rh, m, n;      /* block columns, block rows, lifting, code length */
    int maxiter, codec_id, bin_codec;
    short **hd;            /* [rh][nh] circulant shifts, -1 empty; filled by the caller between open and init */
    short **hb, **hc;      /* non-binary only: always NULL here */
    int *codeword;         /* [n] */
    double *y;             /* [n] decoder input  (LLR, positive = bit 0) */
    double *decword;       /* [n] decoder output */
    short *syndr;          /* [rh*m] */
    double **qy, **qdecword;  /* non-binary only: NULL */
    short *qhard;
    int fht_ncols2convert;
} DEC_STATE;

/* decoders.h:293-308.  decod_open: NULL on unknown/unbuilt decoder id or allocation failure; decod_init: 1 ok,
 * 0 failure, and 1 for a NULL state (sic, decoders.cpp:1014-1015); decoders: >0 converged after that many
 * iterations, 0 input already a codeword (sum-product), <0 = -(iterations run). */
DEC_STATE *decod_open(int decoder_id, int q_bits, int mh, int nh, int M);
int decod_init(void *st);
void decod_close(DEC_STATE *st);
int sum_prod_decod_qc_lm(DEC_STATE *st, double soft[], double decword[], int maxiter, int decision);
int min_sum_decod_qc_lm(DEC_STATE *st, double soft[], double decword[], int maxiter, int decision, double alpha);
int lmin_sum_decod_qc_lm(DEC_STATE *st, double soft[], double decword[], int maxiter, int decision, double alpha, double beta);
int bp_decod_qc_lm(DEC_STATE *st, double soft[], double decword[], int maxiter, int decision);          /* keeps st's syndrome between calls like upstream */
int sum_prod_gf2_decod_qc_lm(DEC_STATE *st, double soft[], double decword[], int maxiter, int decision);
int imin_sum_decod_qc_lm(DEC_STATE *st, double soft[], double decword[], int maxiter, int decision, double alpha, double thr, int qbits, int dbits);
int tdmp_sum_prod_gf2_decod_qc_lm(DEC_STATE *st, double soft[], double decword[], int maxiter, int decision);
/* present for link compatibility with upstream's dispatch (bp_simulation.cpp:716-729); not built: they die() */
int isum_prod_gf2_decod_qc_lm(DEC_STATE *st, double soft[], double decword[], int maxiter, int decision);
int sum_prod_gfq_decod_lm(DEC_STATE *st, double *soft[], short *qhard, double *decword[], int maxiter, double p_thr);
int lche_decod(DEC_STATE *st, double soft[], double decword[], int maxiter, int decision);
int encode_NBQCLDPC(DEC_STATE *st, int *msg);
void left2right(short **matr, int nrow, int ncol);

/* Batched extension (not in upstream): decode B frames laid out [B][n] in one GPU launch.  Same semantics per frame. */
int ldpc_decod_batch(DEC_STATE *st, double *soft, double *decword, int *iters, long long B, int maxiter, int decision);
/* The ldpc_hip context behind a state (NULL before decod_init). */
struct ldpc_hip_ctx;
struct ldpc_hip_ctx *ldpc_decod_ctx(DEC_STATE *st);

#endif
