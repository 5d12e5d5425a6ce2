/*
 * ldpc_oracle.h -- CPU restatement of the reference's binary QC-LDPC decoders.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker (or as the timed CPU baseline), never as a fallback.
 *
 * Each function cites the reference file:line whose arithmetic it restates
 * (paths relative to the upstream eovs/ldpc-lib tree).  The restatement is pinned
 * against the compiled reference (oracle/_ref, built from the upstream sources where
 * they lie) by tests/test_oracle_vs_ref.py and by the golden vectors under
 * tests/golden/ that oracle/make_goldens.py produced from that build.
 *
 * Conventions (decoders.cpp:327-346 `rotate`, bp_simulation.cpp:54):
 *   base matrix hd[j*nh+k], -1 = empty circulant, else shift 0 <= c < M
 *   check (j,n) is connected to variable (k, (n+c) mod M)
 *   variable (k,i) lives at array index k*M+i; check (j,n) at j*M+n
 *   positive LLR  <=>  bit 0
 */
#ifndef LDPC_ORACLE_H
#define LDPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_code orc_code;

/* decoders.cpp:348 decod_open + bp_simulation.cpp:359-361 (hd fill). hd is row-major rh x nh. */
orc_code *orc_open(int rh, int nh, int M, const short *hd);
void orc_close(orc_code *c);
int orc_n(const orc_code *c);
int orc_r(const orc_code *c);
int orc_edges(const orc_code *c); /* number of non-empty circulants */

/* decoders.cpp:4554-4767 min_sum_decod_qc_lm (MS_MUL_CORRECTION variant).
 * y is not modified.  decword[v] = (soft<0) as 0.0/1.0 (decision==0) or the a-posteriori value.
 * returns iter+1 (>0) when the syndrome cleared, -maxsteps otherwise. */
int orc_min_sum(orc_code *c, const double *y, double *decword, int maxsteps, int decision, double alpha);

/* decoders.cpp:5064-5425 lmin_sum_decod_qc_lm (active branch :5106-5290 with MY_VERSION):
 * layered offset min-sum, beta = 0.4 literal (:5163).  alpha/beta arguments of the reference are dead. */
int orc_lmin_sum(orc_code *c, const double *y, double *decword, int maxsteps, int decision);

/* decoders.cpp:1923-2185 sum_prod_decod_qc_lm: likelihood-ratio-domain sum-product.
 * soft[] is CLOBBERED (overwritten with exp(clamped LLR) and then the a-posteriori ratios), as in the reference.
 * returns 0 if the input already is a codeword, iter (1-based) when corrected, -maxiter otherwise. */
int orc_sum_prod(orc_code *c, double *soft, double *decword, int maxiter, int decision);

/* decoders.cpp:5430-5690 imin_sum_decod_qc_lm: int16 normalised min-sum with 6-bit input quantiser. */
int orc_imin_sum(orc_code *c, const double *y, double *decword, int maxsteps, int decision,
                 double alpha, double thr, int qbits, int dbits);

/* decoders.cpp:2584-2744 tdmp_sum_prod_gf2_decod_qc_lm ("TASP", decoder id 7: the decoder_type of every shipped
 * scenario file): layered sum-product in the probability domain.  soft[] is CLOBBERED with P(bit=1) of the channel
 * (:2611-2618).  decword is ALWAYS the hard decision soft_out > 0.5 (the `decision` argument is dead, :2737-2738);
 * post_out (optional, may be NULL) receives the final a-posteriori probabilities for tolerance checks.
 * returns 0 if the input already is a codeword, steps (>0) when the syndrome cleared, -steps otherwise. */
int orc_tdmp_sum_prod(orc_code *c, double *soft, double *decword, int maxsteps, double *post_out);

/* decoders.cpp:2324-2581 sum_prod_gf2_decod_qc_lm ("ASP", decoder id 2): flooding sum-product in the probability
 * domain, general branch (:2482-2556; the all-columns-of-weight-2 shortcut :2431-2480 is not restated and such codes
 * are rejected with -9999).  soft[] is CLOBBERED with P(bit=1) of the channel (:2351-2358); decword = soft_out > 0.5
 * or soft_out (decision != 0).  returns 0 / steps+1 / -steps. */
int orc_sum_prod_gf2(orc_code *c, double *soft, double *decword, int maxsteps, int decision);

/* decoders.cpp:1708-1920 bp_decod_qc_lm ("BP", decoder id 0): Gallager belief propagation in the log domain.
 * soft[] is CLOBBERED (input clamp :1738, then the a-posteriori LLRs).  decword = soft < 0, or soft (decision != 0).
 * returns 0 (input check passed) / iterations / -iterations.  The input check XORs into the syndrome the previous call
 * on this state left behind: orc_bp_stale() exposes that carried [R] byte array. */
int orc_bp(orc_code *c, double *soft, double *decword, int maxiter, int decision);
unsigned char *orc_bp_stale(orc_code *c);

/* Syndrome of hard decisions (soft<0) : decoders.cpp:793-814 check_syndrome. returns 1 if any check fails. */
int orc_syndrome_nonzero(const orc_code *c, const double *soft);

/* QAM_modulator.cpp:69-194 (open + GrayPAM + QAM_modulator) for Q in {4,16,64,256}:
 * in: nbits values (0/1) ; out: 2*ceil(nbits/m) doubles, I/Q interleaved. returns number of symbols. */
int orc_qam_modulate(int Q, const double *bits, int nbits, double *out);

/* QAM_demodulator.cpp:99-566 Demodulate, out_type 0 (LLR = log(p1/p0)), for Q in {4,16}.
 * x: 2*ns doubles (I/Q interleaved), out: ns*m LLRs. sigma is the per-rail noise std, T the cut-off. */
void orc_qam_demodulate(int Q, double T, double sigma, const double *x, int ns, double *out, int out_type);

#ifdef __cplusplus
}
#endif
#endif
