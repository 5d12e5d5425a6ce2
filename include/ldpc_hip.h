/*
 * ldpc_hip.h -- C-ABI of the MI355X (gfx950) batched QC-LDPC belief-propagation path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  It is what a cgo /
 * ctypes / JNI / C++ caller binds.  The upstream library (eovs/ldpc-lib) is C++ with no FFI of its own; each
 * entry point below cites the upstream interface it replaces (file:line in the upstream tree).  The C++
 * source-compatible layer (include/ldpc/decoders.h, include/ldpc/bp_simulation.h) is a thin wrapper over
 * these functions; INTEGRATION.md shows how a maintainer links it.
 *
 * Conventions (identical to upstream, decoders.cpp:327-346 / bp_simulation.cpp:54,738):
 *   base matrix hd[j*nh+k], -1 = empty circulant, else shift 0 <= c < M (values are reduced mod M)
 *   check (j,n) is connected to variable (k,(n+c) mod M); variable (k,i) is LLR index k*M+i
 *   positive LLR <=> bit 0;  parity part = indices [0,R), information part = [R,N)
 *   decoder return value ("iters"): >0 converged after that many iterations, 0 input already a codeword
 *   (sum-product only), <0 = -(iterations run), not converged                      (decoders.cpp:4766,2184,5424)
 *
 * All functions return 0 on success and a negative LDPC_HIP_E* code on failure; ldpc_hip_last_error()
 * gives the message of the calling thread's last failure.  There is NO CPU fallback: if no HIP device /
 * code object is available every call fails loudly.
 */
#ifndef LDPC_HIP_H
#define LDPC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDPC_HIP_ABI_VERSION 1

/* decoders.h:16-28 enum DEC_ID (only the binary decoders on the hot path are built) */
#define LDPC_HIP_BP_DEC 0  /* bp_decod_qc_lm         decoders.cpp:1708 (Gallager BP, log domain) */
#define LDPC_HIP_SP_DEC 1  /* sum_prod_decod_qc_lm   decoders.cpp:1923 */
#define LDPC_HIP_ASP_DEC 2 /* sum_prod_gf2_decod_qc_lm decoders.cpp:2324 (probability-domain flooding sum-product) */
#define LDPC_HIP_MS_DEC 3  /* min_sum_decod_qc_lm    decoders.cpp:4554 */
#define LDPC_HIP_IMS_DEC 4 /* imin_sum_decod_qc_lm   decoders.cpp:5430 */
#define LDPC_HIP_TASP_DEC 7 /* tdmp_sum_prod_gf2_decod_qc_lm decoders.cpp:2584 (decoder_type of the shipped scenario files) */
#define LDPC_HIP_LMS_DEC 8 /* lmin_sum_decod_qc_lm   decoders.cpp:5064 */

#define LDPC_HIP_EINVAL (-1)    /* bad argument */
#define LDPC_HIP_EUNSUPPORTED (-2) /* decoder / code shape not built */
#define LDPC_HIP_EHIP (-3)      /* HIP runtime error (message has hipGetErrorString) */
#define LDPC_HIP_ENOMEM (-4)

typedef struct ldpc_hip_ctx ldpc_hip_ctx;

int ldpc_hip_abi_version(void);
const char *ldpc_hip_last_error(void);
int ldpc_hip_device_count(void);

/* Replaces decod_open() + the hd fill + decod_init()  (decoders.h:293-294, decoders.cpp:348,1009,
 * bp_simulation.cpp:353-382).  hd is row-major rh x nh.  device = HIP device ordinal.
 * *out receives the context; NULL on failure (upstream: decod_open returns NULL). */
int ldpc_hip_open(int decoder_id, int rh, int nh, int M, const int16_t *hd, int device, ldpc_hip_ctx **out);
/* Replaces decod_close() (decoders.h:295, decoders.cpp:1210). */
void ldpc_hip_close(ldpc_hip_ctx *ctx);

int ldpc_hip_n(const ldpc_hip_ctx *ctx);          /* N = nh*M */
int ldpc_hip_r(const ldpc_hip_ctx *ctx);          /* R = rh*M */
int ldpc_hip_edges(const ldpc_hip_ctx *ctx);      /* non-empty circulants */
int ldpc_hip_hard_words(const ldpc_hip_ctx *ctx); /* ceil(N/32): uint32 words per frame of packed hard bits */
/* Name of the decode kernel this context launches (code-specialised AOT / hiprtc instance, table-driven, generic). */
const char *ldpc_hip_kernel_name(const ldpc_hip_ctx *ctx);
/* Name of the kernel the last ldpc_hip_decode_dev call on this context launched.  It differs from ldpc_hip_kernel_name only
 * for IMS_DEC with parameters beyond int8 (MS_DBITS > 8 or alpha > 1), which run on the table-driven int32 kernel. */
const char *ldpc_hip_last_launch(const ldpc_hip_ctx *ctx);

/* Batched replacement of the decoder entry points (decoders.h:297,299,304; dispatch bp_simulation.cpp:716-729):
 *   MS_DEC : min_sum_decod_qc_lm(st, y, decword, maxiter, decision, alpha)
 *   LMS_DEC: lmin_sum_decod_qc_lm(st, y, decword, maxiter, decision, alpha, beta)   (alpha, beta dead upstream)
 *   SP_DEC : sum_prod_decod_qc_lm(st, soft, decword, maxiter, decision)
 *   IMS_DEC: imin_sum_decod_qc_lm(st, y, decword, maxiter, decision, alpha, thr, qbits, dbits)  (int16 min-sum)
 *   BP_DEC:  bp_decod_qc_lm(st, soft, decword, maxiter, decision)                  (d_soft = a-posteriori LLR)
 *   ASP_DEC: sum_prod_gf2_decod_qc_lm(st, soft, decword, maxiter, decision)       (d_soft = a-posteriori P(bit=1))
 *   TASP_DEC: tdmp_sum_prod_gf2_decod_qc_lm(st, soft, decword, maxiter, decision)  (d_soft = final P(bit=1); `decision` dead)
 * All pointers are DEVICE pointers on ctx's device; the work is enqueued on `stream` (a hipStream_t, NULL =
 * default stream) and is asynchronous.  maxiter must be >= 1 (upstream's behaviour for maxiter <= 0 is an artefact of
 * stale state and is not reproduced: LDPC_HIP_EINVAL).  LLRs must be finite.
 *   d_llr   [B][N] float64 in.   NOT modified (upstream SP clobbers its input; the clobbered values are what
 *           d_soft receives, see below)
 *   d_hard  [B][ceil(N/32)] uint32 out: bit (v%32) of word v/32 = hard decision of variable v
 *           (upstream decword[v] = soft<0, resp. soft<1.0 for SP), or NULL
 *   d_iters [B] int32 out: upstream return value, or NULL
 * Numerics: outputs equal upstream's on the same inputs bit for bit -- hard decisions, return values and soft values -- for
 * MS, LMS, IMS, SP, ASP and TASP (fp64 in upstream's operation order, no FMA contraction; exp() by glibc's algorithm);
 * BP: hard decisions and return values identical, soft values within rtol 1e-5 / atol 1e-7 (its exp / log are the device's).
 *   d_soft  [B][N] float64 out, or NULL: the a-posteriori values upstream writes to decword[] when decision==1
 *           (MS/LMS: LLR; SP: likelihood ratio = what upstream leaves in soft[])
 */
int ldpc_hip_decode_dev(ldpc_hip_ctx *ctx, const double *d_llr, long long B, int maxiter, double alpha,
                        uint32_t *d_hard, int32_t *d_iters, double *d_soft, void *stream);

/* BP_DEC only.  Upstream's bp_decod_qc_lm does not clear DEC_STATE::syndr before its input check (decoders.cpp:1742-1762),
 * so frame b's check sees the syndrome frame b-1 left behind (non-zero after a failed frame).  With the chain ON (default)
 * the frames of one ldpc_hip_decode_dev call are decoded as if one after the other on one DEC_STATE, continuing from the
 * last frame of the previous call on this context; ldpc_hip_decode_dev then synchronises the stream.  on == 0: every
 * frame starts from a zero syndrome (asynchronous, independent of batching).  reset_carry != 0 forgets the carried
 * syndrome (what decod_close + decod_open would do). */
int ldpc_hip_set_bp_chain(ldpc_hip_ctx *ctx, int on, int reset_carry);

/* Integer min-sum only: the quantiser arguments of imin_sum_decod_qc_lm (decoders.h:300; defaults MS_THR 1.4,
 * MS_QBITS 6, MS_DBITS 8 of decoders.h:46-48).  `alpha` of the decode calls gives ialpha = (int)(alpha*16). */
int ldpc_hip_set_ims_params(ldpc_hip_ctx *ctx, double thr, int qbits, int dbits);

/* Same with HOST pointers, laid out exactly like upstream's per-frame arrays (PCIe-inclusive, synchronous):
 *   llr [B][N] in (for SP it is overwritten like upstream's soft[] when clobber_sp_input != 0),
 *   decword [B][N] float64 out (0.0/1.0 when decision==0, a-posteriori values when decision==1), iters [B]. */
int ldpc_hip_decode_host(ldpc_hip_ctx *ctx, double *llr, long long B, int maxiter, int decision, double alpha,
                         double *decword, int32_t *iters, int clobber_sp_input);

/* Device-side channel front end, replaces bp_simulation.cpp:444-449,600-612,697-710 for the all-zero codeword
 * (bp_simulation.cpp:568): llr = -2*(sigma*g - 1)/sigma^2, g ~ N(0,1) from a counter-based Philox4x32-10 stream
 * keyed by (seed, global frame index, variable index), so the result does not depend on batch split or GPU
 * count.  modulation_type 0 = BPSK ("SKIP"), 1 = QAM4 (upstream formula with sigmaQAM).  The last
 * M*punctured_blocks LLRs are set to 0.5 (LLR-type decoders) or 0 (SP) exactly as upstream (sic, Q7). */
int ldpc_hip_awgn_llr_dev(ldpc_hip_ctx *ctx, double snr_db, int modulation_type, int punctured_blocks,
                          uint64_t seed, long long first_frame, long long B, double *d_llr, void *stream);

/* 16-QAM chain (QAM_modulator.cpp:142, bp_simulation.cpp:447-449,621-628, QAM_demodulator.cpp:203-275) as
 * evidently intended upstream: all-zero codeword -> constellation points, fresh
 * x + sigmaQAM*g per frame, per-rail soft demap with cut-off T, negated.  N must be a multiple of 4. */
int ldpc_hip_awgn_qam16_llr_dev(ldpc_hip_ctx *ctx, double snr_db, double T, uint64_t seed, long long first_frame,
                                long long B, double *d_llr, void *stream);
/* The same chain for modulation_type 2 (QAM16), 3 (QAM64), 4 (QAM256) (enum MODULATION_TYPE, modulation.h:4-11;
 * QAM_demodulator.cpp:203-561; sigmaQAM bp_simulation.cpp:447-449).  When N is not a multiple of log2 Q the last symbol is
 * padded with zero bits like bp_simulation.cpp:575 and only its first bits are written.  Same noise keys as the 16-QAM entry. */
int ldpc_hip_awgn_qam_llr_dev(ldpc_hip_ctx *ctx, int modulation_type, double snr_db, double T, uint64_t seed, long long first_frame,
                              long long B, double *d_llr, void *stream);

/* Function-level soft demapper: QAM_demodulator.cpp:99-566 Demodulate() for Q in {4,16,64,256}, out_type 0/1 (Q = 4: out_type 0 only).
 * d_x [ns][2] (I,Q interleaved) -> d_out [ns][log2 Q]. */
int ldpc_hip_qam_demod_dev(int Q, double T, double sigma, const double *d_x, long long ns, double *d_out,
                           int out_type, int device, void *stream);

/* Systematic encoder for the dual-diagonal QC-LDPC codes upstream's search produces (qc_encode / random_codeword,
 * bp_simulation.cpp:22-191, with the information bits given instead of drawn).  HOST function.  info_bits: (nh-rh)*M bytes
 * (0/1) for variable positions [rh*M, nh*M); codeword: nh*M bytes out, parity first.  Upstream's simulation never transmits
 * the result (it zeroes the codeword, :568); this entry exists for callers that do and for the sign-symmetry tests. */
int ldpc_hip_encode_host(int rh, int nh, int M, const int16_t *hd, const uint8_t *info_bits, uint8_t *codeword);

/* Bit interleavers of the simulation chain (Permutations_Open / Permutation_Init / Permutation,
 * direct_inverse_perm.cpp:139-900; permutation_type of bp_simulation.h:21-23): mode 0 identity, 1 random, 2 deterministic,
 * 3 block random (block_size), 4 interleaved random (step_size); halfmlog = 1 (BPSK / QAM4), 2, 3, 4 (QAM16 / 64 / 256).
 * HOST function (no GPU needed): fills the two gather maps, out[i] = in[map[i]], direct[N] (encoder -> mapper side) and
 * inverse[N] (demapper -> decoder side), N = c*M, identical to upstream's for the same arguments (same LCG, same order). */
int ldpc_hip_interleaver_build(int b, int c, int M, int halfmlog, int mode, int block_size, int step_size, const int16_t *hd,
                               int32_t *direct, int32_t *inverse);
/* Applies a map to B frames resident on the device: d_out[f][i] = d_in[f][d_map[i]] (d_in != d_out). */
int ldpc_hip_permute_dev(const double *d_in, double *d_out, long long B, int N, const int32_t *d_map, int device, void *stream);

/* Error accounting, replaces bp_simulation.cpp:731-759,805-810 for the all-zero codeword.
 *   d_frame_info [B] int32 out (or NULL): number of wrong information bits (index >= R) of the frame, with bit 30
 *                set when the frame has any wrong bit at all (so 0 == frame correct)
 *   d_counters [5] uint64 in/out (accumulated with atomics; caller zeroes): nse, nde, nue, frames, sum |iters| */
int ldpc_hip_count_errors_dev(ldpc_hip_ctx *ctx, const uint32_t *d_hard, const int32_t *d_iters, long long B,
                              int32_t *d_frame_info, unsigned long long *d_counters, void *stream);

/* One fused Monte-Carlo batch on the device: noise -> decode -> count, frames [first_frame, first_frame+B).
 * Uses internal workspace sized for `B` (grown on demand).  counters[4] (host) receive this batch's
 * {nse, nde, nue, frames}; sum_abs_iters (host, may be NULL) the sum of |iters| for throughput accounting.
 * Synchronous.  No stopping rule here: the sequential stopping rule of bp_simulation.cpp:591,820 is applied
 * by the host layer on the ordered per-frame records (ldpc::bp_simulation). */
int ldpc_hip_simulate(ldpc_hip_ctx *ctx, double snr_db, int modulation_type, int punctured_blocks, int maxiter,
                      double alpha, uint64_t seed, long long first_frame, long long B,
                      unsigned long long counters[4], unsigned long long *sum_abs_iters);

/* Timing aid for bench.py: average duration in milliseconds of the decode kernel launches recorded with
 * HIP events on their own stream since the last reset (events are only recorded while enabled). */
int ldpc_hip_profile_enable(ldpc_hip_ctx *ctx, int enable);
int ldpc_hip_profile_read(ldpc_hip_ctx *ctx, double *total_ms, long long *launches, int reset);

#ifdef __cplusplus
}
#endif
#endif /* LDPC_HIP_H */
