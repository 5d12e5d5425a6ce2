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

#define LDPC_HIP_ABI_VERSION 4

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
/* Where ldpc_hip_open gets the code-specialised kernel of a base matrix that has no ahead-of-time instance (process-wide default; the
 * environment variable LDPC_HIP_JIT = 0 / sync / async overrides it): 0 never (table-driven or shape-unlimited kernels only),
 * 1 hiprtc inside ldpc_hip_open (seconds per new code, milliseconds from the on-disk cache) [default], 2 in the background: the
 * context starts on its table-driven / shape-unlimited kernel and moves to the instance at the first launch after it is ready --
 * every tier returns identical bits, so a run may change tier in flight.  Mode 2 is what a code search wants (upstream's
 * main_good_code_search.cpp:320 calls bp_simulation once per candidate matrix); ldpc::bp_simulation_t selects it.
 * Returns the previous mode, or LDPC_HIP_EINVAL. */
int ldpc_hip_set_jit_mode(int mode);
/* The same choice for the CALLING THREAD only (-1 = no override; the environment variable still wins).  Returns the thread's previous
 * override.  The C++ harnesses use this around their ldpc_hip_open_multi so that concurrent callers do not see each other's mode. */
int ldpc_hip_set_jit_mode_thread(int mode);
/* Name of the kernel the last ldpc_hip_decode_dev call on this context launched.  It differs from ldpc_hip_kernel_name only
 * for IMS_DEC with parameters beyond int8 (MS_DBITS > 8, MS_QBITS > 8 or alpha > 1), which run on the table-driven int32 kernel. */
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
 * MS, LMS, IMS (no transcendental on their path) and for SP, ASP and TASP against an upstream built with glibc >= 2.28 on an
 * FMA-capable x86-64 host (fp64 in upstream's operation order, no FMA contraction; exp() is evaluated with that glibc's own
 * algorithm and evaluation order -- an upstream linked against another libm, e.g. MSVC's from the vs2005/vs2010 projects, may
 * differ from it, and so from this library, in the last ulp of soft values; hard decisions and return values are expected equal);
 * BP: likewise bit for bit (its exp / log are glibc's algorithms on the device too).
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

/* ---- the transmit / receive chain around the decoder ------------------------------------------------------------------
 * Upstream's frame loop sends codeword -> direct interleaver -> mapper -> AWGN -> soft demapper -> inverse interleaver ->
 * puncturing -> decoder and counts decword[i] != codeword[i] (bp_simulation.cpp:566-577, 596-710, 731-759).  A context starts in
 * upstream's shipped wiring -- the all-zero codeword (bp_simulation.cpp:568 overwrites the encoder's output) and
 * permutation_type 0 (files/default_constants.jsonx:6) -- and the two setters below change that for every later channel /
 * count / simulate call on the context. */

/* Interleaver of the chain: Permutations_Open / Permutation_Init (direct_inverse_perm.cpp:139-783) with permutation_type 0..4,
 * permutation_block, permutation_inter of bp_simulation.h:21-23; halfmlog follows the modulation of each call
 * (bp_simulation.cpp:402-411).  Fails (and leaves the identity) when the mode does not accept this code shape. */
int ldpc_hip_set_interleaver(ldpc_hip_ctx *ctx, int permutation_type, int permutation_block, int permutation_inter);
/* Transmitted codewords: HOST array [ncw][N] of 0/1 bytes in decoder order (e.g. from ldpc_hip_encode_host); global frame f
 * carries codeword f % ncw.  ncw == 0 returns to the all-zero codeword.  (Replaces the `codeword` vector of
 * bp_simulation.cpp:506-568; upstream draws ONE random codeword per call and then zeroes it.) */
int ldpc_hip_set_codewords(ldpc_hip_ctx *ctx, const uint8_t *codewords, int ncw);

/* The chain from codeword to decoder input for frames [first_frame, first_frame + B), on the device:
 *   modulation_type 0 BPSK ("SKIP"), 1 QAM4: llr = -2*(sigma*g + 2*bit - 1)/sigma^2 (bp_simulation.cpp:603,610, sigma :445 / :449);
 *   2 / 3 / 4 = QAM16 / 64 / 256 (modulation.h:4-11): GrayPAM mapper (QAM_modulator.cpp:129-194), x + sigmaQAM*g fresh per frame
 *   (the evidently intended chain), per-rail soft demapper with cut-off T (QAM_demodulator.cpp:203-561),
 *   negated (:627-628); the last symbol is padded with zero bits (:575).
 * Then y[i] = buffer[inverse[i]] (:684) and the last M*punctured_blocks LLRs are set to 0.5 (LLR-type decoders) or 0
 * (probability-type) exactly as upstream (sic, Q7; :697-710) -- for EVERY modulation, with the punctured bitrate in sigma (:444).
 * g ~ N(0,1) from a counter-based Philox4x32-10 stream keyed by (seed, global frame index, channel position), so the result
 * does not depend on batch split, GPU count, interleaver or codeword. */
int ldpc_hip_channel_llr_dev(ldpc_hip_ctx *ctx, double snr_db, int modulation_type, int punctured_blocks, double T, uint64_t seed,
                             long long first_frame, long long B, double *d_llr, void *stream);
/* Earlier names of the same chain: BPSK / QAM4 (T unused), and QAM16+ without puncturing. */
int ldpc_hip_awgn_llr_dev(ldpc_hip_ctx *ctx, double snr_db, int modulation_type, int punctured_blocks,
                          uint64_t seed, long long first_frame, long long B, double *d_llr, void *stream);
int ldpc_hip_awgn_qam16_llr_dev(ldpc_hip_ctx *ctx, double snr_db, double T, uint64_t seed, long long first_frame,
                                long long B, double *d_llr, void *stream);
int ldpc_hip_awgn_qam_llr_dev(ldpc_hip_ctx *ctx, int modulation_type, double snr_db, double T, uint64_t seed, long long first_frame,
                              long long B, double *d_llr, void *stream);

/* Function-level mapper: QAM_modulator.cpp:142-194 QAM_modulator() for Q in {4,16,64,256}.  d_bits [ns][log2 Q] bytes 0/1 (first half
 * of a symbol's bits = I rail, MSB first; upstream passes them as doubles) -> d_x [ns][2] (I, Q interleaved) PAM levels. */
int ldpc_hip_qam_modulate_dev(int Q, const uint8_t *d_bits, long long ns, double *d_x, int device, void *stream);

/* Function-level soft demapper: QAM_demodulator.cpp:99-566 Demodulate() for Q in {4,16,64,256}, out_type 0/1 (Q = 4: out_type 0 only).
 * d_x [ns][2] (I,Q interleaved) -> d_out [ns][log2 Q]. */
int ldpc_hip_qam_demod_dev(int Q, double T, double sigma, const double *d_x, long long ns, double *d_out,
                           int out_type, int device, void *stream);

/* Systematic encoder for the dual-diagonal QC-LDPC codes upstream's search produces (qc_encode / random_codeword,
 * bp_simulation.cpp:22-191, with the information bits given instead of drawn).  HOST function.  info_bits: (nh-rh)*M bytes
 * (0/1) for variable positions [rh*M, nh*M); codeword: nh*M bytes out, parity first.  Upstream's simulation never transmits
 * the result (it zeroes the codeword, :568); this entry exists for callers that do and for the sign-symmetry tests. */
int ldpc_hip_encode_host(int rh, int nh, int M, const int16_t *hd, const uint8_t *info_bits, uint8_t *codeword);

/* The same encoder on the device (csrc/ldpc_encode.hpp): d_info_bits [B][(nh-rh)*M] bytes 0/1 -> d_codewords [B][nh*M] bytes, parity
 * first, identical to ldpc_hip_encode_host -- also for base matrices whose parity part consists of several dual-diagonal blocks
 * (bp_simulation.cpp:142-191: encoded from the last block to the first). */
int ldpc_hip_encode_dev(ldpc_hip_ctx *ctx, const uint8_t *d_info_bits, long long B, uint8_t *d_codewords, void *stream);
/* A table of ncw random codewords made on the device (information bits from Philox4x32-10 keyed by seed and codeword index, then the
 * device encoder) and installed like ldpc_hip_set_codewords: global frame f carries codeword f % ncw.  ncw == 0 returns to the all-zero
 * codeword.  (What upstream's random_codeword() is for, bp_simulation.cpp:142-191, before :568 zeroes its result.) */
int ldpc_hip_set_random_codewords(ldpc_hip_ctx *ctx, uint64_t seed, int ncw);

/* Bit interleavers of the simulation chain (Permutations_Open / Permutation_Init / Permutation,
 * direct_inverse_perm.cpp:139-900; permutation_type of bp_simulation.h:21-23): mode 0 identity, 1 random, 2 deterministic,
 * 3 block random (block_size), 4 interleaved random (step_size); halfmlog = 1 (BPSK / QAM4), 2, 3, 4 (QAM16 / 64 / 256).
 * HOST function (no GPU needed): fills the two gather maps, out[i] = in[map[i]], direct[N] (encoder -> mapper side) and
 * inverse[N] (demapper -> decoder side), N = c*M, identical to upstream's for the same arguments (same LCG, same order). */
int ldpc_hip_interleaver_build(int b, int c, int M, int halfmlog, int mode, int block_size, int step_size, const int16_t *hd,
                               int32_t *direct, int32_t *inverse);
/* Applies a map to B frames resident on the device: d_out[f][i] = d_in[f][d_map[i]] (d_in != d_out). */
int ldpc_hip_permute_dev(const double *d_in, double *d_out, long long B, int N, const int32_t *d_map, int device, void *stream);

/* Error accounting, replaces bp_simulation.cpp:731-759,805-810: hard decisions against the transmitted codeword of each frame
 * (global frame first_frame + f carried codeword (first_frame + f) % ncw; all-zero when none is set).
 *   d_frame_info [B] int32 out (or NULL): number of wrong information bits (index >= R) of the frame, with bit 30
 *                set when the frame has any wrong bit at all (so 0 == frame correct)
 *   d_counters [5] uint64 in/out (accumulated with atomics; caller zeroes): nse, nde, nue, frames, sum |iters| */
int ldpc_hip_count_errors_cw_dev(ldpc_hip_ctx *ctx, const uint32_t *d_hard, const int32_t *d_iters, long long first_frame, long long B,
                                 int32_t *d_frame_info, unsigned long long *d_counters, void *stream);
/* Same with first_frame = 0 (enough for the all-zero codeword or a single codeword; EINVAL when several are set). */
int ldpc_hip_count_errors_dev(ldpc_hip_ctx *ctx, const uint32_t *d_hard, const int32_t *d_iters, long long B,
                              int32_t *d_frame_info, unsigned long long *d_counters, void *stream);

/* One fused Monte-Carlo batch on the device: chain -> decode -> count, frames [first_frame, first_frame+B), with the
 * context's interleaver and codewords.  Uses internal workspace (grown on demand).  counters[4] (host) receive this batch's
 * {nse, nde, nue, frames}; sum_abs_iters (host, may be NULL) the sum of |iters| for throughput accounting.
 * Synchronous.  No stopping rule here: the sequential stopping rule of bp_simulation.cpp:591,820 is applied
 * by the host layer on the ordered per-frame records (ldpc::bp_simulation). */
int ldpc_hip_simulate(ldpc_hip_ctx *ctx, double snr_db, int modulation_type, int punctured_blocks, int maxiter,
                      double alpha, uint64_t seed, long long first_frame, long long B,
                      unsigned long long counters[4], unsigned long long *sum_abs_iters);

/* ---- upstream's own noise, bit for bit, on the device (exact replay at device speed) --------------------------------------
 * bp_simulation() draws every noise sample from ONE std::mt19937 through next_random_gaussian(), a fresh
 * std::normal_distribution<double> per call (commons_portable.cpp:140,174-178; bp_simulation.cpp:600-611).  These entry points
 * continue THAT stream on the GPU: the context holds a generator state in std::mt19937's own terms -- the 624 state words and
 * the index of the next word, exactly what `os << generator` prints with libstdc++ -- and every call below consumes from it
 * precisely the 32-bit words the host loop would (four per polar-method attempt, libstdc++ bits/random.tcc:1802-1835,3348-3380).
 * Values equal the host's bit for bit on a host whose libm log() is glibc >= 2.28's FMA variant (the contract of the decoders'
 * exp / log above).  ldpc::bp_simulation_t (include/ldpc/bp_simulation.h) and the drop-in bp_simulation symbol use them, so the
 * exact-replay harness is no longer bound by the host generator (~7e3 frames/s per core at N = 2048). */
/* host only, no GPU: the state 2^log2_words words further on (18 <= log2_words <= 30), as the device computes it: GF(2)
 * polynomial jump x^(2^k) mod the generator's minimal polynomial.  state_out[1..623] are the generator's words; of state_out[0]
 * only bit 31 is state (the recurrence never reads the rest). */
int ldpc_hip_mt_jump_host(const uint32_t state_in[624], int log2_words, uint32_t state_out[624]);
/* Load / read the generator: state[624] + pos (0..624) as libstdc++ streams a std::mt19937.  set also restarts the context's frame
 * count (which of the ldpc_hip_set_codewords codewords a frame carries: frame f -> codeword f % ncw). */
int ldpc_hip_mt_set_state(ldpc_hip_ctx *ctx, const uint32_t state[624], int pos);
int ldpc_hip_mt_get_state(ldpc_hip_ctx *ctx, uint32_t state[624], int *pos);
/* The frame count that belongs to a snapshot: a caller that rolls the generator back with ldpc_hip_mt_set_state (the harness after an
 * early stop, bp_simulation.cpp:820) and keeps drawing restores it too, so that frame f of the run still carries codeword f % ncw. */
long long ldpc_hip_mt_get_frame_index(const ldpc_hip_ctx *ctx);
int ldpc_hip_mt_set_frame_index(ldpc_hip_ctx *ctx, long long frames_taken);
/* The next `count` values of next_random_gaussian() (commons_portable.cpp:174-178) to d_out [count] (device; NULL = draw and drop). */
int ldpc_hip_mt_normal_dev(ldpc_hip_ctx *ctx, long long count, double *d_out, void *stream);
/* Same into a HOST array (convenience; the harness uses it to check once per process that the device stream reproduces THIS host's
 * std::normal_distribution before relying on it). */
int ldpc_hip_mt_normal_host(ldpc_hip_ctx *ctx, long long count, double *out);
/* Decoder input of the next B frames exactly as the frame loop builds it: y = -2*(sigma*g + 2*c - 1)/sigma^2 with g drawn in index
 * order (bp_simulation.cpp:600-611, sigma :445 / :449; c = 0 unless ldpc_hip_set_codewords), inverse interleaver (:684,
 * ldpc_hip_set_interleaver), puncturing (:697-710).  modulation_type 0 or 1.  d_llr [B][N] device, NULL = draw and drop (used to put
 * the generator where a frame-by-frame loop that stopped early would have left it).  Synchronises `stream`. */
int ldpc_hip_mt_llr_dev(ldpc_hip_ctx *ctx, double snr_db, int modulation_type, int punctured_blocks, long long B, double *d_llr, void *stream);
/* ldpc_hip_mt_llr_dev -> decode -> count for the next B frames; HOST arrays frame_info[B], iters[B] as in ldpc_hip_frames_multi.
 * Synchronous.  The stopping rule (bp_simulation.cpp:591,820) is the caller's, on the ordered records. */
int ldpc_hip_mt_frames(ldpc_hip_ctx *ctx, double snr_db, int modulation_type, int punctured_blocks, int maxiter, double alpha, long long B,
                       int32_t *frame_info, int32_t *iters);
/* One rank's share when every rank (process or shard) runs the same generator: the stream advances by all B frames, frames
 * [lo, hi) are decoded and counted here; frame_info / iters have hi - lo entries. */
int ldpc_hip_mt_frames_slice(ldpc_hip_ctx *ctx, double snr_db, int modulation_type, int punctured_blocks, int maxiter, double alpha, long long B,
                             long long lo, long long hi, int32_t *frame_info, int32_t *iters);

/* The same stream with ONE PROCESS PER GPU: every rank holds a context whose generator is in the same state, and the ranks share the
 * generator's tape out among themselves exactly as ldpc_hip_mt_frames_multi does over the shards of one process -- the three small
 * exchanges are the caller's (torch.distributed, MPI, ...).  One round covers the next `frames` frames (<= 65536), rank r decodes
 * frames [frames*r/n, frames*(r+1)/n):
 *   1. ldpc_hip_mt_shard_begin   makes the sub-streams around this rank's frames and counts the accepted polar-method attempts of
 *                                its stretch -> own_count;                                  all-gather the n counts
 *   2. ldpc_hip_mt_shard_emit    counts[n] in: emits this rank's decoder inputs; out: frames_done (whole frames the round completes, the
 *                                same on every rank), found (this rank holds the state the round ends in -> state_next), covered (0: an
 *                                item of this rank lay outside its window);               all-gather (found, covered); broadcast
 *                                state_next from the lowest rank with found = 1
 *   3. ldpc_hip_mt_shard_commit  installs that state and decodes this rank's frames below frames_done: frame_info / iters receive
 *                                min(frames*(r+1)/n, frames_done) - frames*r/n records.
 * If any rank reports covered = 0 or none reports found = 1 -- an estimate failed; probability ~1e-15 per round -- every rank calls
 * ldpc_hip_mt_shard_abandon and runs the round with ldpc_hip_mt_frames_slice instead (the whole tape on every rank): nothing has
 * been changed before commit.  (ldpc_lib_amd.bp_simulation(exact_seed=...) under torch.distributed does all of this.) */
int ldpc_hip_mt_shard_begin(ldpc_hip_ctx *ctx, double snr_db, int modulation_type, int punctured_blocks, long long frames, int rank, int n,
                            unsigned long long *own_count);
int ldpc_hip_mt_shard_emit(ldpc_hip_ctx *ctx, const unsigned long long *counts, int *found, int *covered, uint32_t state_next[624],
                           long long *frames_done);
int ldpc_hip_mt_shard_commit(ldpc_hip_ctx *ctx, const uint32_t state[624], long long frames_done, int maxiter, double alpha,
                             int32_t *frame_info, int32_t *iters);
void ldpc_hip_mt_shard_abandon(ldpc_hip_ctx *ctx);

/* ---- several GPUs of one node (bp_simulation's frame loop sharded; north_star: RCCL all-reduce for the counters only) ----
 * One shard = one context + one HIP stream + one host thread.  devices[i] is the HIP ordinal of shard i; ordinals may repeat
 * (logical shards on one GPU: results are identical, the counters are then summed on the host because RCCL does not accept one
 * device twice in a communicator).  With distinct devices the five counters are all-reduced over RCCL (ncclAllReduce, uint64 sum,
 * one 40-byte message per call); RCCL is dlopen()ed, LDPC_HIP_RCCL_PATH overrides the library.  Frames [first_frame,
 * first_frame + B) are cut into consecutive batches of `batch` frames, batch k goes to shard k mod n; noise is keyed by the global
 * frame index, so counters and records do not depend on n or batch. */
typedef struct ldpc_hip_multi ldpc_hip_multi;
int ldpc_hip_open_multi(int decoder_id, int rh, int nh, int M, const int16_t *hd, const int *devices, int n_shards, ldpc_hip_multi **out);
void ldpc_hip_close_multi(ldpc_hip_multi *m);
int ldpc_hip_multi_shards(const ldpc_hip_multi *m);
ldpc_hip_ctx *ldpc_hip_multi_ctx(ldpc_hip_multi *m, int shard);      /* per-shard settings (ims params, bp chain, profiling) */
void *ldpc_hip_multi_stream(ldpc_hip_multi *m, int shard);          /* the shard's hipStream_t (work of the *_multi calls runs on it) */
const char *ldpc_hip_multi_reduction(const ldpc_hip_multi *m);      /* "rccl" or "host" */
/* Communicators are made once per device list and process (ncclCommInitAll is far more expensive than a short bp_simulation call,
 * and a code search opens a multi context per candidate matrix: main_good_code_search.cpp:320); ldpc_hip_close_multi hands them back to
 * the cache.  comm_inits = ncclCommInitAll calls so far; release_comms destroys the sets no open multi context holds. */
long long ldpc_hip_multi_comm_inits(void);
void ldpc_hip_multi_release_comms(void);
int ldpc_hip_multi_set_interleaver(ldpc_hip_multi *m, int permutation_type, int permutation_block, int permutation_inter);
int ldpc_hip_multi_set_codewords(ldpc_hip_multi *m, const uint8_t *codewords, int ncw);
int ldpc_hip_multi_set_random_codewords(ldpc_hip_multi *m, uint64_t seed, int ncw);
int ldpc_hip_simulate_multi(ldpc_hip_multi *m, double snr_db, int modulation_type, int punctured_blocks, int maxiter, double alpha,
                            uint64_t seed, long long first_frame, long long B, long long batch, unsigned long long counters[4],
                            unsigned long long *sum_abs_iters);
/* Same, and the ordered per-frame records for the sequential stopping rule (bp_simulation.cpp:591,820): HOST arrays
 * frame_info[B] (as d_frame_info above) and iters[B] in global frame order; counters / sum_abs_iters may be NULL. */
int ldpc_hip_frames_multi(ldpc_hip_multi *m, double snr_db, int modulation_type, int punctured_blocks, int maxiter, double alpha,
                          uint64_t seed, long long first_frame, long long B, long long batch, int32_t *frame_info, int32_t *iters,
                          unsigned long long counters[4], unsigned long long *sum_abs_iters);
/* The hot path on batches already RESIDENT on the GPUs (what bench.py times at N > 1): shard i decodes d_llr[i] ([B_per_shard][N]
 * float64 on devices[i]) on its own stream and counts the errors -- its frames are global frames first_frame + i*B_per_shard + f for the
 * codeword choice --, then one all-reduce of the five counters.  d_llr: HOST array of n device pointers.  Synchronous.  If any shard
 * fails before the all-reduce, no shard enters it and the call returns that shard's error (bp_simulation.cpp:716-759 per frame). */
int ldpc_hip_decode_count_multi(ldpc_hip_multi *m, const double *const *d_llr, long long B_per_shard, long long first_frame, int maxiter,
                                double alpha, unsigned long long counters[4], unsigned long long *sum_abs_iters);
/* ldpc_hip_decode_host over the shards: contiguous slices of the batch, one host thread per shard (BP_DEC with the frame chain
 * on is sequential by definition and runs on shard 0). */
int ldpc_hip_decode_host_multi(ldpc_hip_multi *m, double *llr, long long B, int maxiter, int decision, double alpha, double *decword,
                               int32_t *iters, int clobber_sp_input);
/* The exact-replay stream (ldpc_hip_mt_*) over the shards.  The generator's stream is sequential, but jump-ahead reaches any point of
 * it: the round's word tape is cut at the expected positions of the shards' first items, every shard makes only the sub-streams around
 * its own stretch (+- 8 standard deviations) and counts the accepted polar-method attempts in it, the n counts go through the host (a
 * counters-only exchange), and every shard then emits and decodes its contiguous share of each round's frames; the records come back
 * in frame order and all shards are left in the state the sequential loop would be in.  Per-shard generation work is ~1/n; if a
 * margin is ever exceeded the round is redone with the whole tape on every shard (ldpc_hip_multi_mt_stats counts both kinds), so
 * results never depend on the estimates.  LDPC_HIP_MT_SHARDED=0 forces the whole-tape mode, =1 shards even short rounds.
 * advance = draw and drop B frames on every shard (roll-forward after an early stop). */
int ldpc_hip_mt_set_state_multi(ldpc_hip_multi *m, const uint32_t state[624], int pos);
int ldpc_hip_mt_get_state_multi(ldpc_hip_multi *m, uint32_t state[624], int *pos);
int ldpc_hip_mt_set_frame_index_multi(ldpc_hip_multi *m, long long frames_taken);
int ldpc_hip_mt_advance_multi(ldpc_hip_multi *m, double snr_db, int modulation_type, int punctured_blocks, long long B);
int ldpc_hip_mt_frames_multi(ldpc_hip_multi *m, double snr_db, int modulation_type, int punctured_blocks, int maxiter, double alpha,
                             long long B, int32_t *frame_info, int32_t *iters);
void ldpc_hip_multi_mt_stats(const ldpc_hip_multi *m, long long *sharded_rounds, long long *fallback_rounds);

/* Timing aid for bench.py: average duration in milliseconds of the decode kernel launches recorded with
 * HIP events on their own stream since the last reset (events are only recorded while enabled). */
int ldpc_hip_profile_enable(ldpc_hip_ctx *ctx, int enable);
int ldpc_hip_profile_read(ldpc_hip_ctx *ctx, double *total_ms, long long *launches, int reset);

#ifdef __cplusplus
}
#endif
#endif /* LDPC_HIP_H */
