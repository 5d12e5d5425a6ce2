/*
 * harness_oracle.h -- sequential CPU restatement of the reference Monte-Carlo harness
 * bp_simulation() (bp_simulation.cpp:305-841) for binary codes (q_mod == 2).
 *
 * TEST INFRASTRUCTURE ONLY (see ldpc_oracle.h).  PARITY UNPINNED at harness level: the reference's
 * bp_simulation.cpp cannot be built on this image (it needs commons_portable.cpp, which includes
 * <stropts.h>), and the reference has no test or fixture for it.  The decoders it calls ARE pinned
 * (oracle/_ref); the frame loop, RNG draw order, counters and stopping rule below follow the
 * reference text line by line and are cross-checked against the FER figures the survey measured
 * with the compiled reference.
 */
#ifndef HARNESS_ORACLE_H
#define HARNESS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    double ber;            /* nse / experiment / (n - r)      bp_simulation.cpp:840 */
    double fer;            /* nde / experiment                 bp_simulation.cpp:840 */
    long long nse;         /* information-bit errors in errored frames   :807 */
    long long nde;         /* errored frames                             :808 */
    long long nue;         /* errored frames whose decoder reported convergence (iter >= 0)  :809-810 */
    long long experiment;  /* frames simulated */
    long long sum_abs_iter;/* sum over frames of |decoder return value| (for throughput accounting) */
    unsigned int rng_next; /* next raw mt19937 output after the run (state fingerprint) */
} orc_sim_result;

/* H: row-major rh x nh ints, -1 empty.  decoder_type: DEC_ID (decoders.h:16-28) 1 SP, 3 MS, 4 IMS, 8 LMS.
 * modulation_type: 0 BPSK ("SKIP"), 1 QAM4 (bp_simulation.cpp:598-612).  seed: initial_random_seed after
 * reset_random() (commons_portable.cpp:144-158).  iters_out (optional, n_experiments+1 entries): per-frame
 * decoder return values.  Returns 0 on success, <0 on bad arguments. */
int orc_bp_simulation(int rh, int nh, const int *H, int M, int max_iterations, int n_frame_errors,
                      int n_experiments, double snr, double reference_frame_error, int decoder_type,
                      int modulation_type, int punctured_blocks, unsigned int seed,
                      orc_sim_result *out, int *iters_out);

/* Same with an interleaver between channel and decoder (bp_simulation.cpp:684): y[i] = buffer[inverse_map[i]];
 * inverse_map = NULL is the identity (permutation_type 0).  The map itself comes from the compiled upstream code
 * (tests/golden/interleavers.npz) or from the product's builder, which is checked against it. */
int orc_bp_simulation_perm(int rh, int nh, const int *H, int M, int max_iterations, int n_frame_errors,
                           int n_experiments, double snr, double reference_frame_error, int decoder_type,
                           int modulation_type, int punctured_blocks, unsigned int seed, const int *inverse_map,
                           orc_sim_result *out, int *iters_out);

/* The reference's RNG contract (commons_portable.cpp:138-178) exposed for tests: seeds a private
 * std::mt19937, optionally burns n_int01 draws of next_random_int(0,2), then writes n Gaussians. */
void orc_rng_gaussians(unsigned int seed, int n_int01, double *out, int n);

/* AWGN/BPSK LLRs exactly as bp_simulation.cpp:445,600-605 for the all-zero codeword:
 * llr[i] = -2*(sigma*g_i - 1)/(sigma*sigma), sigma = sqrt(10^(-snr/10)/2/rate). */
void orc_awgn_llr(unsigned int seed, int n_int01, double snr, double rate, double *llr, int n);

#ifdef __cplusplus
}
#endif
#endif
