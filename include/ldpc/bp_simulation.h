/*
 * ldpc/bp_simulation.h -- the Monte-Carlo harness of upstream's bp_simulation.h:9-27 / bp_simulation.cpp:305-841
 * on the MI355X batch decoder, in EXACT-REPLAY mode: the channel noise is the stream of the same std::mt19937 in the same
 * draw order as upstream (commons_portable.cpp:138-178, bp_simulation.cpp:512,600-611) -- continued ON THE GPU by default
 * (ldpc_hip_mt_*: the generator's words, libstdc++'s polar method and glibc's log reproduced bit for bit, so only the 8-byte frame
 * records cross PCIe; LDPC_HIP_EXACT_NOISE=host draws on the host instead) -- frames are decoded in GPU batches, and upstream's
 * sequential stopping rule (bp_simulation.cpp:591,820) is replayed over the ordered per-frame results.  On an early stop the
 * generator is rolled back and re-advanced so that it is left exactly where upstream's frame-by-frame loop leaves it (later
 * callers draw from it: main_good_code_search.cpp:316).
 * Result, counters and generator state are identical to upstream's for q_mod == 2, modulation SKIP/QAM4,
 * permutation_type 0 (the only mode the shipped configurations use, files/default_constants.jsonx:6).
 *
 *   ldpc::bp_simulation_throughput_t() the same harness in THROUGHPUT mode: channel noise drawn on the GPUs (counter-based Philox
 *                                      keyed by the global frame index), frames sharded over `devices` behind the C-ABI
 *                                      (ldpc_hip_open_multi: one host thread + stream per GPU, RCCL all-reduce of the counters),
 *                                      upstream's sequential stopping rule replayed over the ordered per-frame records;
 *   LDPC_HIP_DEVICES="0,1,2,3" | "all" selects the GPUs for every entry point below that takes no explicit list.
 *   ldpc::bp_simulation_t<Mat, Env>()  generic over the matrix type (needs n_rows(), n_cols(), operator()(i,j))
 *                                      and over the environment that owns the generator (see RngEnv below);
 *   ldpc::bp_simulation()              standalone: ldpc::Matrix + the library's own generator (ldpc::reset_random).
 * The drop-in definition of upstream's exact `bp_simulation(int, matrix<int> const&, matrix<int>&, ...)` symbol is
 * csrc/compat/bp_simulation_dropin.cpp (compiled inside the upstream tree).
 */
#ifndef LDPC_COMPAT_BP_SIMULATION_H_
#define LDPC_COMPAT_BP_SIMULATION_H_

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "ldpc_hip.h"
#include "ldpc/interleaver.h"

namespace ldpc {

enum { MODULATION_SKIP_ = 0, MODULATION_QAM4_ = 1 };  // modulation.h:4-11
enum { DEC_DECISION_HARD = 0 };                        // decoders.h:7 DEC_DECISION

struct Matrix {  // minimal stand-in for upstream's matrix<int> (data_structures.h:11-57)
    std::vector<int> v;
    int rows = 0, cols = 0;
    Matrix() {}
    Matrix(int r, int c, int init = 0) : v((size_t)r * c, init), rows(r), cols(c) {}
    int n_rows() const { return rows; }
    int n_cols() const { return cols; }
    int &operator()(int r, int c) { return v[(size_t)r * cols + c]; }
    int const &operator()(int r, int c) const { return v[(size_t)r * cols + c]; }
};

// The library's own generator with upstream's contract (commons_portable.cpp:138-178): one global mt19937, seeded
// from initial_random_seed at the first draw after reset_random(); a FRESH distribution object per call.
extern int initial_random_seed;
void reset_random();
std::mt19937 &random_generator();  // seeded on demand
int next_random_int(int min_inclusive, int max_exclusive);
double next_random_gaussian();

// GPUs to shard over: LDPC_HIP_DEVICES = "all" or a comma separated list of HIP ordinals; otherwise {fallback}.
inline std::vector<int> devices_from_env(int fallback = 0) {
    std::vector<int> d;
    if (const char *e = getenv("LDPC_HIP_DEVICES")) {
        const std::string v(e);
        if (v == "all") {
            for (int i = 0, n = ldpc_hip_device_count(); i < n; ++i) d.push_back(i);
        } else {
            size_t pos = 0;
            while (pos < v.size()) {
                size_t end = v.find(',', pos);
                if (end == std::string::npos) end = v.size();
                if (end > pos) d.push_back(atoi(v.substr(pos, end - pos).c_str()));
                pos = end + 1;
            }
        }
    }
    if (d.empty()) d.push_back(fallback);
    return d;
}

struct OwnRngEnv {
    static std::mt19937 &generator() { return random_generator(); }
    static double gaussian() { return next_random_gaussian(); }
    // bp_simulation.cpp:512 -> random_codeword() :142-191 draws (nh-rh)*M values of next_random_int(0,2) (:160-162);
    // the codeword itself is overwritten with zeros afterwards (:568), so only the draws matter here.
    template <class Mat>
    static void burn_codeword_draws(Mat const &H, int M) {
        for (long long i = 0, n = (long long)(H.n_cols() - H.n_rows()) * M; i < n; ++i) (void)next_random_int(0, 2);
    }
    [[noreturn]] static void fail(const char *msg) { fprintf(stderr, "%s\n", msg); exit(1); }  // die()
};

// std::mt19937 <-> (624 state words, index of the next word): libstdc++ streams exactly that (bits/random.tcc operator<<).  The
// export is checked against the generator's own next output; any other standard library fails the check and the harness then
// keeps the noise on the host.
inline unsigned mt_word_after(const uint32_t *w, int pos) {
    unsigned y;
    if (pos < 624) y = w[pos];
    else {   // the block is used up: the next word is x[624] of the sequence that starts with these 624 words
        const unsigned t = (w[0] & 0x80000000u) | (w[1] & 0x7fffffffu);
        y = w[397] ^ (t >> 1) ^ ((t & 1u) ? 0x9908b0dfu : 0u);
    }
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
inline bool mt_export(const std::mt19937 &g, uint32_t w[624], int &pos) {
    std::ostringstream os;
    os << g;
    std::istringstream is(os.str());
    for (int i = 0; i < 624; ++i) { unsigned long v = 0; is >> v; w[i] = (uint32_t)v; }
    long p = -1;
    is >> p;
    if (is.fail() || p < 0 || p > 624) return false;
    pos = (int)p;
    std::mt19937 probe = g;
    return (unsigned)probe() == mt_word_after(w, pos);
}
inline bool mt_import(std::mt19937 &g, const uint32_t w[624], int pos) {
    std::ostringstream os;
    for (int i = 0; i < 624; ++i) os << w[i] << ' ';
    os << pos;
    std::istringstream is(os.str());
    std::mt19937 t;
    is >> t;
    if (is.fail()) return false;
    std::mt19937 probe = t;
    if ((unsigned)probe() != mt_word_after(w, pos)) return false;
    g = t;
    return true;
}

// Once per process: does the device-side stream reproduce THIS host's std::mt19937 + std::normal_distribution (libstdc++'s polar
// method, this libm's log / sqrt)?  4096 samples from a scratch generator, compared bit for bit.  If not -- another standard
// library, a libm whose log differs in the last place -- the harness keeps drawing on the host, so exact replay stays exact.
inline bool device_stream_matches_host(ldpc_hip_ctx *ctx) {
    static std::atomic<int> verdict(-1);   // concurrent first calls both run the check and store the same answer
    if (verdict.load() >= 0) return verdict.load() == 1;
    std::mt19937 probe(20240611u);
    uint32_t w[624];
    int pos = 0;
    bool same = mt_export(probe, w, pos) && ldpc_hip_mt_set_state(ctx, w, pos) == 0;
    std::vector<double> dev(4096);
    same = same && ldpc_hip_mt_normal_host(ctx, (long long)dev.size(), dev.data()) == 0;
    for (size_t i = 0; same && i < dev.size(); ++i) {
        std::normal_distribution<double> dist;   // a fresh one per sample, as upstream (commons_portable.cpp:174-178)
        const double h = dist(probe);
        same = std::memcmp(&h, &dev[i], sizeof h) == 0;
    }
    verdict.store(same ? 1 : 0);
    return same;
}

struct SimCounters { long long nse = 0, nde = 0, nue = 0, experiment = 0, sum_abs_iters = 0; };

template <class Mat, class Env>
std::pair<double, double> bp_simulation_t(int q_mod, Mat const &H, int tailbite_length, int max_iterations,
                                          int n_frame_errors, int n_experiments, double snr,
                                          double reference_frame_error, int decoder_type, int modulation_type,
                                          int permutation_type, int punctured_blocks, int show_process,
                                          SimCounters *counters_out, const std::vector<int> &devices, long long max_batch = 4096,
                                          int permutation_block = 128, int permutation_inter = 1) {
    if (q_mod != 2) Env::fail("bp_simulation: only binary codes (q_mod == 2) are built in ldpc-lib_amd");
    if (modulation_type != MODULATION_SKIP_ && modulation_type != MODULATION_QAM4_)
        Env::fail("bp_simulation: exact-replay mode supports MODULATION_SKIP and MODULATION_QAM4 (upstream's QAM16+ wiring "
                  "is broken, SURVEY Appendix B Q5/Q6; use the device-side chain ldpc_hip_awgn_qam16_llr_dev)");
    const int b = H.n_rows(), c = H.n_cols(), M = tailbite_length;
    const int r = b * M, n = c * M;

    std::vector<int16_t> hd((size_t)b * c);
    for (int i = 0; i < b; ++i) for (int j = 0; j < c; ++j) hd[(size_t)i * c + j] = (int16_t)H(i, j);  // :359-361
    ldpc_hip_multi *ctx = nullptr;   // one DEC_STATE per GPU; the batch of a round is cut into contiguous slices
    if (devices.empty()) Env::fail("bp_simulation: empty device list");
    // a code search calls this once per candidate matrix (main_good_code_search.cpp:320): never wait for hiprtc here, start on the
    // table-driven / shape-unlimited kernel and move to the code-specialised instance when the background compile delivers it
    const int jit_before = ldpc_hip_set_jit_mode_thread(2);   // this thread's opens only: concurrent callers keep their own mode
    const int open_rc = ldpc_hip_open_multi(decoder_type, b, c, M, hd.data(), devices.data(), (int)devices.size(), &ctx);  // :353-355
    (void)ldpc_hip_set_jit_mode_thread(jit_before);
    if (open_rc != 0) Env::fail(ldpc_hip_last_error());
    max_batch *= (long long)devices.size();

    int out_type;  // :451-466
    switch (decoder_type) {
    case LDPC_HIP_SP_DEC: case LDPC_HIP_ASP_DEC: case LDPC_HIP_TASP_DEC: out_type = 1; break;
    case LDPC_HIP_BP_DEC: case LDPC_HIP_MS_DEC: case LDPC_HIP_LMS_DEC: case LDPC_HIP_IMS_DEC: out_type = 0; break;
    default: out_type = 0; Env::fail("Unknown decoder type");
    }
    const int QAM = modulation_type == MODULATION_SKIP_ ? 1 : 4, halfmlog = 1;                  // :405-406
    const double bitrate = (double)(c - b) / (c - punctured_blocks);                             // :444
    const double sigma = std::sqrt(std::pow(10, -snr / 10) / 2 / bitrate);                       // :445
    const double norm_factor = 2.0 * (QAM - 1.0) / 3.0;                                          // :447
    const double sigmaQAM = std::sqrt(std::pow(10., -snr / 10.) / (2 * bitrate * halfmlog * 2) * norm_factor);  // :449
    const double sg = modulation_type == MODULATION_SKIP_ ? sigma : sigmaQAM;

    // :413-427 interleaver between demapper and decoder (identity for permutation_type 0); same maps as upstream's
    Interleaver il;
    {
        std::vector<int> hflat((size_t)b * c);
        for (int i = 0; i < b; ++i) for (int j = 0; j < c; ++j) hflat[(size_t)i * c + j] = H(i, j);
        std::string err;
        if (!build_interleaver(b, c, M, halfmlog, permutation_type, permutation_block, permutation_inter, hflat.data(), il, err))
            Env::fail(err.c_str());
    }
    std::vector<double> chan((size_t)n);

    Env::burn_codeword_draws(H, M);  // :512

    long long nse = 0, nue = 0, nde = 0, experiment = 0, sum_abs_iters = 0;
    std::vector<double> llr, decword;
    std::vector<int32_t> iters;
    long long batch = 64;
    bool stop = false;

    // Noise on the device (default): the GPUs continue the generator's own stream -- same words, same polar-method attempts, same
    // libm log -- so nothing but the 8-byte frame records crosses PCIe.  LDPC_HIP_EXACT_NOISE=host keeps the draws on the host.
    uint32_t mt_words[624];
    int mt_pos = 0;
    const char *noise_env = getenv("LDPC_HIP_EXACT_NOISE");
    bool device_noise = !(noise_env && std::string(noise_env) == "host") && device_stream_matches_host(ldpc_hip_multi_ctx(ctx, 0)) &&
                        mt_export(Env::generator(), mt_words, mt_pos);
    if (device_noise) {
        if (ldpc_hip_multi_set_interleaver(ctx, permutation_type, permutation_block, permutation_inter) != 0) Env::fail(ldpc_hip_last_error());
        if (ldpc_hip_mt_set_state_multi(ctx, mt_words, mt_pos) != 0) Env::fail(ldpc_hip_last_error());
        const long long cap = (max_batch > 65536 * (long long)devices.size()) ? max_batch : 65536 * (long long)devices.size();
        std::vector<int32_t> info;
        batch = 256;
        while (!stop && nde < n_frame_errors && experiment <= n_experiments) {                  // :591
            const long long room = (long long)n_experiments + 1 - experiment;
            const long long B = batch < room ? batch : room;
            info.resize((size_t)B); iters.resize((size_t)B);
            if (ldpc_hip_mt_get_state_multi(ctx, mt_words, &mt_pos) != 0) Env::fail(ldpc_hip_last_error());   // snapshot (+ the frame count: experiment)
            if (ldpc_hip_mt_frames_multi(ctx, snr, modulation_type, punctured_blocks, max_iterations, 0.8 /*MS_ALPHA*/, B, info.data(),
                                         iters.data()) != 0)
                Env::fail(ldpc_hip_last_error());
            long long used = 0;
            for (long long f = 0; f < B; ++f) {                                                 // ordered replay of :591-823
                if (!(nde < n_frame_errors && experiment <= n_experiments)) { stop = true; break; }
                ++experiment; ++used;
                const int iter = iters[(size_t)f];
                sum_abs_iters += iter < 0 ? -iter : iter;
                if (info[(size_t)f] != 0) {                                                      // bit 30: any wrong bit (:805)
                    nse += info[(size_t)f] & ((1 << 30) - 1); ++nde;
                    if (iter >= 0) ++nue;
                    if (show_process)
                        printf("SNR=%5.3lf,step=%4d,s_ers=%d,f_ers=%d,u_ers=%d,BER=%5.3le,FER=%5.3le\n", snr, (int)experiment,
                               (int)nse, (int)nde, (int)nue, (double)nse / experiment / (n - r), (double)nde / experiment);
                    if (nde >= 10 && (double)nde / experiment > 2.5 * reference_frame_error) { stop = true; break; }   // :820
                }
            }
            if (used < B) {  // stopped inside the batch: put the generator where the frame-by-frame loop leaves it
                if (ldpc_hip_mt_set_state_multi(ctx, mt_words, mt_pos) != 0) Env::fail(ldpc_hip_last_error());
                if (ldpc_hip_mt_set_frame_index_multi(ctx, experiment - used) != 0) Env::fail(ldpc_hip_last_error());   // set_state restarts the count
                if (ldpc_hip_mt_advance_multi(ctx, snr, modulation_type, punctured_blocks, used) != 0) Env::fail(ldpc_hip_last_error());
            }
            if (batch < cap) batch = batch * 4 < cap ? batch * 4 : cap;
        }
        if (ldpc_hip_mt_get_state_multi(ctx, mt_words, &mt_pos) != 0) Env::fail(ldpc_hip_last_error());
        if (!mt_import(Env::generator(), mt_words, mt_pos)) Env::fail("bp_simulation: could not hand the generator state back to std::mt19937");
        stop = true;   // the host-noise loop below is skipped
    }
    while (!device_noise && !stop && nde < n_frame_errors && experiment <= n_experiments) {     // :591
        const long long room = (long long)n_experiments + 1 - experiment;
        const long long B = batch < room ? batch : room;
        llr.resize((size_t)B * n); decword.resize((size_t)B * n); iters.resize((size_t)B);
        const std::mt19937 snapshot = Env::generator();
        for (long long f = 0; f < B; ++f) {
            double *y = llr.data() + (size_t)f * n;
            for (int i = 0; i < n; ++i) {                                                       // :601-611
                const double noise = Env::gaussian();
                chan[(size_t)i] = -2.0 * (sg * noise + 2.0 * 0.0 - 1.0) / (sg * sg);              // codeword == 0 (:568)
            }
            for (int i = 0; i < n; ++i) y[i] = chan[(size_t)il.inverse[(size_t)i]];              // :684 inverse permutation
            const double init_val = out_type == 1 ? 0 : 0.5;                                     // :700 (sic)
            const int plen = M * punctured_blocks, pstart = n - plen;
            for (int i = pstart; i < pstart + plen; ++i) y[i] = init_val;                        // :702-709
        }
        if (ldpc_hip_decode_host_multi(ctx, llr.data(), B, max_iterations, DEC_DECISION_HARD, 0.8 /*MS_ALPHA*/, decword.data(),
                                       iters.data(), 0) != 0)
            Env::fail(ldpc_hip_last_error());
        long long used = 0;
        for (long long f = 0; f < B; ++f) {                                                     // ordered replay of :591-823
            if (!(nde < n_frame_errors && experiment <= n_experiments)) { stop = true; break; }
            ++experiment; ++used;
            const int iter = iters[(size_t)f];
            sum_abs_iters += iter < 0 ? -iter : iter;
            const double *d = decword.data() + (size_t)f * n;
            int curr_nse = 0, curr_nse_info = 0;                                                 // :735-742
            for (int i = 0; i < n; ++i)
                if (d[i] != 0.0) { ++curr_nse; if (i >= r) ++curr_nse_info; }
            if (curr_nse > 0) {                                                                  // :805-823
                nse += curr_nse_info; ++nde;
                if (iter >= 0) ++nue;
                if (show_process)
                    printf("SNR=%5.3lf,step=%4d,s_ers=%d,f_ers=%d,u_ers=%d,BER=%5.3le,FER=%5.3le\n", snr, (int)experiment,
                           (int)nse, (int)nde, (int)nue, (double)nse / experiment / (n - r), (double)nde / experiment);
                if (nde >= 10 && (double)nde / experiment > 2.5 * reference_frame_error) { stop = true; break; }
            }
        }
        if (used < B) {  // stopped inside the batch: leave the generator where the frame-by-frame loop would
            Env::generator() = snapshot;
            for (long long i = 0, cnt = used * n; i < cnt; ++i) (void)Env::gaussian();
        }
        if (batch < max_batch) batch *= 4;
    }
    ldpc_hip_close_multi(ctx);                                                                   // :831
    if (counters_out) { counters_out->nse = nse; counters_out->nde = nde; counters_out->nue = nue; counters_out->experiment = experiment; counters_out->sum_abs_iters = sum_abs_iters; }
    return std::make_pair((double)nse / experiment / (n - r), (double)nde / experiment);         // :840
}

// the single-device form (device ordinal; LDPC_HIP_DEVICES overrides it)
template <class Mat, class Env>
std::pair<double, double> bp_simulation_t(int q_mod, Mat const &H, int tailbite_length, int max_iterations,
                                          int n_frame_errors, int n_experiments, double snr,
                                          double reference_frame_error, int decoder_type, int modulation_type,
                                          int permutation_type, int punctured_blocks, int show_process,
                                          SimCounters *counters_out = nullptr, int device = 0, long long max_batch = 4096,
                                          int permutation_block = 128, int permutation_inter = 1) {
    return bp_simulation_t<Mat, Env>(q_mod, H, tailbite_length, max_iterations, n_frame_errors, n_experiments, snr, reference_frame_error,
                                     decoder_type, modulation_type, permutation_type, punctured_blocks, show_process, counters_out,
                                     devices_from_env(device), max_batch, permutation_block, permutation_inter);
}

// Throughput mode.  Same arguments and return value as bp_simulation(); differences from exact-replay mode: the noise is the
// device-side Philox stream (seed), every modulation_type 0..4 is available (QAM16+ as the evidently intended chain), and
// `codewords` (optional, [ncw][n] 0/1 bytes) are really transmitted; ncw > 0 with codewords == nullptr transmits ncw random codewords
// encoded on the device (ldpc_hip_set_random_codewords).  The stopping rule is upstream's, frame by frame in global
// frame order (:591, :805-823), so the result equals a sequential loop over the same noise whatever the batch size or GPU count.
template <class Mat, class Env>
std::pair<double, double> bp_simulation_throughput_t(int q_mod, Mat const &H, int tailbite_length, int max_iterations, int n_frame_errors,
                                                     long long n_experiments, double snr, double reference_frame_error, int decoder_type,
                                                     int modulation_type, int permutation_type, int permutation_block, int permutation_inter,
                                                     int punctured_blocks, int show_process, unsigned long long seed,
                                                     const std::vector<int> &devices, SimCounters *counters_out = nullptr,
                                                     long long batch_per_gpu = 65536, const unsigned char *codewords = nullptr, int ncw = 0) {
    if (q_mod != 2) Env::fail("bp_simulation: only binary codes (q_mod == 2) are built in ldpc-lib_amd");
    if (devices.empty()) Env::fail("bp_simulation: empty device list");
    const int b = H.n_rows(), c = H.n_cols(), M = tailbite_length;
    const long long r = (long long)b * M, n = (long long)c * M;
    std::vector<int16_t> hd((size_t)b * c);
    for (int i = 0; i < b; ++i) for (int j = 0; j < c; ++j) hd[(size_t)i * c + j] = (int16_t)H(i, j);
    ldpc_hip_multi *m = nullptr;
    const int jit_before = ldpc_hip_set_jit_mode_thread(2);   // as in exact-replay mode: no waiting for hiprtc
    const int open_rc = ldpc_hip_open_multi(decoder_type, b, c, M, hd.data(), devices.data(), (int)devices.size(), &m);
    (void)ldpc_hip_set_jit_mode_thread(jit_before);
    if (open_rc != 0) Env::fail(ldpc_hip_last_error());
    if (ldpc_hip_multi_set_interleaver(m, permutation_type, permutation_block, permutation_inter) != 0) Env::fail(ldpc_hip_last_error());
    if (ncw > 0 && codewords && ldpc_hip_multi_set_codewords(m, codewords, ncw) != 0) Env::fail(ldpc_hip_last_error());
    if (ncw > 0 && !codewords && ldpc_hip_multi_set_random_codewords(m, seed, ncw) != 0) Env::fail(ldpc_hip_last_error());   // made on the device
    const int nsh = (int)devices.size();
    long long nse = 0, nue = 0, nde = 0, experiment = 0, sum_abs_iters = 0, first = 0;
    std::vector<int32_t> info, iters;
    bool stop = false;
    long long batch = 1024;   // ramp up like the exact harness: short runs stop after few frames
    while (!stop && nde < n_frame_errors && experiment <= n_experiments) {                       // :591
        const long long room = n_experiments + 1 - experiment;
        long long B = batch * nsh;
        if (B > room) B = room;
        info.resize((size_t)B); iters.resize((size_t)B);
        if (ldpc_hip_frames_multi(m, snr, modulation_type, punctured_blocks, max_iterations, 0.8 /*MS_ALPHA*/, seed, first, B, batch,
                                  info.data(), iters.data(), nullptr, nullptr) != 0)
            Env::fail(ldpc_hip_last_error());
        for (long long f = 0; f < B; ++f) {                                                      // ordered replay of :591-823
            if (!(nde < n_frame_errors && experiment <= n_experiments)) { stop = true; break; }
            ++experiment;
            const int it = iters[(size_t)f];
            sum_abs_iters += it < 0 ? -it : it;
            if (info[(size_t)f] != 0) {                                                          // bit 30: any wrong bit (:805)
                nse += info[(size_t)f] & ((1 << 30) - 1); ++nde;
                if (it >= 0) ++nue;
                if (show_process)
                    printf("SNR=%5.3lf,step=%4d,s_ers=%d,f_ers=%d,u_ers=%d,BER=%5.3le,FER=%5.3le\n", snr, (int)experiment, (int)nse, (int)nde,
                           (int)nue, (double)nse / experiment / (double)(n - r), (double)nde / experiment);
                if (nde >= 10 && (double)nde / experiment > 2.5 * reference_frame_error) { stop = true; break; }   // :820
            }
        }
        first += B;
        if (batch < batch_per_gpu) batch = batch * 4 < batch_per_gpu ? batch * 4 : batch_per_gpu;
    }
    ldpc_hip_close_multi(m);
    if (counters_out) { counters_out->nse = nse; counters_out->nde = nde; counters_out->nue = nue; counters_out->experiment = experiment; counters_out->sum_abs_iters = sum_abs_iters; }
    return std::make_pair((double)nse / experiment / (double)(n - r), (double)nde / experiment);
}

// Standalone entry point with upstream's argument list (bp_simulation.h:9-27); coef_matrix / ncols2convert only
// matter for q_mod > 2 (not built) and are accepted and ignored.
std::pair<double, double> bp_simulation(int q_mod, Matrix const &code_generating_matrix, Matrix &coef_matrix,
                                        int ncols2convert, int tailbite_length, int max_iterations, int n_frame_errors,
                                        int n_experiments, double snr, double reference_frame_error, int decoder_type,
                                        int modulation_type, int permutation_type, int permutation_block,
                                        int permutation_inter, int punctured_blocks, int show_process);

}  // namespace ldpc

#endif
