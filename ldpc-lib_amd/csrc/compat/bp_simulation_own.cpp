// bp_simulation_own.cpp -- standalone instance of the harness: own matrix type, own generator (upstream's RNG
// contract, commons_portable.cpp:138-178), plus a C entry point for FFI callers and the tests.
#include "ldpc/bp_simulation.h"

namespace ldpc {

int initial_random_seed = 1;
namespace {
std::mt19937 g_generator(-1);   // commons_portable.cpp:140
bool g_initialized = false;     // :142
}  // namespace

void reset_random() { g_initialized = false; }  // :144-146

std::mt19937 &random_generator() {               // :148-160 ensure_random_is_initialized
    if (!g_initialized) {
        if (initial_random_seed == 0) {
            std::random_device device;
            initial_random_seed = (int)device();
            if (initial_random_seed == 0) initial_random_seed = 1;
        }
        g_generator = std::mt19937(initial_random_seed);
        g_initialized = true;
    }
    return g_generator;
}

int next_random_int(int min_inclusive, int max_exclusive) {  // :162-166
    std::uniform_int_distribution<int> dist(min_inclusive, max_exclusive - 1);
    return dist(random_generator());
}

double next_random_gaussian() {  // :174-178: a fresh distribution per call (the cached second value is dropped)
    std::normal_distribution<double> dist;
    return dist(random_generator());
}

std::pair<double, double> bp_simulation(int q_mod, Matrix const &H, Matrix &, int, int tailbite_length, int max_iterations,
                                        int n_frame_errors, int n_experiments, double snr, double reference_frame_error,
                                        int decoder_type, int modulation_type, int permutation_type, int permutation_block,
                                        int permutation_inter, int punctured_blocks, int show_process) {
    return bp_simulation_t<Matrix, OwnRngEnv>(q_mod, H, tailbite_length, max_iterations, n_frame_errors, n_experiments, snr,
                                              reference_frame_error, decoder_type, modulation_type, permutation_type,
                                              punctured_blocks, show_process, nullptr, 0, 4096, permutation_block, permutation_inter);
}

}  // namespace ldpc

// C entry point: H row-major rh x nh; seeds the generator like `initial_random_seed = seed; reset_random();`.
// out[0..6] = BER, FER, nse, nde, nue, experiment, sum|iters|; *rng_next = next raw mt19937 output (state fingerprint).
extern "C" int ldpc_bp_simulation_exact_perm(int rh, int nh, const int *H, int M, int max_iterations, int n_frame_errors,
                                             int n_experiments, double snr, double reference_frame_error, int decoder_type,
                                             int modulation_type, int permutation_type, int permutation_block, int permutation_inter,
                                             int punctured_blocks, unsigned seed, int device, double out[7], unsigned *rng_next) {
    ldpc::Matrix mat(rh, nh);
    for (int i = 0; i < rh * nh; ++i) mat.v[(size_t)i] = H[i];
    ldpc::initial_random_seed = (int)seed;
    ldpc::reset_random();
    ldpc::SimCounters cnt;
    const std::pair<double, double> res = ldpc::bp_simulation_t<ldpc::Matrix, ldpc::OwnRngEnv>(
        2, mat, M, max_iterations, n_frame_errors, n_experiments, snr, reference_frame_error, decoder_type, modulation_type,
        permutation_type, punctured_blocks, 0, &cnt, device, 4096, permutation_block, permutation_inter);
    out[0] = res.first; out[1] = res.second; out[2] = (double)cnt.nse; out[3] = (double)cnt.nde; out[4] = (double)cnt.nue;
    out[5] = (double)cnt.experiment; out[6] = (double)cnt.sum_abs_iters;
    if (rng_next) *rng_next = (unsigned)ldpc::random_generator()();
    return 0;
}

extern "C" int ldpc_bp_simulation_exact(int rh, int nh, const int *H, int M, int max_iterations, int n_frame_errors,
                                        int n_experiments, double snr, double reference_frame_error, int decoder_type,
                                        int modulation_type, int punctured_blocks, unsigned seed, int device, double out[7],
                                        unsigned *rng_next) {
    return ldpc_bp_simulation_exact_perm(rh, nh, H, M, max_iterations, n_frame_errors, n_experiments, snr, reference_frame_error,
                                         decoder_type, modulation_type, 0, 128, 1, punctured_blocks, seed, device, out, rng_next);
}

// Throughput mode (device-side noise, frames sharded over `devices`, upstream's sequential stopping rule on the ordered records).
// devices == NULL: LDPC_HIP_DEVICES or device 0.  codewords: [ncw][nh*M] 0/1 bytes or NULL (all-zero codeword).
extern "C" int ldpc_bp_simulation_throughput(int rh, int nh, const int *H, int M, int max_iterations, int n_frame_errors,
                                             long long n_experiments, double snr, double reference_frame_error, int decoder_type,
                                             int modulation_type, int permutation_type, int permutation_block, int permutation_inter,
                                             int punctured_blocks, unsigned long long seed, const int *devices, int n_devices,
                                             long long batch_per_gpu, const unsigned char *codewords, int ncw, double out[7]) {
    ldpc::Matrix mat(rh, nh);
    for (int i = 0; i < rh * nh; ++i) mat.v[(size_t)i] = H[i];
    const std::vector<int> devs = devices && n_devices > 0 ? std::vector<int>(devices, devices + n_devices) : ldpc::devices_from_env(0);
    ldpc::SimCounters cnt;
    const std::pair<double, double> res = ldpc::bp_simulation_throughput_t<ldpc::Matrix, ldpc::OwnRngEnv>(
        2, mat, M, max_iterations, n_frame_errors, n_experiments, snr, reference_frame_error, decoder_type, modulation_type, permutation_type,
        permutation_block, permutation_inter, punctured_blocks, 0, seed, devs, &cnt, batch_per_gpu, codewords, ncw);
    out[0] = res.first; out[1] = res.second; out[2] = (double)cnt.nse; out[3] = (double)cnt.nde; out[4] = (double)cnt.nue;
    out[5] = (double)cnt.experiment; out[6] = (double)cnt.sum_abs_iters;
    return 0;
}
