// bp_simulation_dropin.cpp -- defines upstream's exact symbol
//     std::pair<double,double> bp_simulation(int, matrix<int> const&, matrix<int>&, int x5, double, double, int x7)
// (bp_simulation.h:9-27) on the MI355X path.  Build it INSIDE the upstream tree (its bp_simulation.h,
// data_structures.h and commons_portable.h first on the include path) in place of the frame loop of
// bp_simulation.cpp:305-841; keep upstream's encoder (qc_encode / random_codeword / independent_validation,
// bp_simulation.cpp:22-191) in a file of its own: random_codeword() is still called here for its RNG side effect
// and its "Bad matrix" diagnostics (bp_simulation.cpp:512-527).  See INTEGRATION.md.
//
// Noise is drawn from upstream's own generator object (commons_portable.cpp:140 `std::mt19937 generator`, external
// linkage) through upstream's next_random_gaussian(), so the generator state the rest of the program sees is
// exactly what the original frame loop leaves behind.
#include "bp_simulation.h"  // upstream's: declares bp_simulation(), random_codeword(), pulls matrix<> and bit

#include <random>

#include "ldpc/bp_simulation.h"

extern std::mt19937 generator;             // commons_portable.cpp:140
void ensure_random_is_initialized();       // commons_portable.cpp:148

namespace {
struct UpstreamRngEnv {
    static std::mt19937 &generator() { ensure_random_is_initialized(); return ::generator; }
    static double gaussian() { return next_random_gaussian(); }
    static void burn_codeword_draws(matrix<int> const &H, int M) {
        std::vector<bit> codeword;
        const int rc = random_codeword(H, M, codeword);                       // bp_simulation.cpp:512
        if (rc < 0) printf("Bad matrix: zero codeword will be used\n");        // :517
        else if (rc > 0) printf("Bad encoding: zero codeword will be used\n");  // :525
    }
    [[noreturn]] static void fail(const char *msg) { die("%s", msg); }
};
}  // namespace

std::pair<double, double> bp_simulation(int q_mod, matrix<int> const &code_generating_matrix, matrix<int> &coef_matrix,
                                        int ncols2convert, int tailbite_length, int max_iterations, int n_frame_errors,
                                        int n_experiments, double snr, double reference_frame_error, int decoder_type,
                                        int modulation_type, int permutation_type, int permutation_block,
                                        int permutation_inter, int punctured_blocks, int show_process) {
    (void)coef_matrix; (void)ncols2convert;  // q_mod > 2 only
    return ldpc::bp_simulation_t<matrix<int>, UpstreamRngEnv>(q_mod, code_generating_matrix, tailbite_length, max_iterations,
                                                              n_frame_errors, n_experiments, snr, reference_frame_error,
                                                              decoder_type, modulation_type, permutation_type,
                                                              punctured_blocks, show_process, nullptr, 0, 4096, permutation_block,
                                                              permutation_inter);
}

// the definition above must be THE function upstream's header declares (not an overload)
static_assert(std::is_same<decltype(&bp_simulation),
                           std::pair<double, double> (*)(int, matrix<int> const &, matrix<int> &, int, int, int, int, int,
                                                         double, double, int, int, int, int, int, int, int)>::value,
              "bp_simulation signature differs from upstream's bp_simulation.h:9-27");
