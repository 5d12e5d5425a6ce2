// harness_oracle.cpp -- sequential CPU restatement of bp_simulation() for binary codes.
// TEST INFRASTRUCTURE ONLY; "parity unpinned" at harness level -- see harness_oracle.h.
#include "harness_oracle.h"

#include <cmath>
#include <random>
#include <vector>

#include "ldpc_oracle.h"

namespace {

// commons_portable.cpp:160-178: a FRESH distribution object per call, on one shared mt19937
// (so the polar method's cached second value is thrown away every time -- SURVEY Appendix B Q3).
struct RefRng {
    std::mt19937 gen;
    explicit RefRng(unsigned int seed) : gen(seed) {}
    int next_int(int lo, int hi_excl) {
        std::uniform_int_distribution<int> dist(lo, hi_excl - 1);
        return dist(gen);
    }
    double next_gaussian() {
        std::normal_distribution<double> dist;
        return dist(gen);
    }
};

}  // namespace

extern "C" void orc_rng_gaussians(unsigned int seed, int n_int01, double *out, int n) {
    RefRng rng(seed);
    for (int i = 0; i < n_int01; i++) (void)rng.next_int(0, 2);
    for (int i = 0; i < n; i++) out[i] = rng.next_gaussian();
}

extern "C" void orc_awgn_llr(unsigned int seed, int n_int01, double snr, double rate, double *llr, int n) {
    RefRng rng(seed);
    for (int i = 0; i < n_int01; i++) (void)rng.next_int(0, 2);
    const double sigma = std::sqrt(std::pow(10, -snr / 10) / 2 / rate);  // bp_simulation.cpp:445
    for (int i = 0; i < n; i++) {
        double noise = rng.next_gaussian();
        llr[i] = -2.0 * (sigma * noise + 2.0 * 0.0 - 1.0) / (sigma * sigma);  // :603, codeword == 0 (:568)
    }
}

extern "C" int orc_bp_simulation_perm(int rh, int nh, const int *H, int M, int max_iterations, int n_frame_errors,
                                      int n_experiments, double snr, double reference_frame_error,
                                      int decoder_type, int modulation_type, int punctured_blocks,
                                      unsigned int seed, const int *inverse_map, orc_sim_result *out, int *iters_out) {
    if (!H || !out || rh <= 0 || nh <= rh || M <= 0) return -1;
    if (modulation_type != 0 && modulation_type != 1) return -2;  // QAM16+ wiring is broken upstream (Q5/Q6)
    const int b = rh, c = nh, r = b * M, n = c * M;

    std::vector<short> hd((size_t)rh * nh);
    for (int i = 0; i < rh * nh; i++) hd[i] = (short)H[i];
    orc_code *code = orc_open(rh, nh, M, hd.data());
    if (!code) return -3;

    int out_type;  // bp_simulation.cpp:451-466
    switch (decoder_type) {
    case 1: case 2: case 7: out_type = 1; break;    // SP_DEC, ASP_DEC, TASP_DEC
    case 0: case 3: case 4: case 8: out_type = 0; break;  // BP, MS, IMS, LMS
    default: orc_close(code); return -4;
    }

    const int QAM = modulation_type == 0 ? 1 : 4, halfmlog = 1;             // :405-406
    const double bitrate = (double)(c - b) / (c - punctured_blocks);         // :444
    const double sigma = std::sqrt(std::pow(10, -snr / 10) / 2 / bitrate);   // :445
    const double norm_factor = 2.0 * (QAM - 1.0) / 3.0;                      // :447
    const double sigmaQAM = std::sqrt(std::pow(10., -snr / 10.) / (2 * bitrate * halfmlog * 2) * norm_factor);  // :449

    RefRng rng(seed);
    // :512 random_codeword() -> :160-162 draws (nh - rh)*M values of next_random_int(0,2); the codeword
    // itself is then overwritten with zeros (:568).
    for (int i = b * M; i < n; i++) (void)rng.next_int(0, 2);

    std::vector<double> y(n), decword(n), buffer(n);
    long long nse = 0, nue = 0, nde = 0, experiment = 0, sum_abs_iter = 0;

    while (nde < n_frame_errors && experiment <= n_experiments) {  // :591
        ++experiment;
        const double sg = modulation_type == 0 ? sigma : sigmaQAM;
        for (int i = 0; i < n; ++i) {                              // :601-611
            double noise = rng.next_gaussian();
            buffer[i] = -2.0 * (sg * noise + 2.0 * 0.0 - 1.0) / (sg * sg);
        }
        for (int i = 0; i < n; ++i) y[i] = buffer[inverse_map ? inverse_map[i] : i];   // :684 Permutation(perm_state, 1, buffer, y)
        {                                                          // :699-709 puncturing
            const double init_val = out_type == 1 ? 0 : 0.5;
            const int plen = M * punctured_blocks, pstart = n - plen;
            for (int i = pstart; i < pstart + plen; i++) y[i] = init_val;
        }
        int iter;
        switch (decoder_type) {                                    // :716-729, DEC_DECISION 0, MS_ALPHA 0.8
        case 1: iter = orc_sum_prod(code, y.data(), decword.data(), max_iterations, 0); break;
        case 3: iter = orc_min_sum(code, y.data(), decword.data(), max_iterations, 0, 0.8); break;
        case 4: iter = orc_imin_sum(code, y.data(), decword.data(), max_iterations, 0, 0.8, 1.4, 6, 8); break;
        case 0: iter = orc_bp(code, y.data(), decword.data(), max_iterations, 0); break;
        case 2: iter = orc_sum_prod_gf2(code, y.data(), decword.data(), max_iterations, 0); break;
        case 7: iter = orc_tdmp_sum_prod(code, y.data(), decword.data(), max_iterations, nullptr); break;
        default: iter = orc_lmin_sum(code, y.data(), decword.data(), max_iterations, 0); break;
        }
        if (iters_out) iters_out[experiment - 1] = iter;
        sum_abs_iter += iter < 0 ? -iter : iter;

        int curr_nse = 0, curr_nse_info = 0;                        // :735-742
        for (int i = 0; i < n; i++) {
            if (decword[i] != 0.0) {
                ++curr_nse;
                if (i >= r) ++curr_nse_info;
            }
        }
        if (curr_nse > 0) {                                         // :805-823
            nse += curr_nse_info;
            ++nde;
            if (iter >= 0) ++nue;
            if (nde >= 10 && (double)nde / experiment > 2.5 * reference_frame_error) break;
        }
    }
    orc_close(code);

    out->ber = (double)nse / experiment / (n - r);
    out->fer = (double)nde / experiment;
    out->nse = nse; out->nde = nde; out->nue = nue; out->experiment = experiment;
    out->sum_abs_iter = sum_abs_iter;
    out->rng_next = (unsigned int)rng.gen();
    return 0;
}

extern "C" int orc_bp_simulation(int rh, int nh, const int *H, int M, int max_iterations, int n_frame_errors,
                                 int n_experiments, double snr, double reference_frame_error,
                                 int decoder_type, int modulation_type, int punctured_blocks,
                                 unsigned int seed, orc_sim_result *out, int *iters_out) {
    return orc_bp_simulation_perm(rh, nh, H, M, max_iterations, n_frame_errors, n_experiments, snr, reference_frame_error,
                                  decoder_type, modulation_type, punctured_blocks, seed, nullptr, out, iters_out);
}
