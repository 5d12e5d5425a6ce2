// ldpc_sim.cpp -- `ldpc_sim simulation <scenario.jsonx> <result.jsonx>`: the binary-code path of upstream's `simulation`
// driver (main_simulation.cpp:222-679) on top of this repository's bp_simulation (GPU decoders, upstream's noise order).
//
// What it keeps from upstream: the scenario fields (:260-283), the per-code fields (:343-364), the reduction of the shifts
// modulo the lifting with the special last parity column (:400-414), one generator reset per (code, SNR) so that all codes
// see the same noise (:483), the call into bp_simulation with reference_frame_error = error_minimization/threshold (:492-510),
// the result record written per code (:574-606).  What it does not do: girth / ACE tracing (trace_matrix, out of scope:
// `girth` is copied through), marking files (`_marking` must be "skip"), GF(q) codes (`_q_mod` > 2 are reported and skipped).
//
//   --throughput   device-side noise (counter-based Philox, not upstream's mt19937 stream): 10^6-10^7 frames/s per GPU; every
//                  modulation_type 0..4 and permutation_type 0..4; upstream's stopping rule frame by frame over the ordered records
//   --device N     GPU ordinal;  --devices 0,1,2,3 | all   shard the frames over several GPUs (RCCL all-reduce of the counters)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ldpc/bp_simulation.h"
#include "ldpc/decoders.h"
#include "ldpc/jsonx.h"
#include "ldpc_hip.h"

namespace {

[[noreturn]] void die(const std::string &msg) {
    fprintf(stderr, "%s\n", msg.c_str());
    exit(1);
}

// main_simulation.cpp:400-414: positive shifts are reduced modulo the lifting; a zero in the last column of the bidiagonal
// part (column index rows-1) becomes 1
void reduce_shifts(ldpc::Matrix &H, int M) {
    const int rows = H.n_rows(), cols = H.n_cols();
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j)
            if (H(i, j) > 0) {
                int t = H(i, j) % M;
                if (j == rows - 1) t = t == 0 ? 1 : t;
                H(i, j) = t;
            }
}

struct Pair { double ber, fer; };

}  // namespace

int main(int argc, char **argv) {
    bool throughput = false;
    int random_codewords = 0;   // --throughput only: transmit this many random codewords encoded on the device instead of the all-zero word
    int device = 0;
    std::string devices_arg;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--throughput")) throughput = true;
        else if (!strcmp(argv[i], "--random-codewords") && i + 1 < argc) random_codewords = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--devices") && i + 1 < argc) devices_arg = argv[++i];
        else pos.push_back(argv[i]);
    }
    // jsonx utilities (no GPU needed): re-emit a file in canonical form / print one value addressed like upstream's select()
    if (pos.size() == 3 && pos[0] == "jsonx") {
        try {
            const std::string text = jsonx::parse_file(pos[1]).dump();
            FILE *f = fopen(pos[2].c_str(), "wt");
            if (!f) die("cannot write '" + pos[2] + "'");
            fwrite(text.data(), 1, text.size(), f);
            fclose(f);
        } catch (const jsonx::Error &e) { die(e.what()); }
        return 0;
    }
    if (pos.size() == 3 && pos[0] == "jsonx-get") {
        try {
            const jsonx::Value root = jsonx::parse_file(pos[1]);
            const jsonx::Value &v = root.select(pos[2]);
            if (v.type == jsonx::Value::STRING) printf("%s\n", v.str.c_str());
            else fputs(v.dump().c_str(), stdout);
        } catch (const jsonx::Error &e) { die(e.what()); }
        return 0;
    }
    if (pos.size() != 3 || pos[0] != "simulation")
        die("usage: ldpc_sim simulation <scenario file name> <result file name> [--throughput [--random-codewords N]] [--device N | --devices 0,1,.. | --devices all]\n"
            "       ldpc_sim jsonx <in.jsonx> <out.jsonx>        re-emit in canonical form\n"
            "       ldpc_sim jsonx-get <in.jsonx> <path/to/field>  print one value (falls back to 'defaults' records)");

    if (!devices_arg.empty()) setenv("LDPC_HIP_DEVICES", devices_arg.c_str(), 1);
    const std::vector<int> devices = ldpc::devices_from_env(device);

    try {
        const jsonx::Value scenario = jsonx::parse_file(pos[1]);
        // 1. scenario (main_simulation.cpp:260-283)
        const int seed = (int)scenario.select("settings/random_seed").as_int();
        const long long num_experiments = scenario.select("settings/num_codewords").as_int();
        const int num_frame_errors = (int)scenario.select("settings/error_blocks").as_int();
        const std::string error_name = scenario.select("settings/error_minimization/name").as_string();
        const double error_rate_threshold = scenario.select("settings/error_minimization/threshold").as_double();
        if (error_name != "BER" && error_name != "FER") die("Unknown error name: '" + error_name + "'");
        const int modulation_type = (int)scenario.select("settings/modulation_type").as_int();
        const int permutation_type = (int)scenario.select("settings/permutation_type").as_int();
        const int permutation_block = (int)scenario.select("settings/permutation_block").as_int();
        const int permutation_inter = (int)scenario.select("settings/permutation_inter").as_int();
        printf("scenario OK\n");

        const jsonx::Value &codes = scenario.select("results");
        if (codes.type != jsonx::Value::ARRAY) die("'results' must be an array of code records");

        jsonx::Value out;
        out.set("settings", scenario.select("settings"));
        jsonx::Value &results = out.set("results", jsonx::Value::array());

        int rows = -1, columns = -1;
        for (size_t code_idx = 0; code_idx < codes.items.size(); ++code_idx) {
            const jsonx::Value &code = codes.items[code_idx];
            const std::vector<double> snrs = code.select("_SNRs").as_doubles();
            const int q_mod = code.has("_q_mod") ? (int)code.select("_q_mod").as_int() : 2;
            const int decoder_type = (int)code.select("_decoder_type").as_int();
            const int lifting = (int)code.select("_lifting").as_int();
            const int punctured_blocks = (int)code.select("_punctured_blocks").as_int();
            const int iterations = (int)code.select("_iterations").as_int();
            const std::string marking = code.select("_marking").as_string();
            if (q_mod > 2) { printf("code #%zu: GF(%d) codes (FHT decoder) are not built in ldpc-lib_amd, skipped\n", code_idx, q_mod); continue; }
            if (marking != "skip") die("code #" + std::to_string(code_idx) + ": marking files are not built (\"_marking\" must be \"skip\")");
            int r = 0, c = 0;
            const std::vector<int> cells = code.select("code").as_int_matrix(r, c);
            if (rows == -1) { rows = r; columns = c; }
            else if (rows != r || columns != c) { printf("Warning: unequal matrices in the input!\n"); continue; }   // :366-370
            ldpc::Matrix H(r, c);
            H.v = cells;
            reduce_shifts(H, lifting);
            const double bitrate = (double)(columns - rows) / (columns - punctured_blocks);                                  // :372

            std::vector<double> ber(snrs.size()), fer(snrs.size()), esn0(snrs.size());
            const auto t0 = std::chrono::steady_clock::now();
            for (size_t s = 0; s < snrs.size(); ++s) {
                if (s == 0) printf("====================================================\n");
                printf("code #%zu, original matrix is being processed, SNR = %6.3f\n", code_idx, snrs[s]);
                esn0[s] = snrs[s] + 10.0 * std::log10(2.0 * bitrate);                                                        // :481
                Pair res;
                if (throughput) {
                    const std::pair<double, double> p = ldpc::bp_simulation_throughput_t<ldpc::Matrix, ldpc::OwnRngEnv>(
                        2, H, lifting, iterations, num_frame_errors, num_experiments, snrs[s], error_rate_threshold, decoder_type, modulation_type,
                        permutation_type, permutation_block, permutation_inter, punctured_blocks, 0, (unsigned long long)seed, devices, nullptr, 65536, nullptr,
                        random_codewords);
                    res = {p.first, p.second};
                } else {
                    ldpc::initial_random_seed = seed;
                    ldpc::reset_random();                                                                                     // :483 all codes are tested with same noise
                    ldpc::Matrix coef;
                    const std::pair<double, double> p = ldpc::bp_simulation_t<ldpc::Matrix, ldpc::OwnRngEnv>(
                        2, H, lifting, iterations, num_frame_errors, (int)num_experiments, snrs[s], error_rate_threshold, decoder_type,
                        modulation_type, permutation_type, punctured_blocks, 0, nullptr, devices, 4096, permutation_block, permutation_inter);
                    (void)coef;
                    res = {p.first, p.second};
                }
                if (res.ber < 0 || res.fer < 0) res = {1.0, 1.0};                                                            // :527-531
                ber[s] = res.ber; fer[s] = res.fer;
            }
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

            printf("        |     FER     |     BER\n");
            for (size_t s = 0; s < snrs.size(); ++s) printf("%7.3f | %10.8f  | %10.8f\n", snrs[s], fer[s], ber[s]);
            printf("--------\n");

            // :574-606 the record upstream appends per code
            jsonx::Value log;
            log.set("SNR_per_bit___", jsonx::Value::numbers(snrs));
            log.set("SNR_per_symbol", jsonx::Value::numbers(esn0));
            log.set("BER", jsonx::Value::numbers(ber));
            log.set("FER", jsonx::Value::numbers(fer));
            jsonx::Value d;
            d.set("_decoder_name", jsonx::Value::string(decoder_type >= 0 && decoder_type < 10 ? DEC_FULL_NAME[decoder_type] : "?"));
            d.set("_decoder_type", jsonx::Value::number((long long)decoder_type));
            d.set("_lifting", jsonx::Value::number((long long)lifting));
            d.set("_SNRs", jsonx::Value::numbers(snrs));
            d.set("_punctured_blocks", jsonx::Value::number((long long)punctured_blocks));
            d.set("_iterations", jsonx::Value::number((long long)iterations));
            d.set("_marking", jsonx::Value::string("skip"));
            d.set("code_bitrate", jsonx::Value::number(bitrate));
            d.set("code", jsonx::Value::matrix(r, c, H.v));
            for (const char *key : {"column_weights", "row_weights", "girth", "config_index", "matrix_index", "code_index"})
                if (const jsonx::Value *v = code.find(key)) d.set(key, *v);
            d.set("simulation_logs", jsonx::Value::array({log}));
            d.set("time", jsonx::Value::number(std::round(secs * 1000.0) / 1000.0));
            results.items.push_back(std::move(d));
        }
        FILE *f = fopen(pos[2].c_str(), "wt");
        if (!f) die("cannot write '" + pos[2] + "'");
        const std::string text = out.dump();
        fwrite(text.data(), 1, text.size(), f);
        fclose(f);
    } catch (const jsonx::Error &e) {
        die(e.what());
    }
    return 0;
}
