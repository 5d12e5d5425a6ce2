// ab_ims.hip -- timing harness for build-time variants of the integer min-sum body (ldpc_spec::ims_body) on the shipped example code.
// Build one binary per variant (-DLDPC_IMS_PACK_IY=0/1, ...), run them alternately; each prints its time and a checksum of the
// outputs (hard decisions + iteration counts), which must agree between variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-D...] tools/ab_ims.hip -o tools/ab_ims_X.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../ldpc-lib_amd/csrc/ldpc_spec.hpp"
#include "../ldpc-lib_amd/csrc/code_appendix_c_m64.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#ifndef AB_WAVES
#define AB_WAVES 3
#endif
using ldpc_spec::SpecArgs;
__global__ void __launch_bounds__(64, AB_WAVES) k_ims(const SpecArgs a) { ldpc_spec::ims_body<ldpc_spec::CodeAppendixCM64>(a); }

int main(int argc, char **argv) {
    const long long B = argc > 1 ? atoll(argv[1]) : 65536;
    const double snr = argc > 2 ? atof(argv[2]) : 0.0;
    const int N = 2048, RH = 16, M = 64, rounds = 6;
    const long long distinct = std::min<long long>(B, 4096);
    std::vector<double> h((size_t)distinct * N), coef((size_t)B);
    std::mt19937_64 g(1);
    std::normal_distribution<double> nd;
    const double sigma = std::sqrt(std::pow(10, -snr / 10) / 2 / 0.5);
    for (auto &v : h) v = -2.0 * (sigma * nd(g) - 1.0) / (sigma * sigma);
    for (long long f = 0; f < B; ++f) {
        double en = 0;
        const double *y = &h[(size_t)(f % distinct) * N];
        for (int i = 0; i < N; ++i) en += y[i] * y[i];
        coef[(size_t)f] = std::sqrt((double)N / en);
    }
    double *d_llr, *d_coef;
    unsigned *d_hard;
    int *d_it;
    CK(hipMalloc(&d_llr, sizeof(double) * (size_t)B * N));
    CK(hipMalloc(&d_coef, sizeof(double) * (size_t)B));
    CK(hipMalloc(&d_hard, 4 * (size_t)B * (N / 32)));
    CK(hipMalloc(&d_it, 4 * (size_t)B));
    for (long long f = 0; f < B; f += distinct)
        CK(hipMemcpy(d_llr + (size_t)f * N, h.data(), sizeof(double) * (size_t)std::min(distinct, B - f) * N, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_coef, coef.data(), sizeof(double) * (size_t)B, hipMemcpyHostToDevice));
    const size_t lds = (((size_t)2 * N + 15) & ~(size_t)15) + (size_t)RH * 2 * LDPC_IMS_MSG_COPIES * M * 4 + 16;
    SpecArgs a{};
    a.llr = d_llr; a.hard = d_hard; a.iters = d_it; a.maxiter = 50; a.alpha = 0.8; a.nframes = B;
    a.ims_coef = d_coef; a.ims_thr = 1.4; a.ims_max_quant = 31; a.ims_max_data = 127; a.ims_ialpha = 12;
    void *args[] = {&a};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    for (int r = 0; r <= rounds; ++r) {
        CK(hipEventRecord(e0, 0));
        CK(hipLaunchKernel((const void *)k_ims, dim3((unsigned)B), dim3(64), args, lds, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) { best = std::min(best, ms); sum += ms; }
    }
    std::vector<unsigned> hh((size_t)B * (N / 32));
    std::vector<int> hi((size_t)B);
    CK(hipMemcpy(hh.data(), d_hard, 4 * hh.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hi.data(), d_it, 4 * hi.size(), hipMemcpyDeviceToHost));
    unsigned long long cs = 1469598103934665603ull;
    double mean_it = 0;
    for (unsigned x : hh) { cs ^= x; cs *= 1099511628211ull; }
    for (int x : hi) { cs ^= (unsigned)x; cs *= 1099511628211ull; mean_it += std::abs(x); }
    printf("%-28s frames %lld snr %.1f mean|it| %.2f  min %8.3f ms mean %8.3f ms  %6.3f Mframes/s  checksum %016llx\n", argc > 3 ? argv[3] : "", B, snr,
           mean_it / (double)B, best, sum / rounds, B / best / 1e3, cs);
    return 0;
}
