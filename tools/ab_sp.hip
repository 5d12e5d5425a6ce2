// ab_sp.hip -- timing harness for build-time variants of the sum-product body (ldpc_spec::sp_body) on the shipped example code:
// -DLDPC_SP_BODY_WAVES=W (sp_body) / -DLDPC_SP_WAVES=W (asp_body, bp_body via -DAB_BODY=...) = wavefronts per frame, -DAB_OCC=o = launch-bound
// waves per SIMD.  LDS bytes per frame are the 4th argument (sp 67600, asp 59408, bp 80032 for the example code).  One binary per variant; each prints its
// time and a checksum of hard decisions + iteration counts, which must agree between variants.
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
#ifndef AB_OCC
#define AB_OCC 4
#endif
#ifndef AB_BODY
#define AB_BODY sp_body
#endif
using ldpc_spec::SpecArgs;
#if defined(AB_THREADS)
constexpr int kThreads = AB_THREADS;   // e.g. -DAB_BODY=tasp_body -DAB_THREADS=64 -DAB_OCC=1 / -DAB_BODY=tasp_pair_body -DAB_THREADS=128 -DAB_OCC=2
#elif defined(AB_IS_SP)
constexpr int kThreads = ldpc_spec::kSpBodyWaves * 64;
#else
constexpr int kThreads = ldpc_spec::kSpWaves * 64;
#endif
__global__ void __launch_bounds__(kThreads, AB_OCC) k_sp(const SpecArgs a) { ldpc_spec::AB_BODY<ldpc_spec::CodeAppendixCM64>(a); }

int main(int argc, char **argv) {
    const long long B = argc > 1 ? atoll(argv[1]) : 16384;
    const double snr = argc > 2 ? atof(argv[2]) : 0.0;
    const size_t lds = argc > 4 ? (size_t)atoll(argv[4]) : 0;
    const int N = 2048, rounds = 5;
    const long long distinct = std::min<long long>(B, 4096);
    std::vector<double> h((size_t)distinct * N);
    std::mt19937_64 g(1);
    std::normal_distribution<double> nd;
    const double sigma = std::sqrt(std::pow(10, -snr / 10) / 2 / 0.5);
    for (auto &v : h) v = -2.0 * (sigma * nd(g) - 1.0) / (sigma * sigma);
    double *d_llr;
    unsigned *d_hard;
    int *d_it;
    CK(hipMalloc(&d_llr, sizeof(double) * (size_t)B * N));
    CK(hipMalloc(&d_hard, 4 * (size_t)B * (N / 32)));
    CK(hipMalloc(&d_it, 4 * (size_t)B));
    for (long long f = 0; f < B; f += distinct)
        CK(hipMemcpy(d_llr + (size_t)f * N, h.data(), sizeof(double) * (size_t)std::min(distinct, B - f) * N, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void *)k_sp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SpecArgs a{};
    a.llr = d_llr; a.hard = d_hard; a.iters = d_it; a.maxiter = argc > 5 ? atoi(argv[5]) : 50; a.alpha = 0.8; a.nframes = B;
    void *args[] = {&a};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    for (int r = 0; r <= rounds; ++r) {
        CK(hipEventRecord(e0, 0));
        CK(hipLaunchKernel((const void *)k_sp, dim3((unsigned)B), dim3(kThreads), args, lds, 0));
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
    printf("%-22s frames %lld snr %.1f mean|it| %.2f  min %8.3f ms mean %8.3f ms  %6.3f Mframes/s  checksum %016llx\n", argc > 3 ? argv[3] : "", B, snr,
           mean_it / (double)B, best, sum / rounds, B / best / 1e3, cs);
    return 0;
}
