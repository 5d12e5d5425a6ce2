// ab_flagship.hip -- A/B harness for variants of the flagship min-sum body (ldpc_spec::ms_m64_body) on the shipped example code:
// every variant decodes the SAME 65536 frames (Eb/N0 0 dB: all 50 iterations run), outputs are compared bit for bit with the
// baseline, rounds are interleaved in one process (guide rule 24).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17
// -ffp-contract=off tools/ab_flagship.hip -o tools/ab_flagship.bin.   Results: profiles/r02_flagship_variants.txt
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../ldpc-lib_amd/csrc/ldpc_spec.hpp"
#include "../ldpc-lib_amd/csrc/code_appendix_c_m64.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

namespace ldpc_spec {

// ---- variant W: the record word keeps the READY-TO-USE sign word (row parity folded in, slot 0 on bit 31) and the min1 slot in
// its low byte, so STATE1 and STATE3 do not rebuild it from the raw sign bits (popcount, negate, xor, shift) twice per iteration.
template <class C, bool DUAL_ADD>
__device__ __forceinline__ void ms_m64_body_w(const SpecArgs &a) {
    static_assert(C::M == 64, "one frame per wavefront needs M == 64");
    constexpr int RH = C::RH, NH = C::NH, N = C::NH * 64;
    extern __shared__ double lds[];
    char *const ldsb = reinterpret_cast<char *>(lds);
    const int lane = threadIdx.x;
    const u32 n8 = (u32)lane * 8u;
    const double alpha = a.alpha;
    const long long fr = blockIdx.x;
    auto rot = [&](u32 base, auto S) -> u32 {
        constexpr int c = decltype(S)::value;
        if constexpr (c == 0) return base;
        else return (base + 8u * (u32)c) & 511u;
    };
    const double *const yrow = a.llr + fr * N + lane;
    double m1[RH], m2[RH];
    u32 meta[RH];  // [31:32-RW] sign of the c2v on slot s (own v2c sign xor row sign) on bit 31-s; [7:0] slot of the min1 edge
    static_for<0, RH>([&](auto J) { constexpr int j = decltype(J)::value; m1[j] = 0.0; m2[j] = 0.0; meta[j] = 0u; });

    int res = -a.maxiter;
    for (int iter = 0; iter < a.maxiter; ++iter) {
        double y[NH];
        int yo = 0;
        asm volatile("" : "+v"(yo));
        static_for<0, NH>([&](auto K) { constexpr int k = decltype(K)::value; y[k] = yrow[yo + k * 64]; });
        // ---------------- STATE1
        static_for<0, RH>([&](auto J) {
            constexpr int j = decltype(J)::value;
            u32 mt = meta[j], nb = n8;
            asm volatile("" : "+v"(mt), "+v"(nb));
            const u32 pos = mt & 0xffu;
            u32 Wt = mt;
            static_for<0, C::RW[j]>([&](auto S) {
                constexpr int s = decltype(S)::value;
                constexpr int k = C::COL[j][s];
                double *p = reinterpret_cast<double *>(ldsb + rot(nb, IC<C::SH[j][s]>{}) + k * 512);
                if constexpr (DUAL_ADD) {
                    // no select: lanes whose min1 edge is this slot send min2, the others min1, as two exec-masked LDS operations
                    const double c1 = signed_mag(m1[j], Wt), c2 = signed_mag(m2[j], Wt);
                    Wt = twice(Wt);
                    if (pos == (u32)s) {
                        if constexpr (C::FIRST[j][s]) *p = c2;
                        else __hip_atomic_fetch_add(p, c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        if constexpr (C::FIRST[j][s]) *p = c1;
                        else __hip_atomic_fetch_add(p, c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else {
                    const double aa = sel64(m1[j], m2[j], lanes_eq(pos, (u32)s));
                    const double cv = signed_mag(aa, Wt);
                    Wt = twice(Wt);
                    if constexpr (C::FIRST[j][s]) *p = cv;
                    else __hip_atomic_fetch_add(p, cv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        // ---------------- STATE2
        static_for<0, NH>([&](auto K) {
            constexpr int k = decltype(K)::value;
            double *p = reinterpret_cast<double *>(ldsb + n8 + k * 512);
            const double pr = *p * alpha;
            *p = (y[k] + 0.0) + pr;
            if constexpr (k % 8 == 7) __builtin_amdgcn_sched_barrier(0);
        });
        // ---------------- STATE3
        u32 failw = 0;
        static_for<0, RH>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int RW = C::RW[j];
            u32 mt = meta[j];
            asm volatile("" : "+v"(mt));
            const u32 pos = mt & 0xffu;
            u32 Wt = mt;
            double a1 = m1[j] * alpha, a2 = m2[j] * alpha;
            asm volatile("" : "+v"(a1), "+v"(a2));
            double nm1 = kMaxVal, nm2 = kMaxVal;
            u32 npos = 0, nS = 0, sy = 0;
            u32 nb = n8;
            asm volatile("" : "+v"(nb));
            double r[RW];
            static_for<0, RW>([&](auto S) {
                constexpr int s = decltype(S)::value;
                r[s] = *reinterpret_cast<const double *>(ldsb + rot(nb, IC<C::SH[j][s]>{}) + C::COL[j][s] * 512);
            });
            static_for<0, RW>([&](auto S) {
                constexpr int s = decltype(S)::value;
                sy ^= hi32(r[s]);
                const double aa = sel64(a1, a2, lanes_eq(pos, (u32)s));
                const double x = signed_mag(aa, Wt);
                Wt = twice(Wt);
                const double tt = r[s] - x;
                nS = __builtin_amdgcn_alignbit(nS, hi32(tt), 31);
                const double v = fabs(tt);
                const mask64 c1 = lanes_lt(v, nm1);
                nm2 = fmin(fmax(v, nm1), nm2);
                npos = sel32(npos, (u32)s, c1);
                nm1 = fmin(v, nm1);
            });
            failw |= sy;
            // the word next iteration's STATE1 / STATE3 consume: signs xor row parity, slot 0 on bit 31, min1 slot in the low byte
            const u32 w = (nS ^ (0u - (__popc(nS) & 1u))) << (32 - RW);
            m1[j] = nm1; m2[j] = nm2; meta[j] = w | npos;
            __builtin_amdgcn_sched_barrier(0);
        });
        if (__ballot((failw >> 31) != 0) == 0ull) { res = iter + 1; break; }
    }
    if (lane == 0 && a.iters) a.iters[fr] = res;
    if (a.hard) {
        u64 mine = 0ull;
        static_for<0, NH>([&](auto K) {
            constexpr int k = decltype(K)::value;
            const u64 b = __ballot((hi32(*reinterpret_cast<const double *>(ldsb + n8 + k * 512)) >> 31) != 0);
            if (lane == k) mine = b;
        });
        if (lane < NH) reinterpret_cast<u64 *>(a.hard + fr * (N / 32))[lane] = mine;
    }
    if (a.soft_out) {
        static_for<0, NH>([&](auto K) {
            constexpr int k = decltype(K)::value;
            a.soft_out[fr * N + k * 64 + lane] = *reinterpret_cast<const double *>(ldsb + n8 + k * 512);
        });
    }
}

}  // namespace ldpc_spec

using ldpc_spec::SpecArgs;
__global__ void __launch_bounds__(64, 2) k_base(const SpecArgs a) { ldpc_spec::ms_m64_body<ldpc_spec::CodeAppendixCM64>(a); }
__global__ void __launch_bounds__(64, 2) k_w(const SpecArgs a) { ldpc_spec::ms_m64_body_w<ldpc_spec::CodeAppendixCM64, false>(a); }
__global__ void __launch_bounds__(64, 2) k_w_dual(const SpecArgs a) { ldpc_spec::ms_m64_body_w<ldpc_spec::CodeAppendixCM64, true>(a); }

int main(int argc, char **argv) {
    const long long B = argc > 1 ? atoll(argv[1]) : 65536;
    const double snr = argc > 2 ? atof(argv[2]) : 0.0;
    const int N = 2048, rounds = 6;
    const long long distinct = std::min<long long>(B, 4096);
    std::vector<double> h((size_t)distinct * N);
    std::mt19937_64 g(1);
    std::normal_distribution<double> nd;
    const double sigma = std::sqrt(std::pow(10, -snr / 10) / 2 / 0.5);
    for (auto &v : h) v = -2.0 * (sigma * nd(g) - 1.0) / (sigma * sigma);
    double *d_llr, *d_soft[3];
    unsigned *d_hard[3];
    int *d_it[3];
    CK(hipMalloc(&d_llr, sizeof(double) * (size_t)B * N));
    for (long long f = 0; f < B; f += distinct)
        CK(hipMemcpy(d_llr + (size_t)f * N, h.data(), sizeof(double) * (size_t)std::min(distinct, B - f) * N, hipMemcpyHostToDevice));
    for (int v = 0; v < 3; ++v) {
        CK(hipMalloc(&d_soft[v], sizeof(double) * (size_t)4096 * N));
        CK(hipMalloc(&d_hard[v], 4 * (size_t)B * (N / 32)));
        CK(hipMalloc(&d_it[v], 4 * (size_t)B));
    }
    const void *kern[3] = {(const void *)k_base, (const void *)k_w, (const void *)k_w_dual};
    const char *name[3] = {"baseline ms_m64_body", "W: sign word + min1 slot kept ready in the record", "W + exec-masked dual ds_add (no select in STATE1)"};
    std::vector<float> best(3, 1e9f), sum(3, 0.f);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r <= rounds; ++r)
        for (int v = 0; v < 3; ++v) {
            SpecArgs a{};
            a.llr = d_llr; a.hard = d_hard[v]; a.iters = d_it[v]; a.soft_out = nullptr; a.maxiter = 50; a.alpha = 0.8; a.nframes = B;
            void *args[] = {&a};
            CK(hipEventRecord(e0, 0));
            CK(hipLaunchKernel(kern[v], dim3((unsigned)B), dim3(64), args, N * 8, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r) { best[v] = std::min(best[v], ms); sum[v] += ms; }
        }
    // soft values on the first 4096 frames + bitwise comparison with the baseline
    std::vector<unsigned> hh[3];
    std::vector<int> hi[3];
    std::vector<double> hs[3];
    for (int v = 0; v < 3; ++v) {
        SpecArgs a{};
        a.llr = d_llr; a.hard = nullptr; a.iters = nullptr; a.soft_out = d_soft[v]; a.maxiter = 50; a.alpha = 0.8; a.nframes = 4096;
        void *args[] = {&a};
        CK(hipLaunchKernel(kern[v], dim3(4096), dim3(64), args, N * 8, 0));
        CK(hipDeviceSynchronize());
        hh[v].resize((size_t)B * (N / 32)); hi[v].resize((size_t)B); hs[v].resize((size_t)4096 * N);
        CK(hipMemcpy(hh[v].data(), d_hard[v], 4 * hh[v].size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hi[v].data(), d_it[v], 4 * hi[v].size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hs[v].data(), d_soft[v], 8 * hs[v].size(), hipMemcpyDeviceToHost));
    }
    double mean_it = 0;
    for (int x : hi[0]) mean_it += std::abs(x);
    mean_it /= (double)B;
    printf("# %lld frames, Eb/N0 %.1f dB, mean |iters| %.2f, %d interleaved rounds\n", B, snr, mean_it, rounds);
    for (int v = 0; v < 3; ++v) {
        const bool same = hh[v] == hh[0] && hi[v] == hi[0] && !memcmp(hs[v].data(), hs[0].data(), 8 * hs[0].size());
        printf("%-60s min %8.3f ms  mean %8.3f ms  %6.3f Mframes/s  outputs %s\n", name[v], best[v], sum[v] / rounds, B / best[v] / 1e3,
               same ? "bit-identical to the baseline" : "DIFFER");
    }
    return 0;
}
