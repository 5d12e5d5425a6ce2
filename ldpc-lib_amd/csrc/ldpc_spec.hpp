// ldpc_spec.hpp -- code-specialised decoders for binary QC-LDPC codes on gfx950, one `body<Code>` per decoder / lifting range:
//   ms_m64_body     flooding min-sum, M = 64, one frame == one wavefront (the flagship)
//   ms_small_body   flooding min-sum, M <= 32, floor(64/M) frames per wavefront
//   ms_chunk_body   flooding min-sum, 64 < M <= 128, one wavefront owning two 64-lane chunks
//   ms_body         flooding min-sum, any other M <= 512, ceil(M/64) wavefronts per frame
//   lms_body        layered offset min-sum (lms_small_body: M <= 32, several frames per wavefront)
//   ims_body        integer (int8) min-sum (ims_small_body: M <= 32, several frames per wavefront)
//   sp_body / asp_body / bp_body   flooding sum-product in the likelihood-ratio / probability / log domain (8 waves per frame)
//   tasp_body       TDMP (layered) sum-product in the probability domain
//
// The Tanner graph is a compile-time constant of this kernel: a `Code` type carries the base matrix (block row
// weights, block column and shift of every circulant) as constexpr tables, so the instruction stream contains
// the graph -- no descriptor loads, no scalar decode, no validity branches; a circulant's block column is the
// immediate offset of its ds_read/ds_add, its shift an immediate add (shift 0 needs no address arithmetic at
// all), its position in the row word an immediate.  The same source is used twice:
//   * ahead of time for the shipped example code (SURVEY Appendix C lifted to M = 64), compiled into libldpc_hip.so;
//   * just in time (hiprtc) at ldpc_hip_open() for any other base matrix: the host writes the `Code` struct for the
//     opened matrix, compiles this header for gfx950 and caches the code object per matrix.
// Algorithm, operation order and results are those of ms_flood_kernel / upstream min_sum_decod_qc_lm
// (decoders.cpp:4554-4767): bit-identical hard decisions, iteration counts and soft values.
//
// See ldpc_ms_fast.hpp for the arguments behind: sign-bit tests, LDS fp64 atomics in ascending block-row order, the
// first edge of a block column storing instead of adding, and MAX_VAL clamping folded into the min1/min2 start value.
//
// This header must stay self-contained (hiprtc compiles it without the rest of the tree).
#pragma once

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

namespace ldpc_spec {

typedef unsigned int u32;
typedef unsigned long long u64;

struct SpecArgs {
    const double *llr;   // [B][N]
    u32 *hard;           // [B][N/32] or null
    int *iters;          // [B] or null
    double *soft_out;    // [B][N] or null
    int maxiter;
    double alpha;
    // bp_body only (the other bodies ignore these and decode frame blockIdx.x):
    const u32 *stale;     // [B][RH*ceil(M/64)] u64 words (one per block row and 64-lane chunk, bit = lane): the syndrome the previous
                          // call left behind (upstream's DEC_STATE::syndr), or null = zeros
    u32 *synd_out;        // same layout: the syndrome this call leaves behind, or null
    const int *frame_idx; // [gridDim.x] frame decoded by each workgroup, or null = blockIdx.x
    // ims_body only: imin_sum_decod_qc_lm's quantiser (decoders.cpp:5445-5500)
    const double *ims_coef;   // [B] sqrt(N / sum y^2) per frame, from ims_coef_kernel (the sum is sequential: its rounding is part of the result)
    double ims_thr;
    int ims_max_quant, ims_max_data, ims_ialpha;
    long long nframes;        // frames in the batch (ms_small_body: several frames share a workgroup, the last one may be partly empty;
                              // ms_m64_body: where its frame queue ends)
    unsigned *queue;          // ms_m64_body only, or null: frame queue of a PERSISTENT launch.  The grid is one resident wave per
                              // slot of the chip instead of one workgroup per frame; wave b decodes frame b, then frames
                              // gridDim.x + atomicAdd(queue, 1) until nframes is reached (the host zeroes *queue before the launch).
                              // Frames converge after different numbers of iterations: pulling the next frame the moment a wave is
                              // free costs one atomic per frame and saves a workgroup launch per frame.
};

template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}

__device__ __forceinline__ u32 hi32(double x) { return (u32)__double2hiint(x); }
__device__ __forceinline__ u32 lo32(double x) { return (u32)__double2loint(x); }
__device__ __forceinline__ double mk(u32 hi, u32 lo) { return __hiloint2double((int)hi, (int)lo); }
// |mag| with the sign bit of sign_src (mag >= 0)
__device__ __forceinline__ double signed_mag(double mag, u32 sign_src) {
    return mk((hi32(mag) & 0x7fffffffu) | (sign_src & 0x80000000u), lo32(mag));
}

constexpr double kMaxVal = 32767.0;  // decoders.cpp:4299-4301

// Lane masks live in SGPR pairs and selects use the VOP3 form.  Measured on gfx950 (tools/ubench_valu3.hip): the
// VOP2 form `v_cndmask_b32_e32 ..., vcc` exposes ~20 cycles of latency per use at 2 waves per SIMD, the VOP3 form
// with an SGPR-pair mask issues at the normal half rate (~4.5 cycles).  hipcc shrinks selects to the VOP2 form
// whenever the mask sits in VCC, hence the explicit instruction.
typedef unsigned long long mask64;
__device__ __forceinline__ mask64 lanes_eq(u32 a, u32 b) { return __builtin_amdgcn_uicmp(a, b, 32 /*ICMP_EQ*/); }
__device__ __forceinline__ mask64 lanes_lt(double a, double b) { return __builtin_amdgcn_fcmp(a, b, 4 /*FCMP_OLT*/); }
__device__ __forceinline__ u32 sel32(u32 if0, u32 if1, mask64 m) {
    u32 d;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(if0), "v"(if1), "s"(m));
    return d;
}
// x + x as an explicit full-rate VOP2 add (the compiler would turn a chain of these back into half-rate shifts)
__device__ __forceinline__ u32 twice(u32 x) {
    u32 d;
    asm("v_add_u32 %0, %1, %1" : "=v"(d) : "v"(x));
    return d;
}
__device__ __forceinline__ double sel64(double if0, double if1, mask64 m) {
    return mk(sel32(hi32(if0), hi32(if1), m), sel32(lo32(if0), lo32(if1), m));
}

// Frame-wide "does any check fail?" for workgroups of several waves with ONE barrier: two flag words used alternately.  The
// word a vote uses was cleared by thread 0 during the previous vote, after that vote's barrier (by then every wave has read
// it for the vote before); every body has at least one more workgroup barrier before its next vote, which orders the clear
// before the next atomicOr.  Bodies clear both words before their first barrier (vote_init).
struct FrameVote {
    int *flags;
    int turn;
    __device__ __forceinline__ void init(int *f) { flags = f; turn = 0; if (threadIdx.x == 0) { f[0] = 0; f[1] = 0; } }
    __device__ __forceinline__ bool operator()(bool fail) {
        int *f = flags + turn;
        if (__ballot(fail) != 0ull && (threadIdx.x & 63) == 0) atomicOr(f, 1);
        __syncthreads();
        const bool r = *f != 0;
        turn ^= 1;
        if (threadIdx.x == 0) flags[turn] = 0;
        return r;
    }
};

// exp() as the reference computes it.  The sum-product decoders start from exp(LLR) (or exp(+-LLR/2)); everything after that
// is +, -, *, / in a fixed order, i.e. reproducible bit for bit -- so the one transcendental is evaluated here with glibc's own
// algorithm (sysdeps/ieee754/dbl-64/e_exp.c of glibc >= 2.28 = ARM optimized-routines exp, N = 128, polynomial order 5) in the
// evaluation order of the FMA build that x86-64 hosts with FMA select at run time: then the a-posteriori values of SP / ASP /
// TDMP are IDENTICAL to the CPU reference's, not merely close.  (tools/gen_exp_table.py computes the table;
// tests/test_host_cpu.py checks a C transcription against libm's exp() bit for bit.)  Valid for |x| < 700; callers clamp to 20.
__device__ __attribute__((aligned(16))) const unsigned long long kExpTab[256] = {
    0x0000000000000000ull, 0x3ff0000000000000ull, 0x3c9b3b4f1a88bf6eull, 0x3feff63da9fb3335ull,
    0xbc7160139cd8dc5dull, 0x3fefec9a3e778061ull, 0xbc905e7a108766d1ull, 0x3fefe315e86e7f85ull,
    0x3c8cd2523567f613ull, 0x3fefd9b0d3158574ull, 0xbc8bce8023f98efaull, 0x3fefd06b29ddf6deull,
    0x3c60f74e61e6c861ull, 0x3fefc74518759bc8ull, 0x3c90a3e45b33d399ull, 0x3fefbe3ecac6f383ull,
    0x3c979aa65d837b6dull, 0x3fefb5586cf9890full, 0x3c8eb51a92fdeffcull, 0x3fefac922b7247f7ull,
    0x3c3ebe3d702f9cd1ull, 0x3fefa3ec32d3d1a2ull, 0xbc6a033489906e0bull, 0x3fef9b66affed31bull,
    0xbc9556522a2fbd0eull, 0x3fef9301d0125b51ull, 0xbc5080ef8c4eea55ull, 0x3fef8abdc06c31ccull,
    0xbc91c923b9d5f416ull, 0x3fef829aaea92de0ull, 0x3c80d3e3e95c55afull, 0x3fef7a98c8a58e51ull,
    0xbc801b15eaa59348ull, 0x3fef72b83c7d517bull, 0xbc8f1ff055de323dull, 0x3fef6af9388c8deaull,
    0x3c8b898c3f1353bfull, 0x3fef635beb6fcb75ull, 0xbc96d99c7611eb26ull, 0x3fef5be084045cd4ull,
    0x3c9aecf73e3a2f60ull, 0x3fef54873168b9aaull, 0xbc8fe782cb86389dull, 0x3fef4d5022fcd91dull,
    0x3c8a6f4144a6c38dull, 0x3fef463b88628cd6ull, 0x3c807a05b0e4047dull, 0x3fef3f49917ddc96ull,
    0x3c968efde3a8a894ull, 0x3fef387a6e756238ull, 0x3c875e18f274487dull, 0x3fef31ce4fb2a63full,
    0x3c80472b981fe7f2ull, 0x3fef2b4565e27cddull, 0xbc96b87b3f71085eull, 0x3fef24dfe1f56381ull,
    0x3c82f7e16d09ab31ull, 0x3fef1e9df51fdee1ull, 0xbc3d219b1a6fbffaull, 0x3fef187fd0dad990ull,
    0x3c8b3782720c0ab4ull, 0x3fef1285a6e4030bull, 0x3c6e149289cecb8full, 0x3fef0cafa93e2f56ull,
    0x3c834d754db0abb6ull, 0x3fef06fe0a31b715ull, 0x3c864201e2ac744cull, 0x3fef0170fc4cd831ull,
    0x3c8fdd395dd3f84aull, 0x3feefc08b26416ffull, 0xbc86a3803b8e5b04ull, 0x3feef6c55f929ff1ull,
    0xbc924aedcc4b5068ull, 0x3feef1a7373aa9cbull, 0xbc9907f81b512d8eull, 0x3feeecae6d05d866ull,
    0xbc71d1e83e9436d2ull, 0x3feee7db34e59ff7ull, 0xbc991919b3ce1b15ull, 0x3feee32dc313a8e5ull,
    0x3c859f48a72a4c6dull, 0x3feedea64c123422ull, 0xbc9312607a28698aull, 0x3feeda4504ac801cull,
    0xbc58a78f4817895bull, 0x3feed60a21f72e2aull, 0xbc7c2c9b67499a1bull, 0x3feed1f5d950a897ull,
    0x3c4363ed60c2ac11ull, 0x3feece086061892dull, 0x3c9666093b0664efull, 0x3feeca41ed1d0057ull,
    0x3c6ecce1daa10379ull, 0x3feec6a2b5c13cd0ull, 0x3c93ff8e3f0f1230ull, 0x3feec32af0d7d3deull,
    0x3c7690cebb7aafb0ull, 0x3feebfdad5362a27ull, 0x3c931dbdeb54e077ull, 0x3feebcb299fddd0dull,
    0xbc8f94340071a38eull, 0x3feeb9b2769d2ca7ull, 0xbc87deccdc93a349ull, 0x3feeb6daa2cf6642ull,
    0xbc78dec6bd0f385full, 0x3feeb42b569d4f82ull, 0xbc861246ec7b5cf6ull, 0x3feeb1a4ca5d920full,
    0x3c93350518fdd78eull, 0x3feeaf4736b527daull, 0x3c7b98b72f8a9b05ull, 0x3feead12d497c7fdull,
    0x3c9063e1e21c5409ull, 0x3feeab07dd485429ull, 0x3c34c7855019c6eaull, 0x3feea9268a5946b7ull,
    0x3c9432e62b64c035ull, 0x3feea76f15ad2148ull, 0xbc8ce44a6199769full, 0x3feea5e1b976dc09ull,
    0xbc8c33c53bef4da8ull, 0x3feea47eb03a5585ull, 0xbc845378892be9aeull, 0x3feea34634ccc320ull,
    0xbc93cedd78565858ull, 0x3feea23882552225ull, 0x3c5710aa807e1964ull, 0x3feea155d44ca973ull,
    0xbc93b3efbf5e2228ull, 0x3feea09e667f3bcdull, 0xbc6a12ad8734b982ull, 0x3feea012750bdabfull,
    0xbc6367efb86da9eeull, 0x3fee9fb23c651a2full, 0xbc80dc3d54e08851ull, 0x3fee9f7df9519484ull,
    0xbc781f647e5a3ecfull, 0x3fee9f75e8ec5f74ull, 0xbc86ee4ac08b7db0ull, 0x3fee9f9a48a58174ull,
    0xbc8619321e55e68aull, 0x3fee9feb564267c9ull, 0x3c909ccb5e09d4d3ull, 0x3feea0694fde5d3full,
    0xbc7b32dcb94da51dull, 0x3feea11473eb0187ull, 0x3c94ecfd5467c06bull, 0x3feea1ed0130c132ull,
    0x3c65ebe1abd66c55ull, 0x3feea2f336cf4e62ull, 0xbc88a1c52fb3cf42ull, 0x3feea427543e1a12ull,
    0xbc9369b6f13b3734ull, 0x3feea589994cce13ull, 0xbc805e843a19ff1eull, 0x3feea71a4623c7adull,
    0xbc94d450d872576eull, 0x3feea8d99b4492edull, 0x3c90ad675b0e8a00ull, 0x3feeaac7d98a6699ull,
    0x3c8db72fc1f0eab4ull, 0x3feeace5422aa0dbull, 0xbc65b6609cc5e7ffull, 0x3feeaf3216b5448cull,
    0x3c7bf68359f35f44ull, 0x3feeb1ae99157736ull, 0xbc93091fa71e3d83ull, 0x3feeb45b0b91ffc6ull,
    0xbc5da9b88b6c1e29ull, 0x3feeb737b0cdc5e5ull, 0xbc6c23f97c90b959ull, 0x3feeba44cbc8520full,
    0xbc92434322f4f9aaull, 0x3feebd829fde4e50ull, 0xbc85ca6cd7668e4bull, 0x3feec0f170ca07baull,
    0x3c71affc2b91ce27ull, 0x3feec49182a3f090ull, 0x3c6dd235e10a73bbull, 0x3feec86319e32323ull,
    0xbc87c50422622263ull, 0x3feecc667b5de565ull, 0x3c8b1c86e3e231d5ull, 0x3feed09bec4a2d33ull,
    0xbc91bbd1d3bcbb15ull, 0x3feed503b23e255dull, 0x3c90cc319cee31d2ull, 0x3feed99e1330b358ull,
    0x3c8469846e735ab3ull, 0x3feede6b5579fdbfull, 0xbc82dfcd978e9db4ull, 0x3feee36bbfd3f37aull,
    0x3c8c1a7792cb3387ull, 0x3feee89f995ad3adull, 0xbc907b8f4ad1d9faull, 0x3feeee07298db666ull,
    0xbc55c3d956dcaebaull, 0x3feef3a2b84f15fbull, 0xbc90a40e3da6f640ull, 0x3feef9728de5593aull,
    0xbc68d6f438ad9334ull, 0x3feeff76f2fb5e47ull, 0xbc91eee26b588a35ull, 0x3fef05b030a1064aull,
    0x3c74ffd70a5fddcdull, 0x3fef0c1e904bc1d2ull, 0xbc91bdfbfa9298acull, 0x3fef12c25bd71e09ull,
    0x3c736eae30af0cb3ull, 0x3fef199bdd85529cull, 0x3c8ee3325c9ffd94ull, 0x3fef20ab5fffd07aull,
    0x3c84e08fd10959acull, 0x3fef27f12e57d14bull, 0x3c63cdaf384e1a67ull, 0x3fef2f6d9406e7b5ull,
    0x3c676b2c6c921968ull, 0x3fef3720dcef9069ull, 0xbc808a1883ccb5d2ull, 0x3fef3f0b555dc3faull,
    0xbc8fad5d3ffffa6full, 0x3fef472d4a07897cull, 0xbc900dae3875a949ull, 0x3fef4f87080d89f2ull,
    0x3c74a385a63d07a7ull, 0x3fef5818dcfba487ull, 0xbc82919e2040220full, 0x3fef60e316c98398ull,
    0x3c8e5a50d5c192acull, 0x3fef69e603db3285ull, 0x3c843a59ac016b4bull, 0x3fef7321f301b460ull,
    0xbc82d52107b43e1full, 0x3fef7c97337b9b5full, 0xbc892ab93b470dc9ull, 0x3fef864614f5a129ull,
    0x3c74b604603a88d3ull, 0x3fef902ee78b3ff6ull, 0x3c83c5ec519d7271ull, 0x3fef9a51fbc74c83ull,
    0xbc8ff7128fd391f0ull, 0x3fefa4afa2a490daull, 0xbc8dae98e223747dull, 0x3fefaf482d8e67f1ull,
    0x3c8ec3bc41aa2008ull, 0x3fefba1bee615a27ull, 0x3c842b94c3a9eb32ull, 0x3fefc52b376bba97ull,
    0x3c8a64a931d185eeull, 0x3fefd0765b6e4540ull, 0xbc8e37bae43be3edull, 0x3fefdbfdad9cbe14ull,
    0x3c77893b4d91cd9dull, 0x3fefe7c1819e90d8ull, 0x3c5305c14160cc89ull, 0x3feff3c22b8f71f1ull
};
template <class Tab>
__device__ __forceinline__ double exp_glibc_t(double x, Tab T) {
    const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p52, NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
    const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    const double z = InvLn2N * x;
    double kd = z + Shift;                                         // round to nearest integer, kept in the low mantissa bits
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= Shift;
    const double r = __fma_rn(kd, NegLn2loN, __fma_rn(kd, NegLn2hiN, x));   // x - k ln2/N
    const unsigned long long idx = 2 * (ki % 128);
    const ulonglong2 pair = *reinterpret_cast<const ulonglong2 *>(&T[idx]);   // (tail, sbits) sit side by side, 16-byte aligned: one load
    const double tail = __longlong_as_double((long long)pair.x);
    const unsigned long long sbits = pair.y + (ki << (52 - 7));
    const double r2 = r * r;
    const double tmp = __fma_rn(r2 * r2, __fma_rn(r, C5, C4), __fma_rn(r2, __fma_rn(r, C3, C2), tail + r));
    const double scale = __longlong_as_double((long long)sbits);
    return __fma_rn(scale, tmp, scale);
}
__device__ __forceinline__ double exp_glibc(double x) { return exp_glibc_t(x, kExpTab); }

// log() as the reference computes it (Gallager BP, decoder 0, is the one decoder with log inside its iteration): glibc >= 2.28's
// algorithm (sysdeps/ieee754/dbl-64/e_log.c = ARM optimized-routines log, N = 128, polynomial orders 6 and 12) in the evaluation
// order of the FMA build x86-64 hosts with FMA select at run time -- read off that build's instruction sequence.  kLogData is
// printed by tools/gen_log_table.py: [0] ln2hi [1] ln2lo [2..6] A [7..17] B [18 + 2i] invc_i [19 + 2i] logc_i.
// tests/test_host_cpu.py checks a C transcription against libm's log() bit for bit, all special cases included.
__device__ __attribute__((aligned(16))) const unsigned long long kLogData[274] = {
    0x3fe62e42fefa3800ull, 0x3d2ef35793c76730ull, 0xbfe0000000000001ull, 0x3fd555555551305bull,
    0xbfcfffffffeb4590ull, 0x3fc999b324f10111ull, 0xbfc55575e506c89full, 0xbfe0000000000000ull,
    0x3fd5555555555577ull, 0xbfcffffffffffdcbull, 0x3fc999999995dd0cull, 0xbfc55555556745a7ull,
    0x3fc24924a344de30ull, 0xbfbfffffa4423d65ull, 0x3fbc7184282ad6caull, 0xbfb999eb43b068ffull,
    0x3fb78182f7afd085ull, 0xbfb5521375d145cdull, 0x3ff734f0c3e0de9full, 0xbfd7cc7f79e69000ull,
    0x3ff713786a2ce91full, 0xbfd76feec20d0000ull, 0x3ff6f26008fab5a0ull, 0xbfd713e31351e000ull,
    0x3ff6d1a61f138c7dull, 0xbfd6b85b38287800ull, 0x3ff6b1490bc5b4d1ull, 0xbfd65d5590807800ull,
    0x3ff69147332f0cbaull, 0xbfd602d076180000ull, 0x3ff6719f18224223ull, 0xbfd5a8ca86909000ull,
    0x3ff6524f99a51ed9ull, 0xbfd54f4356035000ull, 0x3ff63356aa8f24c4ull, 0xbfd4f637c36b4000ull,
    0x3ff614b36b9ddc14ull, 0xbfd49da7fda85000ull, 0x3ff5f66452c65c4cull, 0xbfd445923989a800ull,
    0x3ff5d867b5912c4full, 0xbfd3edf439b0b800ull, 0x3ff5babccb5b90deull, 0xbfd396ce448f7000ull,
    0x3ff59d61f2d91a78ull, 0xbfd3401e17bda000ull, 0x3ff5805612465687ull, 0xbfd2e9e2ef468000ull,
    0x3ff56397cee76bd3ull, 0xbfd2941b3830e000ull, 0x3ff54725e2a77f93ull, 0xbfd23ec58cda8800ull,
    0x3ff52aff42064583ull, 0xbfd1e9e129279000ull, 0x3ff50f22dbb2bddfull, 0xbfd1956d2b48f800ull,
    0x3ff4f38f4734ded7ull, 0xbfd141679ab9f800ull, 0x3ff4d843cfde2840ull, 0xbfd0edd094ef9800ull,
    0x3ff4bd3ec078a3c8ull, 0xbfd09aa518db1000ull, 0x3ff4a27fc3e0258aull, 0xbfd047e65263b800ull,
    0x3ff4880524d48434ull, 0xbfcfeb224586f000ull, 0x3ff46dce1b192d0bull, 0xbfcf474a7517b000ull,
    0x3ff453d9d3391854ull, 0xbfcea4443d103000ull, 0x3ff43a2744b4845aull, 0xbfce020d44e9b000ull,
    0x3ff420b54115f8fbull, 0xbfcd60a22977f000ull, 0x3ff40782da3ef4b1ull, 0xbfccc00104959000ull,
    0x3ff3ee8f5d57fe8full, 0xbfcc202956891000ull, 0x3ff3d5d9a00b4ce9ull, 0xbfcb81178d811000ull,
    0x3ff3bd60c010c12bull, 0xbfcae2c9ccd3d000ull, 0x3ff3a5242b75dab8ull, 0xbfca45402e129000ull,
    0x3ff38d22cd9fd002ull, 0xbfc9a877681df000ull, 0x3ff3755bc5847a1cull, 0xbfc90c6d69483000ull,
    0x3ff35dce49ad36e2ull, 0xbfc87120a645c000ull, 0x3ff34679984dd440ull, 0xbfc7d68fb4143000ull,
    0x3ff32f5cceffcb24ull, 0xbfc73cb83c627000ull, 0x3ff3187775a10d49ull, 0xbfc6a39a9b376000ull,
    0x3ff301c8373e3990ull, 0xbfc60b3154b7a000ull, 0x3ff2eb4ebb95f841ull, 0xbfc5737d76243000ull,
    0x3ff2d50a0219a9d1ull, 0xbfc4dc7b8fc23000ull, 0x3ff2bef9a8b7fd2aull, 0xbfc4462c51d20000ull,
    0x3ff2a91c7a0c1babull, 0xbfc3b08abc830000ull, 0x3ff293726014b530ull, 0xbfc31b996b490000ull,
    0x3ff27dfa5757a1f5ull, 0xbfc2875490a44000ull, 0x3ff268b39b1d3bbfull, 0xbfc1f3b9f879a000ull,
    0x3ff2539d838ff5bdull, 0xbfc160c8252ca000ull, 0x3ff23eb7aac9083bull, 0xbfc0ce7f57f72000ull,
    0x3ff22a012ba940b6ull, 0xbfc03cdc49fea000ull, 0x3ff2157996cc4132ull, 0xbfbf57bdbc4b8000ull,
    0x3ff201201dd2fc9bull, 0xbfbe370896404000ull, 0x3ff1ecf4494d480bull, 0xbfbd17983ef94000ull,
    0x3ff1d8f5528f6569ull, 0xbfbbf9674ed8a000ull, 0x3ff1c52311577e7cull, 0xbfbadc79202f6000ull,
    0x3ff1b17c74cb26e9ull, 0xbfb9c0c3e7288000ull, 0x3ff19e010c2c1ab6ull, 0xbfb8a646b372c000ull,
    0x3ff18ab07bb670bdull, 0xbfb78d01b3ac0000ull, 0x3ff1778a25efbcb6ull, 0xbfb674f145380000ull,
    0x3ff1648d354c31daull, 0xbfb55e0e6d878000ull, 0x3ff151b990275fddull, 0xbfb4485cdea1e000ull,
    0x3ff13f0ea432d24cull, 0xbfb333d94d6aa000ull, 0x3ff12c8b7210f9daull, 0xbfb22079f8c56000ull,
    0x3ff11a3028ecb531ull, 0xbfb10e4698622000ull, 0x3ff107fbda8434afull, 0xbfaffa6c6ad20000ull,
    0x3ff0f5ee0f4e6bb3ull, 0xbfadda8d4a774000ull, 0x3ff0e4065d2a9fceull, 0xbfabbcece4850000ull,
    0x3ff0d244632ca521ull, 0xbfa9a1894012c000ull, 0x3ff0c0a77ce2981aull, 0xbfa788583302c000ull,
    0x3ff0af2f83c636d1ull, 0xbfa5715e67d68000ull, 0x3ff09ddb98a01339ull, 0xbfa35c8a49658000ull,
    0x3ff08cabaf52e7dfull, 0xbfa149e364154000ull, 0x3ff07b9f2f4e28fbull, 0xbf9e72c082eb8000ull,
    0x3ff06ab58c358f19ull, 0xbf9a55f152528000ull, 0x3ff059eea5ecf92cull, 0xbf963d62cf818000ull,
    0x3ff04949cdd12c90ull, 0xbf9228fb8caa0000ull, 0x3ff038c6c6f0ada9ull, 0xbf8c317b20f90000ull,
    0x3ff02865137932a9ull, 0xbf8419355daa0000ull, 0x3ff0182427ea7348ull, 0xbf781203c2ec0000ull,
    0x3ff008040614b195ull, 0xbf60040979240000ull, 0x3fefe01ff726fa1aull, 0x3f6feff384900000ull,
    0x3fefa11cc261ea74ull, 0x3f87dc41353d0000ull, 0x3fef6310b081992eull, 0x3f93cea3c4c28000ull,
    0x3fef25f63ceeadcdull, 0x3f9b9fc114890000ull, 0x3feee9c8039113e7ull, 0x3fa1b0d8ce110000ull,
    0x3feeae8078cbb1abull, 0x3fa58a5bd001c000ull, 0x3fee741aa29d0c9bull, 0x3fa95c8340d88000ull,
    0x3fee3a91830a99b5ull, 0x3fad276aef578000ull, 0x3fee01e009609a56ull, 0x3fb07598e598c000ull,
    0x3fedca01e577bb98ull, 0x3fb253f5e30d2000ull, 0x3fed92f20b7c9103ull, 0x3fb42edd8b380000ull,
    0x3fed5cac66fb5cceull, 0x3fb606598757c000ull, 0x3fed272caa5ede9dull, 0x3fb7da76356a0000ull,
    0x3fecf26e3e6b2ccdull, 0x3fb9ab434e1c6000ull, 0x3fecbe6da2a77902ull, 0x3fbb78c7bb0d6000ull,
    0x3fec8b266d37086dull, 0x3fbd431332e72000ull, 0x3fec5894bd5d5804ull, 0x3fbf0a3171de6000ull,
    0x3fec26b533bb9f8cull, 0x3fc067152b914000ull, 0x3febf583eeece73full, 0x3fc147858292b000ull,
    0x3febc4fd75db96c1ull, 0x3fc2266ecdca3000ull, 0x3feb951e0c864a28ull, 0x3fc303d7a6c55000ull,
    0x3feb65e2c5ef3e2cull, 0x3fc3dfc33c331000ull, 0x3feb374867c9888bull, 0x3fc4ba366b7a8000ull,
    0x3feb094b211d304aull, 0x3fc5933928d1f000ull, 0x3feadbe885f2ef7eull, 0x3fc66acd2418f000ull,
    0x3feaaf1d31603da2ull, 0x3fc740f8ec669000ull, 0x3fea82e63fd358a7ull, 0x3fc815c0f51af000ull,
    0x3fea5740ef09738bull, 0x3fc8e92954f68000ull, 0x3fea2c2a90ab4b27ull, 0x3fc9bb3602f84000ull,
    0x3fea01a01393f2d1ull, 0x3fca8bed1c2c0000ull, 0x3fe9d79f24db3c1bull, 0x3fcb5b515c01d000ull,
    0x3fe9ae2505c7b190ull, 0x3fcc2967ccbcc000ull, 0x3fe9852ef297ce2full, 0x3fccf635d5486000ull,
    0x3fe95cbaeea44b75ull, 0x3fcdc1bd3446c000ull, 0x3fe934c69de74838ull, 0x3fce8c01b8cfe000ull,
    0x3fe90d4f2f6752e6ull, 0x3fcf5509c0179000ull, 0x3fe8e6528effd79dull, 0x3fd00e6c121fb800ull,
    0x3fe8bfce9fcc007cull, 0x3fd071b80e93d000ull, 0x3fe899c0dabec30eull, 0x3fd0d46b9e867000ull,
    0x3fe87427aa2317fbull, 0x3fd13687334bd000ull, 0x3fe84f00acb39a08ull, 0x3fd1980d67234800ull,
    0x3fe82a49e8653e55ull, 0x3fd1f8ffe0cc8000ull, 0x3fe8060195f40260ull, 0x3fd2595fd7636800ull,
    0x3fe7e22563e0a329ull, 0x3fd2b9300914a800ull, 0x3fe7beb377dcb5adull, 0x3fd3187210436000ull,
    0x3fe79baa679725c2ull, 0x3fd377266dec1800ull, 0x3fe77907f2170657ull, 0x3fd3d54ffbaf3000ull,
    0x3fe756cadbd6130cull, 0x3fd432eee32fe000ull
};
template <class Tab>
__device__ __forceinline__ double log_glibc_t(double x, Tab T) {
    // the polynomial coefficients are compile-time constants whatever table pointer the look-up below uses (bp_body passes a copy in
    // LDS: through it they were 11 LDS reads per evaluation)
    auto D = [&](int i) { return __longlong_as_double((long long)kLogData[i]); };
    unsigned long long ix = (unsigned long long)__double_as_longlong(x);
    const u32 top = (u32)(ix >> 48);
    const unsigned long long LO = 0x3fee000000000000ull /* 1.0 - 0x1p-4 */, HI = 0x3ff1090000000000ull /* 1.0 + 0x1.09p-4 */;
    if (ix - LO < HI - LO) {                                          // close to 1.0
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double r = x - 1.0, r2 = r * r, r3 = r * r2;
        const double t3 = __fma_rn(r3, D(17), __fma_rn(r2, D(16), __fma_rn(r, D(15), D(14))));
        const double t2 = __fma_rn(r3, t3, __fma_rn(r2, D(13), __fma_rn(r, D(12), D(11))));
        const double t1 = __fma_rn(r3, t2, __fma_rn(r2, D(10), __fma_rn(r, D(9), D(8))));
        double w = r * 0x1p27;
        const double rhi = r + w - w, rlo = r - rhi;
        w = rhi * rhi * D(7);                                         // B[0] == -0.5
        const double hi = r + w;
        double lo = r - hi + w;
        lo = __fma_rn(D(7) * rlo, rhi + r, lo);
        return __fma_rn(t1, r3, lo) + hi;
    }
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) {                          // x < 0x1p-1022, inf or nan
        if (ix * 2 == 0) return -__longlong_as_double(0x7ff0000000000000ll);   // log(+-0) = -inf
        if (ix == 0x7ff0000000000000ull) return x;                     // log(inf) = inf
        if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return __longlong_as_double(0x7ff8000000000000ll) ;   // x < 0 or NaN: NaN (payload irrelevant here)
        ix = (unsigned long long)__double_as_longlong(x * 0x1p52);     // subnormal: normalise
        ix -= 52ull << 52;
    }
    const unsigned long long tmp = ix - 0x3fe6000000000000ull;         // x = 2^k z, z in [OFF, 2 OFF)
    const int i = (int)((tmp >> (52 - 7)) % 128);
    const long long k = (long long)tmp >> 52;
    const unsigned long long iz = ix - (tmp & (0xfffull << 52));
    const ulonglong2 cpair = *reinterpret_cast<const ulonglong2 *>(&T[18 + 2 * i]);   // (invc, logc): one 16-byte load
    const double invc = __longlong_as_double((long long)cpair.x), logc = __longlong_as_double((long long)cpair.y), z = __longlong_as_double((long long)iz);
    const double r = __fma_rn(z, invc, -1.0);
    const double kd = (double)k;
    const double w = __fma_rn(kd, D(0), logc);
    const double hi = w + r;
    const double lo = __fma_rn(kd, D(1), w - hi + r);
    const double r2 = r * r;
    const double p = __fma_rn(r, D(6), D(5));
    const double q = __fma_rn(r2, p, __fma_rn(r, D(4), D(3)));
    return __fma_rn(r * r2, q, __fma_rn(r2, D(2), lo)) + hi;
}
__device__ __forceinline__ double log_glibc(double x) { return log_glibc_t(x, kLogData); }
// exp() for arguments that may leave exp_glibc's range: glibc takes a separate path for |x| >= 512 (overflow / underflow handling);
// there the result is +inf / 0 or deep in the subnormal range, where ocml's exp agrees on the values that matter (inf, 0).  The BP
// decoder's arguments stay far below 512 unless a message was exactly 0 or 1 in the probability sense (log(0) = -inf).
template <class Tab>
__device__ __forceinline__ double exp_glibc_wide(double x, Tab T) {
    return fabs(x) < 512.0 ? exp_glibc_t(x, T) : exp(x);
}
constexpr int kBpTabWords = 256 + 274;   // kExpTab + kLogData: bp_body keeps a copy in LDS when the code leaves room (divergent table
                                         // look-ups from global memory cost the kernel a quarter of its speed)

// Upstream clamps with mind()/maxd() (decoders.cpp:104-105: a < b ? a : b and a < b ? b : a) or with `if (v < lo) v = lo`.
// For a value that is not NaN and ordinary constant bounds v_min_f64 / v_max_f64 return the same double in ONE instruction;
// the compare + select form is a v_cmp plus two v_cndmask_b32_e32 reading VCC, each with ~20 exposed cycles on gfx950
// (tools/ubench_valu3.hip).  A NaN cannot reach these clamps: every operand is a finite LLR, a probability in [0,1] with a
// strictly positive denominator, or the result of mind(.., finite) which already maps NaN to the bound.
__device__ __forceinline__ double at_most(double x, double hi) { return fmin(x, hi); }
__device__ __forceinline__ double at_least(double x, double lo) { return fmax(x, lo); }

// Hard decisions of the variables [0, N) -> the frame's packed words, by T threads (whole waves): lane l of a wave looks at variable
// 64 g + l, so that neighbouring lanes read neighbouring LDS words (conflict free), and a ballot makes two packed words at once.
// (The loop this replaces gave word w to thread w and walked its 32 variables: every lane of a wave then read the same LDS bank,
// a 64-way conflict per read -- ~19 % of the LDS-active cycles of the layered M = 512 and the TDMP M = 126 kernels in round 2's
// counters, for the same bits.)
// Frames of a PERSISTENT launch (SpecArgs::queue, see there): the workgroup decodes frame `first`, then frames gridDim.x + ticket
// until nframes is reached.  Between two frames every wave passes a barrier, so the next frame's set-up finds the LDS image free.
// Without a queue: the one frame.  QUEUED = false (several frames per workgroup: the small-lifting bodies): the one frame, always.
template <bool QUEUED, class F>
__device__ __forceinline__ void for_each_frame(const SpecArgs &a, long long fr, F body) {
    if constexpr (!QUEUED) {
        body(fr);
    } else {
        __shared__ unsigned q_ticket;
        while (fr < a.nframes) {
            body(fr);
            if (!a.queue) break;
            __syncthreads();
            if (threadIdx.x == 0) q_ticket = atomicAdd(a.queue, 1u);
            __syncthreads();
            fr = (long long)gridDim.x + (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)q_ticket);   // every wave ends at fr >= nframes
        }
    }
}

template <int N, int T, class Pred>
__device__ __forceinline__ void pack_hard(u32 *dst, const int tid, Pred pred) {
    constexpr int G = (N + 63) / 64, HW = (N + 31) / 32;
    const int lane = tid & 63;
    for (int g = tid >> 6; g < G; g += T / 64) {
        const int v = 64 * g + lane;
        const u64 m = __ballot(v < N && pred(v < N ? v : 0));
        if (lane == 0) {
            dst[2 * g] = (u32)m;
            if (2 * g + 1 < HW) dst[2 * g + 1] = (u32)(m >> 32);
        }
    }
}


template <class C>
__device__ __forceinline__ void ms_m64_body(const SpecArgs &a) {
    static_assert(C::M == 64, "ms_m64_body: one frame per wavefront needs M == 64");
    constexpr int RH = C::RH, NH = C::NH, N = C::NH * 64;
    extern __shared__ double lds[];  // [N] soft / acc (fp64)
    char *const ldsb = reinterpret_cast<char *>(lds);
    const int lane = threadIdx.x;
    const u32 n8 = (u32)lane * 8u;
    const double alpha = a.alpha;
    long long fr = blockIdx.x;  // one wave per frame; with a.queue the wave goes on to further frames (uniform: an SGPR pair)

    // LDS byte offset (inside a block column) of variable (lane + shift) mod 64.  `base` is an opaque per-row copy of
    // n8: it keeps the compiler from hoisting/CSE-ing the ~40 distinct rotated addresses out of the iteration loop
    // into long-lived VGPRs (that costs more in spills than the two integer ops per non-zero shift it saves).
    auto rot = [&](u32 base, auto S) -> u32 {
        constexpr int c = decltype(S)::value;
        if constexpr (c == 0) return base;
        else return (base + 8u * (u32)c) & 511u;
    };

  while (fr < a.nframes) {   // one pass without a queue
    const double *const yrow = a.llr + fr * N + lane;  // this frame's channel LLRs, variable (k, lane) at yrow[64 k]

    double m1[RH], m2[RH];
    u32 meta[RH];  // [31:32-RW] sign of the c2v on slot s (own v2c sign xor row parity) on bit 31-s, ready to use; [7:0] slot of the min1 edge
    static_for<0, RH>([&](auto J) {
        constexpr int j = decltype(J)::value;
        m1[j] = 0.0; m2[j] = 0.0; meta[j] = 0u;   // :4579-4596
    });

    int res = -a.maxiter;
    for (int iter = 0; iter < a.maxiter; ++iter) {
        // The channel LLRs are needed only in STATE2.  Instead of pinning 64 VGPRs for the whole kernel they are
        // re-read every iteration (16 KiB per frame: L2 / Infinity-Cache hits after the first pass) right here, so the
        // loads fly under STATE1's ALU work, and the registers are free again during STATE3 where pressure peaks.
        double y[NH];
        int yo = 0;
        asm volatile("" : "+v"(yo));  // opaque per iteration: the loads must not be hoisted out of the loop again
        static_for<0, NH>([&](auto K) {
            constexpr int k = decltype(K)::value;
            y[k] = yrow[yo + k * 64];
        });
        // ---------------- STATE1 (:4633-4667): acc[v] = sum of c2v, ascending block row
        static_for<0, RH>([&](auto J) {
            constexpr int j = decltype(J)::value;
            u32 mt = meta[j], nb = n8;
            // A volatile asm is ordered with the LDS operations around it, and everything this block row computes
            // depends on its outputs: the row's ALU work therefore stays between the previous row's LDS operations
            // and its own (otherwise instruction selection emits all 112 c2v computations first and spills them).
            asm volatile("" : "+v"(mt), "+v"(nb));
            const u32 pos = mt & 0xffu;
            // sign of the c2v on slot s = (own v2c sign) xor (row sign); slot s sits on bit RW-1-s of the row word.
            // Wt carries slot 0 on bit 31; every further slot is one full-rate add (Wt += Wt) instead of a shift.
            u32 Wt = mt;   // the record keeps the sign word ready: slot s on bit 31 - s, row parity folded in
            static_for<0, C::RW[j]>([&](auto S) {
                constexpr int s = decltype(S)::value;
                constexpr int k = C::COL[j][s];
                const double aa = sel64(m1[j], m2[j], lanes_eq(pos, (u32)s));
                const double cv = signed_mag(aa, Wt);
                Wt = twice(Wt);
                double *p = reinterpret_cast<double *>(ldsb + rot(nb, IC<C::SH[j][s]>{}) + k * 512);
                if constexpr (C::FIRST[j][s]) *p = cv;
                else __hip_atomic_fetch_add(p, cv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            });
            __builtin_amdgcn_sched_barrier(0);  // one block row at a time: keeps the live set (and the spills) small
        });
        // ---------------- STATE2 (:4670-4685): soft = y + acc*alpha (two roundings)
        static_for<0, NH>([&](auto K) {
            constexpr int k = decltype(K)::value;
            double *p = reinterpret_cast<double *>(ldsb + n8 + k * 512);
            const double pr = *p * alpha;
            *p = (y[k] + 0.0) + pr;   // + 0.0 canonicalises a -0.0 input (see ldpc_kernels.hpp); exact otherwise
            if constexpr (k % 8 == 7) __builtin_amdgcn_sched_barrier(0);  // 8 columns in flight, not 32 (VGPR budget)
        });
        // ---------------- STATE3 (:4690-4755)
        u32 failw = 0;
        static_for<0, RH>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int RW = C::RW[j];
            u32 mt = meta[j];
            // opaque: otherwise the compiler keeps STATE1's 112 select masks and shifted sign words alive across the
            // whole iteration to reuse them here (SGPR + VGPR spills to scratch); recomputing costs 3 ops per edge.
            asm volatile("" : "+v"(mt));
            const u32 pos = mt & 0xffu;
            u32 Wt = mt;   // the record keeps the sign word ready: slot s on bit 31 - s, row parity folded in
            double a1 = m1[j] * alpha, a2 = m2[j] * alpha;
            asm volatile("" : "+v"(a1), "+v"(a2));  // two products per ROW, not one per edge
            double nm1 = kMaxVal, nm2 = kMaxVal;    // start value == the MAX_VAL clamp of :4730
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
                const double tt = r[s] - x;              // v2c
                nS = __builtin_amdgcn_alignbit(nS, hi32(tt), 31);  // (nS << 1) | sign(tt): slot s lands on bit RW-1-s
                const double v = fabs(tt);
                const mask64 c1 = lanes_lt(v, nm1);      // strict: the first minimum keeps the position
                nm2 = fmin(fmax(v, nm1), nm2);           // = c1 ? nm1 : min(v, nm2)
                npos = sel32(npos, (u32)s, c1);
                nm1 = fmin(v, nm1);
            });
            failw |= sy;
            m1[j] = nm1; m2[j] = nm2; meta[j] = ((nS ^ (0u - (__popc(nS) & 1u))) << (32 - RW)) | npos;
            __builtin_amdgcn_sched_barrier(0);
        });
        if (__ballot((failw >> 31) != 0) == 0ull) { res = iter + 1; break; }  // :4761-4766
    }

    // ---------------- outputs
    if (lane == 0 && a.iters) a.iters[fr] = res;
    if (a.hard) {
        u64 mine = 0ull;
        static_for<0, NH>([&](auto K) {
            constexpr int k = decltype(K)::value;
            const u64 b = __ballot((hi32(*reinterpret_cast<const double *>(ldsb + n8 + k * 512)) >> 31) != 0);
            if (lane == k) mine = b;
        });
        // block column k = variables 64k..64k+63 = packed words 2k, 2k+1
        if (lane < NH) reinterpret_cast<u64 *>(a.hard + fr * (N / 32))[lane] = mine;
    }
    if (a.soft_out) {
        static_for<0, NH>([&](auto K) {
            constexpr int k = decltype(K)::value;
            a.soft_out[fr * N + k * 64 + lane] = *reinterpret_cast<const double *>(ldsb + n8 + k * 512);
        });
    }
    if (!a.queue) break;
    u32 ticket = 0;
    if (lane == 0) ticket = atomicAdd(a.queue, 1u);
    fr = (long long)gridDim.x + (long long)(u32)__builtin_amdgcn_readfirstlane((int)ticket);   // every wave ends here: fr >= nframes
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Flooding min-sum for ANY lifting M <= 512 (one frame per workgroup of W = ceil(M/64) wavefronts): the algorithm of
// ms_m64_body with workgroup barriers where several waves share a frame -- after every block row of STATE1 (the next
// row adds into the same variables and the order of the fp64 adds is part of the result), around STATE2, and in the
// syndrome vote.  M == 64 codes use ms_m64_body (no barriers, power-of-two rotation).
// ---------------------------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void ms_body(const SpecArgs &a) {
    constexpr int RH = C::RH, NH = C::NH, M = C::M, N = NH * M, W = (M + 63) / 64;
    constexpr bool POW2 = (M & (M - 1)) == 0;
    extern __shared__ double lds[];  // [N] soft / acc, then one flag word
    char *const ldsb = reinterpret_cast<char *>(lds);
    int *const flag = reinterpret_cast<int *>(ldsb + (size_t)N * 8);
    const int n = threadIdx.x;
    const bool valid = (M % 64 == 0) || n < M;
    const u32 n8 = (u32)(valid ? n : 0) * 8u;
    const double alpha = a.alpha;
    for_each_frame<true>(a, (long long)blockIdx.x, [&](const long long fr) {

    auto rot = [&](u32 base, auto S) -> u32 {
        constexpr int c = decltype(S)::value;
        if constexpr (c == 0) return base;
        else if constexpr (POW2) return (base + 8u * (u32)c) & (u32)(8 * M - 1);
        else { const u32 t = base + 8u * (u32)c, w = t - (u32)(8 * M); return t < w ? t : w; }   // unsigned min: t - 8M wraps when t < 8M (no VCC select)
    };
    FrameVote fvote;
    if constexpr (W > 1) fvote.init(flag);
    auto vote = [&](bool fail) -> bool {
        if constexpr (W == 1) return __ballot(fail) != 0ull;
        else return fvote(fail);
    };
    const double *const yrow = a.llr + fr * N + (valid ? n : 0);

    double m1[RH], m2[RH];
    u32 meta[RH];
    static_for<0, RH>([&](auto J) { constexpr int j = decltype(J)::value; m1[j] = 0.0; m2[j] = 0.0; meta[j] = 0u; });

    int res = -a.maxiter;
    for (int iter = 0; iter < a.maxiter; ++iter) {
        double y[NH];
        int yo = 0;
        asm volatile("" : "+v"(yo));
        static_for<0, NH>([&](auto K) { constexpr int k = decltype(K)::value; y[k] = yrow[yo + k * M]; });
        // ---------------- STATE1
        static_for<0, RH>([&](auto J) {
            constexpr int j = decltype(J)::value;
            u32 mt = meta[j], nb = n8;
            asm volatile("" : "+v"(mt), "+v"(nb) :: "memory");  // compiler fence: this row's work stays behind the previous barrier
            const u32 pos = mt & 0xffu;
            u32 Wt = mt;   // the record keeps the sign word ready: slot s on bit 31 - s, row parity folded in
            static_for<0, C::RW[j]>([&](auto S) {
                constexpr int s = decltype(S)::value;
                const double aa = sel64(m1[j], m2[j], lanes_eq(pos, (u32)s));
                const double cv = signed_mag(aa, Wt);
                Wt = twice(Wt);
                double *p = reinterpret_cast<double *>(ldsb + rot(nb, IC<C::SH[j][s]>{}) + C::COL[j][s] * (8 * M));
                if (valid) {
                    if constexpr (C::FIRST[j][s]) *p = cv;
                    else __hip_atomic_fetch_add(p, cv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            });
            if constexpr (W > 1) __syncthreads();
            else __builtin_amdgcn_sched_barrier(0);
        });
        // ---------------- STATE2
        if (valid) {
            static_for<0, NH>([&](auto K) {
                constexpr int k = decltype(K)::value;
                double *p = reinterpret_cast<double *>(ldsb + n8 + k * (8 * M));
                const double pr = *p * alpha;
                *p = (y[k] + 0.0) + pr;
                if constexpr (k % 8 == 7) __builtin_amdgcn_sched_barrier(0);
            });
        }
        if constexpr (W > 1) __syncthreads();
        // ---------------- STATE3
        u32 failw = 0;
        static_for<0, RH>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int RW = C::RW[j];
            u32 mt = meta[j];
            asm volatile("" : "+v"(mt) :: "memory");
            const u32 pos = mt & 0xffu;
            u32 Wt = mt;   // the record keeps the sign word ready: slot s on bit 31 - s, row parity folded in
            double a1 = m1[j] * alpha, a2 = m2[j] * alpha;
            asm volatile("" : "+v"(a1), "+v"(a2));
            double nm1 = kMaxVal, nm2 = kMaxVal;
            u32 npos = 0, nS = 0, sy = 0;
            u32 nb = n8;
            asm volatile("" : "+v"(nb));
            double r[RW];
            static_for<0, RW>([&](auto S) {
                constexpr int s = decltype(S)::value;
                r[s] = *reinterpret_cast<const double *>(ldsb + rot(nb, IC<C::SH[j][s]>{}) + C::COL[j][s] * (8 * M));
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
            m1[j] = nm1; m2[j] = nm2; meta[j] = ((nS ^ (0u - (__popc(nS) & 1u))) << (32 - RW)) | npos;
            // materialise the new record HERE: it is only consumed by the next iteration, and without this the compiler
            // sinks all 112 min1/min2 updates below the convergence branch and keeps every v2c alive across the vote
            asm volatile("" : "+v"(m1[j]), "+v"(m2[j]), "+v"(meta[j]));
            __builtin_amdgcn_sched_barrier(0);
        });
        if (!vote(valid && (failw >> 31) != 0)) { res = iter + 1; break; }   // the vote's barriers also fence the next STATE1
    }

    if (threadIdx.x == 0 && a.iters) a.iters[fr] = res;
    if (a.hard) {
        constexpr int HW = (N + 31) / 32;
        if constexpr (M % 64 == 0) {
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            static_for<0, NH>([&](auto K) {
                constexpr int k = decltype(K)::value;
                const u64 b = __ballot((*reinterpret_cast<const u32 *>(ldsb + n8 + k * (8 * M) + 4) >> 31) != 0);
                if (lane == 0) reinterpret_cast<u64 *>(a.hard + fr * HW)[k * W + wave] = b;
            });
        } else {
            pack_hard<N, W * 64>(a.hard + fr * HW, threadIdx.x, [&](int v) { return (*reinterpret_cast<const u32 *>(ldsb + (size_t)v * 8 + 4) >> 31) != 0; });
        }
    }
    if (a.soft_out && valid) {
        static_for<0, NH>([&](auto K) {
            constexpr int k = decltype(K)::value;
            a.soft_out[fr * N + k * M + n] = *reinterpret_cast<const double *>(ldsb + n8 + k * (8 * M));
        });
    }
    });
}

// ---------------------------------------------------------------------------------------------------------------
// Flooding min-sum for SMALL liftings (M <= 32): F = floor(64 / M) frames share one wavefront, lane = f*M + n.
// Same arithmetic and order as ms_m64_body per frame.  The frames of a wave converge at different iterations and upstream
// stops a frame the moment its syndrome clears (its output comes from that very iteration), so a converged frame's lanes
// freeze -- no more LDS writes, no record updates -- while the others go on; the wave ends when every frame has.
// LDS: soft[f][N] fp64 (F*N*8 <= 16 KiB).
// ---------------------------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void ms_small_body(const SpecArgs &a) {
    constexpr int RH = C::RH, NH = C::NH, M = C::M, N = NH * M, F = 64 / M;
    static_assert(M <= 32 && F >= 2, "ms_small_body: M <= 32");
    constexpr bool POW2 = (M & (M - 1)) == 0;
    extern __shared__ double lds[];  // [F][N] soft / acc
    const int lane = threadIdx.x;
    const int f = lane / M, n = lane - f * M;
    const long long fr = (long long)blockIdx.x * F + f;
    const bool live = f < F && fr < a.nframes;             // lanes past F*M and frames past the batch sit out
    char *const ldsb = reinterpret_cast<char *>(lds) + (size_t)(live ? f : 0) * N * 8;   // this lane's frame
    const u32 n8 = (u32)n * 8u;
    const double alpha = a.alpha;
    const u64 frame_lanes = (M == 32 ? 0xffffffffull : ((1ull << M) - 1ull)) << ((live ? f : 0) * M);

    auto rot = [&](u32 base, auto S) -> u32 {
        constexpr int c = decltype(S)::value;
        if constexpr (c == 0) return base;
        else if constexpr (POW2) return (base + 8u * (u32)c) & (u32)(8 * M - 1);
        else { const u32 t = base + 8u * (u32)c, w = t - (u32)(8 * M); return t < w ? t : w; }
    };
    const double *const yrow = a.llr + (live ? fr : 0) * N + n;

    double m1[RH], m2[RH];
    u32 meta[RH];
    static_for<0, RH>([&](auto J) { constexpr int j = decltype(J)::value; m1[j] = 0.0; m2[j] = 0.0; meta[j] = 0u; });

    bool done = !live;
    int res = -a.maxiter;
    for (int iter = 0; iter < a.maxiter; ++iter) {
        double y[NH];
        int yo = 0;
        asm volatile("" : "+v"(yo));
        static_for<0, NH>([&](auto K) { constexpr int k = decltype(K)::value; y[k] = yrow[yo + k * M]; });
        // ---------------- STATE1 (one divergent region per phase: lanes of frames that have converged sit out)
        if (!done) static_for<0, RH>([&](auto J) {
            constexpr int j = decltype(J)::value;
            u32 mt = meta[j], nb = n8;
            asm volatile("" : "+v"(mt), "+v"(nb));
            const u32 pos = mt & 0xffu;
            u32 Wt = mt;   // the record keeps the sign word ready: slot s on bit 31 - s, row parity folded in
            static_for<0, C::RW[j]>([&](auto S) {
                constexpr int s = decltype(S)::value;
                const double aa = sel64(m1[j], m2[j], lanes_eq(pos, (u32)s));
                const double cv = signed_mag(aa, Wt);
                Wt = twice(Wt);
                double *p = reinterpret_cast<double *>(ldsb + rot(nb, IC<C::SH[j][s]>{}) + C::COL[j][s] * (8 * M));
                if constexpr (C::FIRST[j][s]) *p = cv;
                else __hip_atomic_fetch_add(p, cv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("" ::: "memory");
        // ---------------- STATE2
        if (!done) {
            static_for<0, NH>([&](auto K) {
                constexpr int k = decltype(K)::value;
                double *p = reinterpret_cast<double *>(ldsb + n8 + k * (8 * M));
                const double pr = *p * alpha;
                *p = (y[k] + 0.0) + pr;
                if constexpr (k % 8 == 7) __builtin_amdgcn_sched_barrier(0);
            });
        }
        asm volatile("" ::: "memory");
        // ---------------- STATE3
        u32 failw = 0;
        if (!done) static_for<0, RH>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int RW = C::RW[j];
            u32 mt = meta[j];
            asm volatile("" : "+v"(mt));
            const u32 pos = mt & 0xffu;
            u32 Wt = mt;   // the record keeps the sign word ready: slot s on bit 31 - s, row parity folded in
            double a1 = m1[j] * alpha, a2 = m2[j] * alpha;
            asm volatile("" : "+v"(a1), "+v"(a2));
            double nm1 = kMaxVal, nm2 = kMaxVal;
            u32 npos = 0, nS = 0, sy = 0;
            u32 nb = n8;
            asm volatile("" : "+v"(nb));
            double r[RW];
            static_for<0, RW>([&](auto S) {
                constexpr int s = decltype(S)::value;
                r[s] = *reinterpret_cast<const double *>(ldsb + rot(nb, IC<C::SH[j][s]>{}) + C::COL[j][s] * (8 * M));
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
            m1[j] = nm1; m2[j] = nm2; meta[j] = ((nS ^ (0u - (__popc(nS) & 1u))) << (32 - RW)) | npos;
            asm volatile("" : "+v"(m1[j]), "+v"(m2[j]), "+v"(meta[j]));
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("" ::: "memory");
        const u64 failing = __ballot(!done && (failw >> 31) != 0);          // per check, frames that are still running
        if (!done && (failing & frame_lanes) == 0ull) { done = true; res = iter + 1; }   // :4761-4766, per frame
        if (__ballot(!done) == 0ull) break;
    }

    if (!live) return;
    if (n == 0 && a.iters) a.iters[fr] = res;
    if (a.hard) {
        constexpr int HW = (N + 31) / 32;
        for (int w = n; w < HW; w += M) {
            u32 bits = 0;
            for (int b = 0; b < 32; ++b) {
                const int v = 32 * w + b;
                if (v < N) bits |= (*reinterpret_cast<const u32 *>(ldsb + (size_t)v * 8 + 4) >> 31) << b;
            }
            a.hard[fr * HW + w] = bits;
        }
    }
    if (a.soft_out) {
        static_for<0, NH>([&](auto K) {
            constexpr int k = decltype(K)::value;
            a.soft_out[fr * N + k * M + n] = *reinterpret_cast<const double *>(ldsb + n8 + k * (8 * M));
        });
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Flooding min-sum for 64 < M <= 128 on ONE wavefront per frame: every lane owns the checks / variables n and n + 64 of
// each circulant (two "chunks").  Same arithmetic and order as ms_body, but a single wave needs no workgroup barrier at
// all -- ms_body synchronises its waves after every block row of STATE1 (the order of the fp64 adds into a variable is
// part of the result), 20 barriers per iteration -- because LDS executes one wave's accesses in program order.  The two
// chunks give the scheduler two independent instruction streams.  This is the kernel for upstream's shipped liftings
// (126, 67).  Records of both chunks live in VGPRs (160), channel values are re-read per chunk in STATE2.
// ---------------------------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void ms_chunk_body(const SpecArgs &a) {
    constexpr int RH = C::RH, NH = C::NH, M = C::M, N = NH * M, CH = (M + 63) / 64;
    static_assert(M > 64 && CH == 2, "ms_chunk_body: 64 < M <= 128");
    constexpr bool POW2 = (M & (M - 1)) == 0;
    extern __shared__ double lds[];  // [N] soft / acc
    char *const ldsb = reinterpret_cast<char *>(lds);
    const int lane = threadIdx.x;
    const double alpha = a.alpha;
    for_each_frame<true>(a, (long long)blockIdx.x, [&](const long long fr) {
    bool ok[CH];
    u32 n8[CH];
    static_for<0, CH>([&](auto Q) {
        constexpr int ch = decltype(Q)::value;
        ok[ch] = (64 * ch + 64 <= M) || (64 * ch + lane < M);
        n8[ch] = (u32)(ok[ch] ? 64 * ch + lane : 0) * 8u;
    });
    auto rot = [&](u32 base, auto S) -> u32 {
        constexpr int c = decltype(S)::value;
        if constexpr (c == 0) return base;
        else if constexpr (POW2) return (base + 8u * (u32)c) & (u32)(8 * M - 1);
        else { const u32 t = base + 8u * (u32)c, w = t - (u32)(8 * M); return t < w ? t : w; }
    };

    double m1[CH][RH], m2[CH][RH];
    u32 meta[CH][RH];
    static_for<0, CH>([&](auto Q) {
        static_for<0, RH>([&](auto J) {
            constexpr int ch = decltype(Q)::value, j = decltype(J)::value;
            m1[ch][j] = 0.0; m2[ch][j] = 0.0; meta[ch][j] = 0u;
        });
    });

    int res = -a.maxiter;
    for (int iter = 0; iter < a.maxiter; ++iter) {
        // ---------------- STATE1: block rows ascending, both chunks of a row before the next row
        static_for<0, RH>([&](auto J) {
            constexpr int j = decltype(J)::value;
            static_for<0, CH>([&](auto Q) {
                constexpr int ch = decltype(Q)::value;
                u32 mt = meta[ch][j], nb = n8[ch];
                asm volatile("" : "+v"(mt), "+v"(nb) :: "memory");
                const u32 pos = mt & 0xffu;
                u32 Wt = mt;   // the record keeps the sign word ready: slot s on bit 31 - s, row parity folded in
                static_for<0, C::RW[j]>([&](auto S) {
                    constexpr int s = decltype(S)::value;
                    const double aa = sel64(m1[ch][j], m2[ch][j], lanes_eq(pos, (u32)s));
                    const double cv = signed_mag(aa, Wt);
                    Wt = twice(Wt);
                    double *p = reinterpret_cast<double *>(ldsb + rot(nb, IC<C::SH[j][s]>{}) + C::COL[j][s] * (8 * M));
                    if (ok[ch]) {
                        if constexpr (C::FIRST[j][s]) *p = cv;                   // the column's first edge stores (both chunks: disjoint variables)
                        else __hip_atomic_fetch_add(p, cv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                });
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("" ::: "memory");
        // ---------------- STATE2, one chunk at a time, channel values in groups of 8 (the next group is in flight while
        // this one is used): keeps the transient registers at 32 next to the 160 of the two chunks' records
        static_for<0, CH>([&](auto Q) {
            constexpr int ch = decltype(Q)::value;
            constexpr int G = 8, NG = (NH + G - 1) / G;
            double y[2][G];
            int yo = ok[ch] ? 64 * ch + lane : 0;
            asm volatile("" : "+v"(yo));
            auto request = [&](auto GI) {
                constexpr int g = decltype(GI)::value;
                static_for<g * G, (g * G + G < NH ? g * G + G : NH)>([&](auto K) { constexpr int k = decltype(K)::value; y[g & 1][k % G] = a.llr[fr * N + yo + k * M]; });
            };
            request(IC<0>{});
            static_for<0, NG>([&](auto GI) {
                constexpr int g = decltype(GI)::value;
                if constexpr (g + 1 < NG) request(IC<g + 1>{});
                __builtin_amdgcn_sched_barrier(0);
                if (ok[ch]) {
                    static_for<g * G, (g * G + G < NH ? g * G + G : NH)>([&](auto K) {
                        constexpr int k = decltype(K)::value;
                        double *p = reinterpret_cast<double *>(ldsb + n8[ch] + k * (8 * M));
                        const double pr = *p * alpha;
                        *p = (y[g & 1][k % G] + 0.0) + pr;
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        asm volatile("" ::: "memory");
        // ---------------- STATE3
        u32 failw = 0;
        static_for<0, RH>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int RW = C::RW[j];
            static_for<0, CH>([&](auto Q) {
                constexpr int ch = decltype(Q)::value;
                u32 mt = meta[ch][j];
                asm volatile("" : "+v"(mt));
                const u32 pos = mt & 0xffu;
                u32 Wt = mt;   // the record keeps the sign word ready: slot s on bit 31 - s, row parity folded in
                double a1 = m1[ch][j] * alpha, a2 = m2[ch][j] * alpha;
                asm volatile("" : "+v"(a1), "+v"(a2));
                double nm1 = kMaxVal, nm2 = kMaxVal;
                u32 npos = 0, nS = 0, sy = 0;
                u32 nb = n8[ch];
                asm volatile("" : "+v"(nb));
                double r[RW];
                static_for<0, RW>([&](auto S) {
                    constexpr int s = decltype(S)::value;
                    r[s] = *reinterpret_cast<const double *>(ldsb + rot(nb, IC<C::SH[j][s]>{}) + C::COL[j][s] * (8 * M));
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
                failw |= ok[ch] ? sy : 0u;
                m1[ch][j] = nm1; m2[ch][j] = nm2; meta[ch][j] = ((nS ^ (0u - (__popc(nS) & 1u))) << (32 - RW)) | npos;
                asm volatile("" : "+v"(m1[ch][j]), "+v"(m2[ch][j]), "+v"(meta[ch][j]));
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("" ::: "memory");
        if (__ballot((failw >> 31) != 0) == 0ull) { res = iter + 1; break; }
    }

    if (lane == 0 && a.iters) a.iters[fr] = res;
    if (a.hard) {
        constexpr int HW = (N + 31) / 32;
        pack_hard<N, 64>(a.hard + fr * HW, lane, [&](int v) { return (*reinterpret_cast<const u32 *>(ldsb + (size_t)v * 8 + 4) >> 31) != 0; });
    }
    if (a.soft_out) {
        static_for<0, CH>([&](auto Q) {
            constexpr int ch = decltype(Q)::value;
            if (ok[ch]) {
                static_for<0, NH>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    a.soft_out[fr * N + k * M + 64 * ch + lane] = *reinterpret_cast<const double *>(ldsb + n8[ch] + k * (8 * M));
                });
            }
        });
    }
    });
}

// ---------------------------------------------------------------------------------------------------------------
// Layered offset min-sum (upstream lmin_sum_decod_qc_lm, decoders.cpp:5064-5425, active branch :5106-5290 with
// MY_VERSION; semantics SURVEY Appendix A.3), code-specialised like the flooding kernel above, for any lifting
// M <= 512: one frame per workgroup of W = ceil(M/64) wavefronts, a-posteriori values in LDS (8 B per variable,
// 128 KiB at (16384,8192)), check records in VGPRs.  Block rows (layers) are strictly sequential: one workgroup
// barrier per layer when W > 1; inside a layer every variable is touched by exactly one check, so the two passes
// of a layer (read v2c / write back) need no synchronisation.  beta = 0.4 (:5163); upstream's alpha/beta arguments
// are dead.  Bit-identical to the reference (hard bits, iteration counts, soft values).
// ---------------------------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void lms_body(const SpecArgs &a) {
    constexpr int RH = C::RH, NH = C::NH, M = C::M, N = NH * M, W = (M + 63) / 64;
    constexpr bool POW2 = (M & (M - 1)) == 0;
    constexpr double beta = 0.4;
    extern __shared__ double lds[];  // [N] soft, then one flag word
    char *const ldsb = reinterpret_cast<char *>(lds);
    int *const flag = reinterpret_cast<int *>(ldsb + (size_t)N * 8);
    const int n = threadIdx.x;       // check row inside a circulant == variable index inside a block column
    const bool valid = (M % 64 == 0) || n < M;
    const u32 n8 = (u32)(valid ? n : 0) * 8u;
    for_each_frame<true>(a, (long long)blockIdx.x, [&](const long long fr) {

    // byte offset inside a block column of variable (n + shift) mod M
    auto rot = [&](u32 base, auto S) -> u32 {
        constexpr int c = decltype(S)::value;
        if constexpr (c == 0) return base;
        else if constexpr (POW2) return (base + 8u * (u32)c) & (u32)(8 * M - 1);
        else { const u32 t = base + 8u * (u32)c, w = t - (u32)(8 * M); return t < w ? t : w; }   // unsigned min: t - 8M wraps when t < 8M (no VCC select)
    };
    FrameVote fvote;
    if constexpr (W > 1) fvote.init(flag);
    auto vote = [&](bool fail) -> bool {
        if constexpr (W == 1) return __ballot(fail) != 0ull;
        else return fvote(fail);
    };
    // check_syndrome (decoders.cpp:793-814) of the current soft values
    auto syndrome_fail = [&]() -> bool {
        u32 failw = 0;
        static_for<0, RH>([&](auto J) {
            constexpr int j = decltype(J)::value;
            u32 sy = 0, nb = n8;
            asm volatile("" : "+v"(nb));
            static_for<0, C::RW[j]>([&](auto S) {
                constexpr int s = decltype(S)::value;
                // only the sign dword of the 8-byte word.  Lanes n and n + 16 then share an LDS bank -- this pass is where the 19 % of
                // conflict cycles in this kernel's counters come from -- but the conflict is the cheap way to read every second dword:
                // fetching all 8 bytes (conflict free) moves twice the data and was 8 % slower end to end (41.8 -> 45.5 ms per 16384
                // frames of the M = 512 lifting, round 3).
                sy ^= *reinterpret_cast<const u32 *>(ldsb + rot(nb, IC<C::SH[j][s]>{}) + C::COL[j][s] * (8 * M) + 4);
            });
            failw |= sy;
        });
        return valid && (failw >> 31);
    };

    if (valid) {
        static_for<0, NH>([&](auto K) {   // :5088 soft = y
            constexpr int k = decltype(K)::value;
            *reinterpret_cast<double *>(ldsb + n8 + k * (8 * M)) = a.llr[fr * N + k * M + n] + 0.0;
        });
    }
    double m1[RH], m2[RH];
    u32 meta[RH];
    static_for<0, RH>([&](auto J) { constexpr int j = decltype(J)::value; m1[j] = 0.0; m2[j] = 0.0; meta[j] = 0u; });
    if constexpr (W > 1) __syncthreads();

    int res = -a.maxiter;                                  // :5424 when the loop runs dry
    bool fail = vote(syndrome_fail());                     // :5111-5115
    if (!fail) res = 1;                                    // :5119 at iter 0
    for (int iter = 0; fail && iter < a.maxiter; ++iter) {
        static_for<0, RH>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int RW = C::RW[j];
            u32 mt = meta[j], nb = n8;
            asm volatile("" : "+v"(mt), "+v"(nb));         // anchor: keeps this layer's ALU work behind the barrier
            const u32 pos = mt & 0xffu;
            u32 Wt = mt;   // the record keeps the sign word ready: slot s on bit 31 - s, row parity folded in
            double nm1 = kMaxVal, nm2 = kMaxVal;           // :5133-5134 (no clamp of the v2c magnitudes in this decoder)
            u32 npos = 0, nS = 0;
            double r[RW], tv[RW];
            static_for<0, RW>([&](auto S) {                // :5141-5177
                constexpr int s = decltype(S)::value;
                r[s] = *reinterpret_cast<const double *>(ldsb + rot(nb, IC<C::SH[j][s]>{}) + C::COL[j][s] * (8 * M));
            });
            static_for<0, RW>([&](auto S) {
                constexpr int s = decltype(S)::value;
                const double aa = sel64(m1[j], m2[j], lanes_eq(pos, (u32)s));
                const double pc = signed_mag(aa, Wt);       // previous c2v (no alpha, no beta here)
                Wt = twice(Wt);
                const double tt = r[s] - pc;
                tv[s] = tt;
                nS = __builtin_amdgcn_alignbit(nS, hi32(tt), 31);   // sign kept even when the magnitude clips to 0 (:5164-5168)
                double mag = fabs(tt) - beta;
                mag = at_least(mag, 0.0);                     // :5166-5167 `if (mag < 0) mag = 0`
                const mask64 c1 = lanes_lt(mag, nm1);       // process_check_node :5012-5027
                nm2 = fmin(fmax(mag, nm1), nm2);
                npos = sel32(npos, (u32)s, c1);
                nm1 = fmin(mag, nm1);
            });
            const u32 W0 = (nS ^ (0u - (__popc(nS) & 1u))) << (32 - RW);
            u32 Wn = W0;
            static_for<0, RW>([&](auto S) {                // :5182-5206
                constexpr int s = decltype(S)::value;
                const double aa = sel64(nm1, nm2, lanes_eq(npos, (u32)s));
                const double cv = signed_mag(aa, Wn);
                Wn = twice(Wn);
                if (valid) *reinterpret_cast<double *>(ldsb + rot(nb, IC<C::SH[j][s]>{}) + C::COL[j][s] * (8 * M)) = tv[s] + cv;
            });
            m1[j] = nm1; m2[j] = nm2; meta[j] = W0 | npos;
            if constexpr (W > 1) __syncthreads();           // the next layer reads what this one wrote
            else __builtin_amdgcn_sched_barrier(0);
        });
        fail = vote(syndrome_fail());                       // :5281-5284
        if (!fail) res = iter + 1;                          // :5287
    }

    if (threadIdx.x == 0 && a.iters) a.iters[fr] = res;
    if (a.hard) {
        if constexpr (M % 64 == 0) {
            // this wave's 64 variables of block column k are the packed words 2*(k*W + wave) and +1
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            static_for<0, NH>([&](auto K) {
                constexpr int k = decltype(K)::value;
                const u64 b = __ballot((*reinterpret_cast<const u32 *>(ldsb + n8 + k * (8 * M) + 4) >> 31) != 0);
                if (lane == 0) reinterpret_cast<u64 *>(a.hard + fr * (N / 32))[k * W + wave] = b;
            });
        } else {
            constexpr int HW = (N + 31) / 32;
            pack_hard<N, W * 64>(a.hard + fr * HW, threadIdx.x, [&](int v) { return (*reinterpret_cast<const u32 *>(ldsb + (size_t)v * 8 + 4) >> 31) != 0; });
        }
    }
    if (a.soft_out && valid) {
        static_for<0, NH>([&](auto K) {
            constexpr int k = decltype(K)::value;
            a.soft_out[fr * N + k * M + n] = *reinterpret_cast<const double *>(ldsb + n8 + k * (8 * M));
        });
    }
    });
}

// ---------------------------------------------------------------------------------------------------------------
// Layered offset min-sum for SMALL liftings (M <= 32): floor(64/M) frames per wavefront, lane = f*M + n, the layers of all
// frames in lockstep (see ms_small_body).  Upstream checks the syndrome once per iteration and stops a frame there; a
// frame that has stopped freezes (one divergent region per iteration) while the others continue.
// ---------------------------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void lms_small_body(const SpecArgs &a) {
    constexpr int RH = C::RH, NH = C::NH, M = C::M, N = NH * M, F = 64 / M;
    static_assert(M <= 32 && F >= 2, "lms_small_body: M <= 32");
    constexpr bool POW2 = (M & (M - 1)) == 0;
    constexpr double beta = 0.4;     // decoders.cpp:5163 (the alpha / beta arguments are dead upstream)
    extern __shared__ double lds[];  // [F][N] soft
    const int lane = threadIdx.x;
    const int f = lane / M, n = lane - f * M;
    const long long fr = (long long)blockIdx.x * F + f;
    const bool live = f < F && fr < a.nframes;
    char *const ldsb = reinterpret_cast<char *>(lds) + (size_t)(live ? f : 0) * N * 8;
    const u32 n8 = (u32)n * 8u;
    const u64 frame_lanes = (M == 32 ? 0xffffffffull : ((1ull << M) - 1ull)) << ((live ? f : 0) * M);

    auto rot = [&](u32 base, auto S) -> u32 {
        constexpr int c = decltype(S)::value;
        if constexpr (c == 0) return base;
        else if constexpr (POW2) return (base + 8u * (u32)c) & (u32)(8 * M - 1);
        else { const u32 t = base + 8u * (u32)c, w = t - (u32)(8 * M); return t < w ? t : w; }
    };
    auto syndrome_word = [&]() -> u32 {        // check_syndrome (decoders.cpp:793-814): sign bit = this check fails
        u32 failw = 0;
        static_for<0, RH>([&](auto J) {
            constexpr int j = decltype(J)::value;
            u32 sy = 0, nb = n8;
            asm volatile("" : "+v"(nb));
            static_for<0, C::RW[j]>([&](auto S) {
                constexpr int s = decltype(S)::value;
                sy ^= *reinterpret_cast<const u32 *>(ldsb + rot(nb, IC<C::SH[j][s]>{}) + C::COL[j][s] * (8 * M) + 4);
            });
            failw |= sy;
        });
        return failw;
    };

    if (live) {
        static_for<0, NH>([&](auto K) {   // :5088 soft = y
            constexpr int k = decltype(K)::value;
            *reinterpret_cast<double *>(ldsb + n8 + k * (8 * M)) = a.llr[fr * N + k * M + n] + 0.0;
        });
    }
    double m1[RH], m2[RH];
    u32 meta[RH];
    static_for<0, RH>([&](auto J) { constexpr int j = decltype(J)::value; m1[j] = 0.0; m2[j] = 0.0; meta[j] = 0u; });
    asm volatile("" ::: "memory");

    int res = -a.maxiter;                                  // :5424 when the loop runs dry
    bool done = !live;
    {
        const u64 failing = __ballot(!done && (syndrome_word() >> 31) != 0);   // :5111-5115
        if (!done && (failing & frame_lanes) == 0ull) { done = true; res = 1; } // :5119 at iter 0
    }
    for (int iter = 0; iter < a.maxiter; ++iter) {
        if (__ballot(!done) == 0ull) break;
        if (!done) {
            static_for<0, RH>([&](auto J) {
                constexpr int j = decltype(J)::value;
                constexpr int RW = C::RW[j];
                u32 mt = meta[j], nb = n8;
                asm volatile("" : "+v"(mt), "+v"(nb));
                const u32 pos = mt & 0xffu;
                u32 Wt = mt;   // the record keeps the sign word ready: slot s on bit 31 - s, row parity folded in
                double nm1 = kMaxVal, nm2 = kMaxVal;       // :5133-5134
                u32 npos = 0, nS = 0;
                double r[RW], tv[RW];
                static_for<0, RW>([&](auto S) {            // :5141-5177
                    constexpr int s = decltype(S)::value;
                    r[s] = *reinterpret_cast<const double *>(ldsb + rot(nb, IC<C::SH[j][s]>{}) + C::COL[j][s] * (8 * M));
                });
                static_for<0, RW>([&](auto S) {
                    constexpr int s = decltype(S)::value;
                    const double aa = sel64(m1[j], m2[j], lanes_eq(pos, (u32)s));
                    const double pc = signed_mag(aa, Wt);
                    Wt = twice(Wt);
                    const double tt = r[s] - pc;
                    tv[s] = tt;
                    nS = __builtin_amdgcn_alignbit(nS, hi32(tt), 31);
                    double mag = fabs(tt) - beta;
                    mag = at_least(mag, 0.0);              // :5166-5167
                    const mask64 c1 = lanes_lt(mag, nm1);  // process_check_node :5012-5027
                    nm2 = fmin(fmax(mag, nm1), nm2);
                    npos = sel32(npos, (u32)s, c1);
                    nm1 = fmin(mag, nm1);
                });
                const u32 W0 = (nS ^ (0u - (__popc(nS) & 1u))) << (32 - RW);
            u32 Wn = W0;
                static_for<0, RW>([&](auto S) {            // :5182-5206
                    constexpr int s = decltype(S)::value;
                    const double aa = sel64(nm1, nm2, lanes_eq(npos, (u32)s));
                    const double cv = signed_mag(aa, Wn);
                    Wn = twice(Wn);
                    *reinterpret_cast<double *>(ldsb + rot(nb, IC<C::SH[j][s]>{}) + C::COL[j][s] * (8 * M)) = tv[s] + cv;
                });
                m1[j] = nm1; m2[j] = nm2; meta[j] = W0 | npos;
                asm volatile("" ::: "memory");             // the next layer reads what this one wrote (LDS runs a wave's accesses in order)
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        asm volatile("" ::: "memory");
        const u64 failing = __ballot(!done && (syndrome_word() >> 31) != 0);   // :5281-5284
        if (!done && (failing & frame_lanes) == 0ull) { done = true; res = iter + 1; }   // :5287
    }

    if (!live) return;
    if (n == 0 && a.iters) a.iters[fr] = res;
    if (a.hard) {
        constexpr int HW = (N + 31) / 32;
        for (int w = n; w < HW; w += M) {
            u32 bits = 0;
            for (int b = 0; b < 32; ++b) {
                const int v = 32 * w + b;
                if (v < N) bits |= (*reinterpret_cast<const u32 *>(ldsb + (size_t)v * 8 + 4) >> 31) << b;
            }
            a.hard[fr * HW + w] = bits;
        }
    }
    if (a.soft_out) {
        static_for<0, NH>([&](auto K) {
            constexpr int k = decltype(K)::value;
            a.soft_out[fr * N + k * M + n] = *reinterpret_cast<const double *>(ldsb + n8 + k * (8 * M));
        });
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Likelihood-ratio sum-product (upstream sum_prod_decod_qc_lm, decoders.cpp:1923-2185; semantics SURVEY Appendix
// A.2), code-specialised, for liftings that are a multiple of 64.  One frame per workgroup of SPW = 8 wavefronts.
// The expensive operations are the three fp64 divisions per edge and iteration; the block columns (x 64-lane chunks)
// are dealt to the 8 waves at compile time, heaviest first to the least loaded wave, so every wave carries the same
// number of edges whatever the column weights are (this code has weights 2, 3 and 14).  Phases per iteration:
//   A  (column units)  AA_u = yd * prod_{other edges of the column, rows ascending} ZZ ; ZZ_u <- (AA_u-1)/(AA_u+1)
//                      (:2017-2060) -- the prefix of the product is carried in a register and the tail read from the
//                      not yet overwritten entries, i.e. exactly the reference's multiplication order, in place
//   B  (row units)     s = 1.0 * prod_{edges of the row, columns ascending} ZZ[e][(n+c) mod M]            (:2047-2050)
//   C  (column units)  A = s[(t-c) mod M] / ZZ ; A = (1+A)/(1-A) ; clamp [-5.2e-9 (sic), 1.9e8] ; ZZ <- A ; soft *= A
//   D  (row units)     syndrome = xor over the row's edges of (soft < 1.0)                                (:2129-2149)
// LDS: ZZ[E][M] fp64 + s[R] fp64 + one hard-decision byte per variable (66 KiB at (2048,1024): two frames per CU);
// yd and soft of a wave's own columns stay in VGPRs.  exp() is exp_glibc (the reference's own algorithm, bit for bit), every
// other operation is exact and in the reference's order: hard decisions, iteration counts AND the a-posteriori ratios are
// identical to the CPU reference's.
// ---------------------------------------------------------------------------------------------------------------
// a / b for operands whose range is known: the instruction sequence the compiler emits for an fp64 division is
//   v_div_scale (x2), v_rcp_f64, two Newton steps, q = a*r, e = fma(-b, q, a), v_div_fmas, v_div_fixup;
// v_div_scale / v_div_fmas / v_div_fixup only act when the denominator is zero, an operand is infinite / NaN / denormal, the
// numerator is below 2^-969, or the exponents differ by 768 or more (ISA: V_DIV_SCALE_F64) -- otherwise they pass their input
// through and the result is fma(e, r, q).  The probability-domain decoders divide quantities that are bounded away from all of
// that by construction: messages are clamped to [1e-4, 1 - 1e-4] (TDMP) resp. [1e-6, 1 - 1e-6] (ASP), channel priors lie in
// [4.2e-18, 1 - 4.2e-18] (LLR / 2 clamped to +-20), so a posterior or extrinsic probability is >= 4.2e-18 * clamp^(column weight):
// >= 4e-274 for TDMP (column weight <= block rows <= 64) and >= 4e-258 for ASP with column weight <= 40 -- both far above
// 2^-969 = 2e-292 -- and every denominator is >= the clamp bound (e.g. aa + x - 2 aa x >= min(aa, 1 - aa)) and <= 1e7.  A numerator
// may be exactly 0 (1 - sov): 0 * r = 0, e = 0, result +0 as with the full sequence.  So the three instructions are dropped: the
// SAME eight remaining instructions, hence the same bits, 27 % fewer instructions per division.  ASP columns heavier than 40
// keep the compiler's division (div_if).
__device__ __forceinline__ double div_ranged(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    double f = __fma_rn(-b, r, 1.0);
    r = __fma_rn(r, f, r);
    f = __fma_rn(-b, r, 1.0);
    r = __fma_rn(r, f, r);
    const double q = a * r;
    const double e = __fma_rn(-b, q, a);
    return __fma_rn(e, r, q);
}
template <bool RANGED>
__device__ __forceinline__ double div_if(double a, double b) {
    if constexpr (RANGED) return div_ranged(a, b);
    else return a / b;
}

#ifndef LDPC_SP_WAVES
#define LDPC_SP_WAVES 8           // (macros: tools/ab_sp.hip builds the variants)
#endif
#ifndef LDPC_SP_BODY_WAVES
#define LDPC_SP_BODY_WAVES 8
#endif
constexpr int kSpWaves = LDPC_SP_WAVES;       // wavefronts per frame of the asp / bp bodies (two frames per CU at 4 waves per SIMD, <= 128 VGPRs)
constexpr int kSpBodyWaves = LDPC_SP_BODY_WAVES;   // sp_body: eight too since round 3.  In round 2 four waves per frame at 2 waves per SIMD (251 VGPRs, no spills) beat eight
                                  // with 101 spilled registers by 9-13 % (profiles/r02_sp_variants.txt).  The registers were rotated LDS addresses of the
                                  // unrolled phases, hoisted out of the iteration loop; with the lane id laundered once per iteration (see the loops) eight
                                  // waves need 126 registers, spill nothing and are 3.5 % (0 dB) to 9 % (2 dB) faster than four (profiles/r03_sp_family_ab.txt)

template <class C, int WAVES = kSpWaves>
struct SpView {  // column view of the code + static work split, all computed at compile time
    static constexpr int CH = (C::M + 63) / 64;          // 64-lane chunks per circulant (the last one may be partly idle)
    int row_off[C::RH + 1] = {};                         // row-major edge id of (row j, slot 0)
    int cw[C::NH] = {};                                  // column weights
    int ce[C::NH][C::RH] = {};                           // edge ids of a column, rows ascending
    int cj[C::NH][C::RH] = {};                           // their block rows
    int cc[C::NH][C::RH] = {};                           // their shifts
    int col_wave[C::NH * ((C::M + 63) / 64)] = {};       // wave that owns unit (k, chunk)
    int col_slot[C::NH * ((C::M + 63) / 64)] = {};       // index of the unit inside its wave's list
    int units_max = 0;                                   // max units per wave
    int ne = 0;
    constexpr SpView() {
        for (int j = 0; j < C::RH; ++j) { row_off[j] = ne; ne += C::RW[j]; }
        row_off[C::RH] = ne;
        for (int j = 0; j < C::RH; ++j)
            for (int s = 0; s < C::RW[j]; ++s) {
                const int k = C::COL[j][s];
                ce[k][cw[k]] = row_off[j] + s; cj[k][cw[k]] = j; cc[k][cw[k]] = C::SH[j][s];
                ++cw[k];
            }
        int load[WAVES] = {}, cnt[WAVES] = {};
        bool used[C::NH * ((C::M + 63) / 64)] = {};
        for (int it = 0; it < C::NH * CH; ++it) {         // heaviest remaining unit -> least loaded wave
            int best = -1;
            for (int u = 0; u < C::NH * CH; ++u)
                if (!used[u] && (best < 0 || cw[u / CH] > cw[best / CH])) best = u;
            int w = 0;
            for (int x = 1; x < WAVES; ++x) if (load[x] < load[w]) w = x;
            used[best] = true; col_wave[best] = w; col_slot[best] = cnt[w]++; load[w] += cw[best / CH];
        }
        for (int x = 0; x < WAVES; ++x) if (cnt[x] > units_max) units_max = cnt[x];
    }
};

template <class C>
__device__ __forceinline__ void sp_body(const SpecArgs &a) {
    constexpr SpView<C, kSpBodyWaves> V{};
    constexpr int RH = C::RH, NH = C::NH, M = C::M, N = NH * M, R = RH * M, CH = (M + 63) / 64, T = kSpBodyWaves * 64;
    constexpr int NE = V.ne, UMAX = V.units_max;
    // The three divisions per edge and iteration without the scaling / fix-up instructions (div_ranged) for row weights <= 12.  Ranges:
    // messages after the clamp of :2120 lie in [-5.2e-9, 1.9e8], positive ones are >= 2^-54 (they are (1+a)/(1-a) of an a >= -1 + ulp);
    // AA = exp(llr) * (product of <= RH of them) is finite, |AA - 1| is 0 or >= 2^-53, |AA + 1| likewise, so z = (AA-1)/(AA+1) has
    // 2^-54 <= |z| <= 2^54 or is 0; a row product s of <= 12 of them stays inside 2^+-650 -- no operand is denormal, no numerator below
    // 2^-969, no exponent difference >= 768.  What is left are exact zeros and the poles: 0 / 0 (z = 0 is a factor of its own row
    // product) is NaN either way; x / 0 with x != 0 is +-inf with the full sequence and NaN without it, and so is what follows from it --
    // but every such value ends in the clamp of :2120, where mind() / maxd() send NaN and +inf alike to 1.9e8 and -inf cannot arise
    // from (1+A)/(1-A) without passing through NaN first.  Heavier rows keep the compiler's division.
    constexpr bool RD = C::WMAX <= 12;
    extern __shared__ double lds[];
    char *const zzb = reinterpret_cast<char *>(lds);                        // ZZ[e][t] at e*M*8 + t*8
    char *const sb = zzb + (size_t)NE * M * 8;                              // s[j][n]
    unsigned char *const hb = reinterpret_cast<unsigned char *>(sb + (size_t)R * 8);  // [N]
    int *const flag = reinterpret_cast<int *>(hb + ((N + 15) & ~15));
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // lanes beyond the lifting in the last 64-lane chunk of a circulant sit out
    auto lane_ok = [&](auto CHI) { constexpr int c = decltype(CHI)::value; return (c * 64 + 64 <= M) || (c * 64 + lane < M); };
    const long long fr = blockIdx.x;

    FrameVote fvote;
    fvote.init(flag);
    auto vote = [&](bool fail) -> bool { return fvote(fail); };
    // rows (x chunks) are dealt round-robin to the waves: unit (j, ch) -> wave (j*CH + ch) % kSpBodyWaves
    auto syndrome_fail = [&]() -> bool {
        bool f = false;
        static_for<0, RH * CH>([&](auto U) {
            constexpr int u = decltype(U)::value, j = u / CH, ch = u % CH;
            if (wave == u % kSpBodyWaves && lane_ok(IC<ch>{})) {
                const int n = ch * 64 + lane;
                unsigned sy = 0;
                static_for<0, C::RW[j]>([&](auto S) {
                    constexpr int s = decltype(S)::value;
                    int t = n + C::SH[j][s]; if (t >= M) t -= M;
                    sy ^= hb[C::COL[j][s] * M + t];
                });
                f |= sy != 0;
            }
        });
        return f;
    };

    double yd[UMAX], sf[UMAX];
    static_for<0, UMAX>([&](auto Q) { yd[decltype(Q)::value] = 1.0; sf[decltype(Q)::value] = 1.0; });
    static_for<0, NH * CH>([&](auto U) {
        constexpr int u = decltype(U)::value, k = u / CH, ch = u % CH, q = V.col_slot[u];
        if (wave == V.col_wave[u] && lane_ok(IC<ch>{})) {
            const int t = ch * 64 + lane;
            const double yl = at_least(at_most(a.llr[fr * N + k * M + t], 20.0), -20.0);   // :1949 INPUT_LIMIT
            yd[q] = sf[q] = exp_glibc(yl);
            hb[k * M + t] = yd[q] < 1.0;
            static_for<0, V.cw[k]>([&](auto X) {                                     // :1957-1959
                *reinterpret_cast<double *>(zzb + (size_t)V.ce[k][decltype(X)::value] * M * 8 + t * 8) = 1.0;
            });
        }
    });
    __syncthreads();

    int res = -a.maxiter;
    bool conv = !vote(syndrome_fail());                                              // :1964-2002
    if (conv) res = 0;
    for (int iter = 0; !conv && iter < a.maxiter; ++iter) {
        asm volatile("" : "+v"(lane));   // per iteration: keeps the rotated LDS addresses of the unrolled phases out of the loop-invariant set (registers)
        // ---- phase A
        static_for<0, NH * CH>([&](auto U) {
            constexpr int u = decltype(U)::value, k = u / CH, ch = u % CH, q = V.col_slot[u], CW = V.cw[k];
            if (wave == V.col_wave[u] && lane_ok(IC<ch>{})) {
                const int t8 = (ch * 64 + lane) * 8;
                double zo[CW];
                static_for<0, CW>([&](auto X) {
                    constexpr int x = decltype(X)::value;
                    zo[x] = *reinterpret_cast<const double *>(zzb + (size_t)V.ce[k][x] * M * 8 + t8);
                });
                double prefix = yd[q];
                static_for<0, CW>([&](auto X) {
                    constexpr int x = decltype(X)::value;
                    double AA = prefix;                                               // :2027-2041, ascending rows
                    static_for<x + 1, CW>([&](auto W2) { AA *= zo[decltype(W2)::value]; });
                    *reinterpret_cast<double *>(zzb + (size_t)V.ce[k][x] * M * 8 + t8) = div_if<RD>(AA - 1, AA + 1);  // :2044
                    prefix *= zo[x];
                });
            }
        });
        __syncthreads();
        // ---- phase B
        static_for<0, RH * CH>([&](auto U) {
            constexpr int u = decltype(U)::value, j = u / CH, ch = u % CH;
            if (wave == u % kSpBodyWaves && lane_ok(IC<ch>{})) {
                const int n = ch * 64 + lane;
                double s = 1.0;                                                       // :2010
                static_for<0, C::RW[j]>([&](auto S) {
                    constexpr int sl = decltype(S)::value;
                    int t = n + C::SH[j][sl]; if (t >= M) t -= M;
                    s *= *reinterpret_cast<const double *>(zzb + (size_t)(V.row_off[j] + sl) * M * 8 + t * 8);  // :2047-2050
                });
                *reinterpret_cast<double *>(sb + (size_t)(j * M + n) * 8) = s;
            }
        });
        __syncthreads();
        // ---- phase C
        static_for<0, NH * CH>([&](auto U) {
            constexpr int u = decltype(U)::value, k = u / CH, ch = u % CH, q = V.col_slot[u], CW = V.cw[k];
            if (wave == V.col_wave[u] && lane_ok(IC<ch>{})) {
                const int t = ch * 64 + lane;
                double soft = yd[q];                                                  // :2011
                static_for<0, CW>([&](auto X) {
                    constexpr int x = decltype(X)::value;
                    int nn = t - V.cc[k][x]; if (nn < 0) nn += M;                    // rotate by M-circ (:2113)
                    double *zp = reinterpret_cast<double *>(zzb + (size_t)V.ce[k][x] * M * 8 + t * 8);
                    double A = div_if<RD>(*reinterpret_cast<const double *>(sb + (size_t)(V.cj[k][x] * M + nn) * 8), *zp);
                    A = div_if<RD>(1 + A, 1 - A);
                    A = at_least(at_most(A, 1.9e+8), -5.2e-9);                        // :2120 (negative lower clamp is upstream's)
                    *zp = A;
                    soft *= A;
                });
                sf[q] = soft;
                hb[k * M + t] = soft < 1.0;
            }
        });
        __syncthreads();
        // ---- phase D
        if (!vote(syndrome_fail())) { conv = true; res = iter + 1; }                 // :2151-2166
    }

    if (threadIdx.x == 0 && a.iters) a.iters[fr] = res;
    if (a.hard) {
        pack_hard<N, T>(a.hard + fr * ((N + 31) / 32), threadIdx.x, [&](int v) { return hb[v] != 0; });
    }
    if (a.soft_out) {
        static_for<0, NH * CH>([&](auto U) {
            constexpr int u = decltype(U)::value, k = u / CH, ch = u % CH, q = V.col_slot[u];
            if (wave == V.col_wave[u] && lane_ok(IC<ch>{})) a.soft_out[fr * N + k * M + ch * 64 + lane] = sf[q];
        });
    }
}

// ---------------------------------------------------------------------------------------------------------------
// TDMP ("turbo-decoding message passing") sum-product in the probability domain -- upstream
// tdmp_sum_prod_gf2_decod_qc_lm (decoders.cpp:2584-2744, map_bin :2191-2228), decoder id 7, the decoder_type of every
// shipped scenario file.  Layered like lms_body: block rows are sequential, one frame per workgroup, a-posteriori probabilities
// P(bit=1) in LDS.  The per-edge state Z (one fp64 per edge and check, which the reference keeps in an R x max-row-weight matrix)
// lives in VGPRs.  Two fp64 divisions per edge and iteration.  The channel transform uses exp_glibc (the reference's exp(), bit for
// bit), so the probabilities, hard decisions and step counts are identical to the CPU reference's.  As upstream, the result is
// always the hard decision (`decision` is dead, :2737).
//
// TWO lanes per check (round 3).  Rounds 1-2 kept a check's whole Z row in one lane: 224 registers for the example code, one wave
// per SIMD, nothing to hide the division chains behind (7 cycles per instruction).  Here two lanes (l and l + 32 of a wave) share
// check n of the current block row: lane A owns the row's first ceil(RW/2) edges in ascending order, lane B the others in DESCENDING
// order (an odd row pads B with a factor 1.0).  In that order map_bin's two running products are the same code on both lanes:
//   C1[i] = e[i] * C1[i-1]       A: SF[0..] left to right            B: SB[RW-1..] right to left          (:2209-2216)
//   X     = the partner's last C1 (one swap of the wave's halves: SB[LA] for A, SF[LA-1] for B)
//   C2[i] = e[i] * C2[i+1], C2[L] = X     A: SB[LA-1..1]             B: SF[LA..RW-2]
//   q[i]  = (1 - C1[i-1] * C2[i+1]) / 2   (i = 0: 1 - C2[1])         every product has upstream's operands, so upstream's bits
// Half the state per lane (128 registers) and the LDS address of every edge kept in registers, two to a VGPR (they differ
// between the lanes of a pair, so they cannot be immediates): 242 registers, two waves per SIMD, and the layers of two frames
// overlap on every SIMD: 19-23 % less time than one lane per check (profiles/r03_tdmp_pair_ab.txt).  Liftings up to 256.
// LDS: [N] a-posteriori probabilities, one spare slot per thread (what padded edges and idle lanes read and write), flag words.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 swap_halves(u32 v, bool upper) {   // lane l <-> lane l ^ 32 (gfx950: v_permlane32_swap_b32 + one select)
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // r[0]: the lower half's values in both halves, r[1]: the upper half's
    return upper ? r[0] : r[1];
}
__device__ __forceinline__ double swap_halves(double v, bool upper) { return mk(swap_halves(hi32(v), upper), swap_halves(lo32(v), upper)); }

template <class C>
__device__ __forceinline__ void tasp_body(const SpecArgs &a) {
    constexpr int RH = C::RH, NH = C::NH, M = C::M, N = NH * M, TH = ((2 * M + 63) / 64) * 64, LMAX = (C::WMAX + 1) / 2;
    static_assert(RH <= 64, "tasp_body: div_ranged's lower bound on the probabilities assumes column weights <= 64");
    // the packed addresses are byte addresses while the LDS image fits 64 KB, else 8-byte word indices (one more shift per access)
    constexpr bool WIDE = (size_t)N * 8 + (size_t)TH * 8 + 16 > 65536;
    constexpr int AS = WIDE ? 3 : 0;
    static_assert(((size_t)N * 8 + (size_t)TH * 8 + 16) >> AS <= 65536, "tasp_body: 16-bit packed LDS addresses");
    constexpr double T = 0.0001, TT = 0;                                   // :2597-2598
    extern __shared__ double lds[];
    char *const ldsb = reinterpret_cast<char *>(lds);
    int *const flag = reinterpret_cast<int *>(ldsb + (size_t)N * 8 + (size_t)TH * 8);
    // lanes 0-31 of a wave are the A halves of 32 consecutive checks, lanes 32-63 the B halves of the same checks: every half-wave
    // then reads and writes 32 consecutive doubles of ONE block column (A and B lanes interleaved hit two columns per half-wave and
    // conflict in about half of the banks: 454 M conflict cycles per launch of the shipped scenario against 70 M)
    const int t = threadIdx.x, n = (t >> 6) * 32 + (t & 31);
    const bool isB = (t & 32) != 0, valid = n < M;
    const u32 spare = (u32)N * 8u + (u32)t * 8u;
    const long long fr = blockIdx.x;

    // LDS byte address of local edge k of block row j for this lane, two to a register
    u32 pk[RH][(LMAX + 1) / 2];
    static_for<0, RH>([&](auto J) {
        constexpr int j = decltype(J)::value, RW = C::RW[j], L = (RW + 1) / 2, LB = RW / 2;
        static_assert(RW >= 2, "tasp_body: map_bin needs at least two edges per check");
        static_for<0, (LMAX + 1) / 2>([&](auto H) { pk[j][decltype(H)::value] = (spare >> AS) | ((spare >> AS) << 16); });
        static_for<0, L>([&](auto K) {
            constexpr int k = decltype(K)::value, sA = k, sB = k < LB ? RW - 1 - k : 0;
            int nA = n + C::SH[j][sA]; if (nA >= M) nA -= M;
            int nB = n + C::SH[j][sB]; if (nB >= M) nB -= M;
            const u32 adA = (u32)(C::COL[j][sA] * M + nA) * 8u, adB = k < LB ? (u32)(C::COL[j][sB] * M + nB) * 8u : spare;
            const u32 ad = (valid ? (isB ? adB : adA) : spare) >> AS;
            if constexpr (k % 2 == 0) pk[j][k / 2] = (pk[j][k / 2] & 0xffff0000u) | ad;
            else pk[j][k / 2] = (pk[j][k / 2] & 0x0000ffffu) | (ad << 16);
        });
    });
    auto adr = [&](auto J, auto K) -> u32 {
        constexpr int j = decltype(J)::value, k = decltype(K)::value;
        if constexpr (k % 2 == 0) return (pk[j][k / 2] & 0xffffu) << AS;
        else return (pk[j][k / 2] >> 16) << AS;
    };
    // true for the one padded slot of an odd row on the B lane (and only there)
    auto padded = [&](auto J, auto K) -> bool {
        constexpr int j = decltype(J)::value, k = decltype(K)::value;
        if constexpr (k >= C::RW[j] / 2) return isB; else return false;
    };

    FrameVote fvote;
    fvote.init(flag);
    auto syndrome_fail = [&]() -> bool {                                    // check_syndrome_thr :2274-2306, thr 0.5
        bool f = false;
        static_for<0, RH>([&](auto J) {
            constexpr int j = decltype(J)::value, L = (C::RW[j] + 1) / 2;
            u32 sy = 0;
            static_for<0, L>([&](auto K) {
                const bool one = *reinterpret_cast<const double *>(ldsb + adr(J, K)) > 0.5;
                sy ^= (u32)(one && !padded(J, K));
            });
            sy ^= swap_halves(sy, isB);
            f |= sy != 0;
        });
        return valid && f;
    };

    for (int v = t; v < N; v += TH) {                                       // :2611-2618
        const double x = a.llr[fr * N + v] * 0.5;
        const double y = at_least(at_most(x, 20.0), -20.0);
        const double e0 = exp_glibc(y), e1 = exp_glibc(-y);
        lds[v] = e1 / (e0 + e1);
    }
    *reinterpret_cast<double *>(ldsb + spare) = 0.5;
    double Z[RH][LMAX];
    static_for<0, RH>([&](auto J) {
        static_for<0, (C::RW[decltype(J)::value] + 1) / 2>([&](auto K) { Z[decltype(J)::value][decltype(K)::value] = 0.5; });  // :2637
    });
    __syncthreads();

    int res = 0;
    bool fail = fvote(syndrome_fail());                                     // :2653-2660: already a codeword -> 0
    int steps = 0;
    while (fail && steps < a.maxiter) {
        static_for<0, RH>([&](auto J) {
            constexpr int j = decltype(J)::value, RW = C::RW[j], L = (RW + 1) / 2;
            static_for<0, (L + 1) / 2>([&](auto H) { u32 w = pk[j][decltype(H)::value]; asm volatile("" : "+v"(w)); pk[j][decltype(H)::value] = w; });   // the unpacked addresses are per-layer values, not 64 more live registers
            double y[L], e[L], C1[L], C2[L + 1], q[L];
            static_for<0, L>([&](auto K) {
                constexpr int k = decltype(K)::value;
                const double x = *reinterpret_cast<const double *>(ldsb + adr(J, K));
                const double aa = Z[j][k];
                double v = div_ranged(x * (1.0 - aa), aa + x - 2.0 * aa * x);   // :2686 rho = gamma - lambda; the denominator is >= min(aa, 1 - aa) >= 1e-4
                v = at_most(at_least(v, TT), 1 - TT);                        // :2694-2695
                y[k] = v;
                const double p = 1 - 2 * v;                                  // map_bin :2206
                e[k] = padded(J, K) ? 1.0 : p;
            });
            C1[0] = e[0];
            static_for<1, L>([&](auto I) { constexpr int i = decltype(I)::value; C1[i] = e[i] * C1[i - 1]; });
            C2[L] = swap_halves(C1[L - 1], isB);
            static_for<0, L - 1>([&](auto I) { constexpr int i = L - 1 - decltype(I)::value; C2[i] = e[i] * C2[i + 1]; });
            q[0] = (1 - C2[1]) / 2;                                          // :2219-2227
            static_for<1, L>([&](auto I) { constexpr int i = decltype(I)::value; q[i] = (1 - C1[i - 1] * C2[i + 1]) / 2; });
            static_for<0, L>([&](auto K) {
                constexpr int k = decltype(K)::value;
                double v = q[k];
                v = at_most(at_least(v, T), 1.0 - T);                        // :2703-2704
                Z[j][k] = v;
                const double g = div_ranged(y[k] * v, 1.0 - y[k] - v + 2 * y[k] * v);  // :2716 gamma = rho + lambda; denominator >= min(v, 1 - v) >= 1e-4
                *reinterpret_cast<double *>(ldsb + adr(J, K)) = g;
            });
            __syncthreads();
        });
        fail = fvote(syndrome_fail());                                       // :2723 (only the value after the last layer counts)
        steps = steps + 1;
    }
    res = fail ? -steps : steps;                                             // :2740-2743 (0 when the input was a codeword)

    if (threadIdx.x == 0 && a.iters) a.iters[fr] = res;
    if (a.hard) {
        constexpr int HW = (N + 31) / 32;
        pack_hard<N, TH>(a.hard + fr * HW, threadIdx.x, [&](int v) { return lds[v] > 0.5; });
    }
    if (a.soft_out) {
        for (int v = t; v < N; v += TH) a.soft_out[fr * N + v] = lds[v];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Flooding sum-product in the probability domain ("advanced sum-product", upstream sum_prod_gf2_decod_qc_lm,
// decoders.cpp:2324-2581, decoder id 2; general branch :2482-2556 -- codes whose block columns all have weight 2
// take a different upstream branch and are rejected by the host).  Same work split as sp_body: 8 waves per frame,
// block rows dealt round-robin, block columns dealt at compile time so that every wave carries the same number of
// edges.  Per-edge state (one fp64 per edge and CHECK position, upstream's state[slot][check]) in LDS.
//   1 (row units)     map_bin over the row's edges at check n (:2191-2228): products only
//   2 (column units)  P1 = p * prod d, P0 = (1-p) * prod (1-d) over the column's edges, rows ascending, d read at
//                     check (t - c) mod M; soft_out = P1/(P0+P1); then per edge p1 = so/d, p0 = (1-so)/(1-d),
//                     state <- clamp(p1/(p1+p0), 1e-6, 1-1e-6)                                  (:2488-2556)
//   3 (row units)     syndrome of soft_out > 0.5
// exp() of the channel transform is exp_glibc (the reference's, bit for bit): probabilities, hard decisions and step counts
// are identical to the CPU reference's.
// ---------------------------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void asp_body(const SpecArgs &a) {
    constexpr SpView<C> V{};
    constexpr int RH = C::RH, NH = C::NH, M = C::M, N = NH * M, CH = (M + 63) / 64, T = kSpWaves * 64;
    constexpr int NE = V.ne, UMAX = V.units_max;
    extern __shared__ double lds[];
    char *const stb = reinterpret_cast<char *>(lds);                         // state[e][n] at e*M*8 + n*8
    unsigned char *const hb = reinterpret_cast<unsigned char *>(stb + (size_t)NE * M * 8);  // [N] soft_out > 0.5
    int *const flag = reinterpret_cast<int *>(hb + ((N + 15) & ~15));
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // lanes beyond the lifting in the last 64-lane chunk of a circulant sit out
    auto lane_ok = [&](auto CHI) { constexpr int c = decltype(CHI)::value; return (c * 64 + 64 <= M) || (c * 64 + lane < M); };
    const long long fr = blockIdx.x;

    FrameVote fvote;
    fvote.init(flag);
    auto vote = [&](bool fail) -> bool { return fvote(fail); };
    auto syndrome_fail = [&]() -> bool {
        bool f = false;
        static_for<0, RH * CH>([&](auto U) {
            constexpr int u = decltype(U)::value, j = u / CH, ch = u % CH;
            if (wave == u % kSpWaves && lane_ok(IC<ch>{})) {
                const int n = ch * 64 + lane;
                unsigned sy = 0;
                static_for<0, C::RW[j]>([&](auto S) {
                    constexpr int s = decltype(S)::value;
                    int t = n + C::SH[j][s]; if (t >= M) t -= M;
                    sy ^= hb[C::COL[j][s] * M + t];
                });
                f |= sy != 0;
            }
        });
        return f;
    };

    double p1ch[UMAX], so[UMAX];   // channel P(bit=1) and a-posteriori probability of this wave's own columns
    static_for<0, UMAX>([&](auto Q) { p1ch[decltype(Q)::value] = 0.5; so[decltype(Q)::value] = 0.5; });
    static_for<0, NH * CH>([&](auto U) {
        constexpr int u = decltype(U)::value, k = u / CH, ch = u % CH, q = V.col_slot[u];
        if (wave == V.col_wave[u] && lane_ok(IC<ch>{})) {
            const int t = ch * 64 + lane;
            const double x = a.llr[fr * N + k * M + t] * 0.5;                 // :2351-2358
            const double y = at_least(at_most(x, 20.0), -20.0);
            const double e0 = exp_glibc(y), e1 = exp_glibc(-y);
            const double p = e1 / (e0 + e1);
            p1ch[q] = so[q] = p;
            hb[k * M + t] = p > 0.5;
            static_for<0, V.cw[k]>([&](auto X) {                              // :2361-2379 state <- rotated channel probability
                constexpr int x2 = decltype(X)::value;
                int nn = t - V.cc[k][x2]; if (nn < 0) nn += M;
                *reinterpret_cast<double *>(stb + (size_t)V.ce[k][x2] * M * 8 + nn * 8) = p;
            });
        }
    });
    __syncthreads();

    int res = 0;
    bool fail = vote(syndrome_fail());                                        // :2393-2399
    int steps = 0;
    while (fail && steps < a.maxiter) {
        asm volatile("" : "+v"(lane));   // per iteration: keeps the rotated LDS addresses of the unrolled phases out of the loop-invariant set (registers)
        // ---- phase 1: check nodes
        static_for<0, RH * CH>([&](auto U) {
            constexpr int u = decltype(U)::value, j = u / CH, ch = u % CH, RW = C::RW[j];
            static_assert(RW >= 2, "asp_body: map_bin needs at least two edges per check");
            if (wave == u % kSpWaves && lane_ok(IC<ch>{})) {
                const int n8 = (ch * 64 + lane) * 8;
                double P[RW], SF[RW], SB[RW];
                static_for<0, RW>([&](auto S) {
                    constexpr int s = decltype(S)::value;
                    P[s] = 1 - 2 * *reinterpret_cast<const double *>(stb + (size_t)(V.row_off[j] + s) * M * 8 + n8);  // :2206
                });
                SF[0] = P[0];
                static_for<1, RW - 1>([&](auto I) { constexpr int i = decltype(I)::value; SF[i] = P[i] * SF[i - 1]; });
                SB[RW - 1] = P[RW - 1];
                static_for<0, RW - 2>([&](auto I) { constexpr int i = RW - 2 - decltype(I)::value; SB[i] = P[i] * SB[i + 1]; });
                *reinterpret_cast<double *>(stb + (size_t)(V.row_off[j] + 0) * M * 8 + n8) = (1 - SB[1]) / 2;
                static_for<1, RW - 1>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    *reinterpret_cast<double *>(stb + (size_t)(V.row_off[j] + i) * M * 8 + n8) = (1 - SF[i - 1] * SB[i + 1]) / 2;
                });
                *reinterpret_cast<double *>(stb + (size_t)(V.row_off[j] + RW - 1) * M * 8 + n8) = (1 - SF[RW - 2]) / 2;
            }
        });
        __syncthreads();
        // ---- phase 2: symbol nodes + local data update
        static_for<0, NH * CH>([&](auto U) {
            constexpr int u = decltype(U)::value, k = u / CH, ch = u % CH, q = V.col_slot[u], CW = V.cw[k];
            if (wave == V.col_wave[u] && lane_ok(IC<ch>{})) {
                const int t = ch * 64 + lane;
                double d[CW];
                double P1 = p1ch[q], P0 = 1 - p1ch[q];                        // :2492-2496
                static_for<0, CW>([&](auto X) {
                    constexpr int x = decltype(X)::value;
                    int nn = t - V.cc[k][x]; if (nn < 0) nn += M;
                    d[x] = *reinterpret_cast<const double *>(stb + (size_t)V.ce[k][x] * M * 8 + nn * 8);
                    P1 *= d[x];                                               // :2511-2512, rows ascending
                    P0 *= 1 - d[x];
                });
                constexpr bool RD = CW <= 40;                                 // see div_ranged: P1 >= 4.2e-18 * 1e-6^CW
                const double sov = div_if<RD>(P1, P0 + P1);                   // :2519
                so[q] = sov;
                hb[k * M + t] = sov > 0.5;
                static_for<0, CW>([&](auto X) {                               // :2540-2548
                    constexpr int x = decltype(X)::value;
                    int nn = t - V.cc[k][x]; if (nn < 0) nn += M;
                    const double p1 = div_if<RD>(sov, d[x]);                  // d in [1e-6, 1 - 1e-6]: map_bin of clamped states
                    const double p0 = div_if<RD>(1 - sov, 1 - d[x]);
                    const double dd = div_if<RD>(p1, p1 + p0);
                    *reinterpret_cast<double *>(stb + (size_t)V.ce[k][x] * M * 8 + nn * 8) = at_least(at_most(dd, 1.0 - 0.000001), 0.000001);
                });
            }
        });
        __syncthreads();
        fail = vote(syndrome_fail());                                         // :2566
        steps = steps + 1;
    }
    res = fail ? -steps : (steps == 0 ? 0 : steps);                           // 0: input codeword; steps+1 upstream == steps here

    if (threadIdx.x == 0 && a.iters) a.iters[fr] = res;
    if (a.hard) {
        pack_hard<N, T>(a.hard + fr * ((N + 31) / 32), threadIdx.x, [&](int v) { return hb[v] != 0; });
    }
    if (a.soft_out) {
        static_for<0, NH * CH>([&](auto U) {
            constexpr int u = decltype(U)::value, k = u / CH, ch = u % CH, q = V.col_slot[u];
            if (wave == V.col_wave[u] && lane_ok(IC<ch>{})) a.soft_out[fr * N + k * M + ch * 64 + lane] = so[q];
        });
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Gallager belief propagation in the log domain (upstream bp_decod_qc_lm, decoders.cpp:1708-1920, decoder id 0).
// Work split of sp_body (8 waves per frame; block rows dealt round-robin, block columns dealt at compile time).
// Per-edge message ZZ[e][t] (fp64) and sign BB[e][t] are indexed by VARIABLE position like upstream's ZZ[j][k*M+t].
//   A  (column units)  A = exp(soft - ZZ); ZZ <- log|(A-1)/(A+1)|; BB <- A < 1                        (:1803-1812)
//   A' (row units)     s[j][n] = sum of the row's ZZ at (n+c) mod M, columns ascending; bs = xor of BB  (:1815-1824)
//   B  (column units)  soft = yd; rows ascending: A = exp(s - ZZ); ZZ <- clamp(+-log((1+A)/(1-A)), 19.07); soft += ZZ
//   C  (row units)     syndrome of soft < 0
// Upstream does not clear its syndrome array before the input check (:1742-1762), so that check sees the syndrome the
// previous call on the same state left behind (non-zero after a failed frame): `stale` carries
// it in, `synd_out` carries it out, and the host chains frames in order (ldpc_hip.hip).
// exp() / log() are evaluated with glibc's own algorithms (exp_glibc, log_glibc above), so the a-posteriori LLRs equal the CPU
// reference's bit for bit like everything else (round 1 used ocml's and matched to rtol 1e-5).
// ---------------------------------------------------------------------------------------------------------------
// The column phases (A and B) as DATA: what wave w does is read with scalar loads at run time.  Unrolled like the row phases (round 2)
// they were 224 inlined exp() + log() pairs = 300 KB of code against a 64 KB instruction cache that two CUs share, 205 spilled
// registers, 40 s of compile time.  As a loop over the unit's edge list the two phases are UMAX (= 4) copies of one exp / log pair
// each: 39 KB, no scratch, 4 s to compile, 25 % less time per launch (profiles/r03_sp_family_ab.txt).
template <class C>
struct BpPlan {
    static constexpr int CH = (C::M + 63) / 64, W = kSpWaves, UMAX = SpView<C>{}.units_max, RH = C::RH;
    struct Edge { int zoff, boff, jm, rot; };   // ZZ[e][0] in bytes, BB[e][0], j*M, shift
    int k[W][UMAX] = {}, ch[W][UMAX] = {}, cw[W][UMAX] = {};   // unit q of wave w: block column (-1: none), 64-lane chunk, column weight
    Edge edge[W][UMAX][RH] = {};
    constexpr BpPlan() {
        constexpr SpView<C> V{};
        for (int w = 0; w < W; ++w)
            for (int q = 0; q < UMAX; ++q) k[w][q] = -1;
        for (int u = 0; u < C::NH * CH; ++u) {
            const int w = V.col_wave[u], q = V.col_slot[u], kk = u / CH;
            k[w][q] = kk; ch[w][q] = u % CH; cw[w][q] = V.cw[kk];
            for (int x = 0; x < V.cw[kk]; ++x) edge[w][q][x] = Edge{V.ce[kk][x] * C::M * 8, V.ce[kk][x] * C::M, V.cj[kk][x] * C::M, V.cc[kk][x]};
        }
    }
};
template <class C>
__device__ const BpPlan<C> kBpPlan{};

template <class C>
__device__ __forceinline__ void bp_body(const SpecArgs &a) {
    constexpr SpView<C> V{};
    constexpr int RH = C::RH, NH = C::NH, M = C::M, N = NH * M, R = RH * M, CH = (M + 63) / 64, T = kSpWaves * 64;
    constexpr int NE = V.ne, UMAX = V.units_max, RUMAX = (RH * CH + kSpWaves - 1) / kSpWaves;
    extern __shared__ double lds[];
    char *const zzb = reinterpret_cast<char *>(lds);                                  // ZZ[e][t] at e*M*8 + t*8
    char *const sb = zzb + (size_t)NE * M * 8;                                        // s[j][n]
    unsigned char *const bbb = reinterpret_cast<unsigned char *>(sb + (size_t)R * 8); // BB[e][t]
    unsigned char *const bsb = bbb + (size_t)NE * M;                                  // bs[j][n]
    unsigned char *const hb = bsb + R;                                                // [N] soft < 0
    constexpr size_t kFlagOff = (((size_t)(NE * M + R) * 9 + N) + 15) & ~(size_t)15;  // 16-byte aligned behind the byte arrays
    int *const flag = reinterpret_cast<int *>(zzb + kFlagOff);
    constexpr bool TAB_LDS = kFlagOff + 16 + (size_t)kBpTabWords * 8 <= (size_t)160 * 1024;   // same rule as the host (plan_spec)
    const unsigned long long *etab = kExpTab, *ltab = kLogData;
    if constexpr (TAB_LDS) {
        unsigned long long *t = reinterpret_cast<unsigned long long *>(zzb + kFlagOff + 16);
        for (int i = threadIdx.x; i < kBpTabWords; i += T) t[i] = i < 256 ? kExpTab[i] : kLogData[i - 256];   // visible after the barrier below
        etab = t; ltab = t + 256;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto lane_ok = [&](auto CHI) { constexpr int c = decltype(CHI)::value; return (c * 64 + 64 <= M) || (c * 64 + lane < M); };
    const long long fr = a.frame_idx ? a.frame_idx[blockIdx.x] : (long long)blockIdx.x;
    const BpPlan<C> &P = kBpPlan<C>;

    FrameVote fvote;
    fvote.init(flag);
    auto vote = [&](bool fail) -> bool { return fvote(fail); };
    u64 left[RUMAX];   // syndrome bits of this wave's row units as last computed (what upstream leaves in st->syndr)
    static_for<0, RUMAX>([&](auto Q) { left[decltype(Q)::value] = 0ull; });
    // `ln` = the lane id, laundered once per iteration: the rotated LDS addresses of the unrolled row phases are loop invariant, and
    // hoisted out of the iteration loop they are ~100 live registers of which half go to scratch
    auto syndrome_fail = [&](bool with_stale, int ln) -> bool {
        bool f = false;
        static_for<0, RH * CH>([&](auto U) {
            constexpr int u = decltype(U)::value, j = u / CH, ch = u % CH;
            if (wave == u % kSpWaves && lane_ok(IC<ch>{})) {
                const int n = ch * 64 + ln;
                unsigned sy = 0;
                if (with_stale && a.stale) sy = (u32)(reinterpret_cast<const u64 *>(a.stale)[fr * (RH * CH) + u] >> lane) & 1u;
                static_for<0, C::RW[j]>([&](auto S) {
                    constexpr int s = decltype(S)::value;
                    int t = n + C::SH[j][s]; if (t >= M) t -= M;
                    sy ^= hb[C::COL[j][s] * M + t];
                });
                left[u / kSpWaves] = __ballot(sy != 0);
                f |= sy != 0;
            }
        });
        return f;
    };
    // this wave's unit q: (block column or -1, variable index of this lane, lane takes part, column weight) -- wave-uniform but for t
    auto unit = [&](int q, int &k, int &t, int &cw) -> bool {
        k = P.k[wave][q];
        const int ch = P.ch[wave][q];
        cw = P.cw[wave][q];
        t = ch * 64 + lane;
        return k >= 0 && (M % 64 == 0 || t < M);
    };

    double yd[UMAX], so[UMAX];
    static_for<0, UMAX>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        int k, t, cw;
        yd[q] = so[q] = 0.0;
        if (unit(q, k, t, cw)) {
            const double y = at_least(at_most(a.llr[fr * N + k * M + t], 20.0), -20.0);   // :1738 INPUT_LIMIT
            yd[q] = so[q] = y;
            hb[k * M + t] = y < 0;
            for (int x = 0; x < cw; ++x) *reinterpret_cast<double *>(zzb + P.edge[wave][q][x].zoff + t * 8) = 0.0;   // :1731-1733
        }
    });
    __syncthreads();

    bool fail = vote(syndrome_fail(true, lane));                                          // :1742-1766
    // Re-decode pass of the frame chain (frame_idx given, ldpc_hip.hip): the stale syndrome only enters this input check.  If the
    // check fails now and failed in the earlier pass too (that pass returned non-zero: it iterated), every later step is the
    // same as before -- the earlier outputs stand, nothing to redo.  Only a frame that is a codeword at the input (earlier
    // result 0) or whose own syndrome equals the stale one can change.
    if (a.frame_idx && fail && a.iters && a.iters[fr] != 0) return;
    int iter = 0;
    while (fail && iter < a.maxiter) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        // ---- A: variable-node activation
        static_for<0, UMAX>([&](auto Q) {   // UMAX copies of the edge loop (so[q] in registers); one copy with so[] picked by selects is 5 % slower
            constexpr int q = decltype(Q)::value;
            int k, t, cw;
            if (unit(q, k, t, cw)) {
                const double soq = so[q];
#pragma unroll 1
                for (int x = 0; x < cw; ++x) {
                    const typename BpPlan<C>::Edge ed = P.edge[wave][q][x];
                    double *z = reinterpret_cast<double *>(zzb + ed.zoff + t * 8);
                    const double A = exp_glibc_wide(soq - *z, etab);
                    *z = log_glibc_t(fabs((A - 1) / (A + 1)), ltab);
                    bbb[ed.boff + t] = A < 1;
                }
            }
        });
        __syncthreads();
        // ---- A': check sums
        static_for<0, RH * CH>([&](auto U) {
            constexpr int u = decltype(U)::value, j = u / CH, ch = u % CH;
            if (wave == u % kSpWaves && lane_ok(IC<ch>{})) {
                const int n = ch * 64 + ln;
                double s = 0.0;
                unsigned bs = 0;
                static_for<0, C::RW[j]>([&](auto S) {
                    constexpr int x = decltype(S)::value, e = V.row_off[j] + x;
                    int t = n + C::SH[j][x]; if (t >= M) t -= M;
                    s += *reinterpret_cast<const double *>(zzb + (size_t)e * M * 8 + t * 8);
                    bs ^= bbb[e * M + t];
                });
                *reinterpret_cast<double *>(sb + (size_t)(j * M + n) * 8) = s;
                bsb[j * M + n] = (unsigned char)bs;
            }
        });
        __syncthreads();
        // ---- B: check-node activation seen from the variable, a-posteriori sums
        static_for<0, UMAX>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            int k, t, cw;
            if (unit(q, k, t, cw)) {
                double soft = yd[q];                                                  // :1834
#pragma unroll 1
                for (int x = 0; x < cw; ++x) {
                    const typename BpPlan<C>::Edge ed = P.edge[wave][q][x];
                    int nn = t - ed.rot; if (nn < 0) nn += M;                         // rotate by m - circ (:1847)
                    double *z = reinterpret_cast<double *>(zzb + ed.zoff + t * 8);
                    double A = exp_glibc_wide(*reinterpret_cast<const double *>(sb + (size_t)(ed.jm + nn) * 8) - *z, etab);
                    const int b = bsb[ed.jm + nn] ^ bbb[ed.boff + t];
                    A = (double)(1 - 2 * b) * log_glibc_t((1 + A) / (1 - A), ltab);
                    const double zn = at_least(at_most(A, 19.07), -19.07);
                    *z = zn;
                    soft += zn;
                }
                so[q] = soft;
                hb[k * M + t] = soft < 0;
            }
        });
        __syncthreads();
        fail = vote(syndrome_fail(false, ln));                                          // :1869-1893 (array cleared at :1788)
        iter = iter + 1;
    }
    const int res = fail ? -iter : iter;

    if (threadIdx.x == 0 && a.iters) a.iters[fr] = res;
    if (a.synd_out) {
        static_for<0, RH * CH>([&](auto U) {
            constexpr int u = decltype(U)::value;
            if (wave == u % kSpWaves && lane == 0) reinterpret_cast<u64 *>(a.synd_out)[fr * (RH * CH) + u] = left[u / kSpWaves];
        });
    }
    if (a.hard) {
        pack_hard<N, T>(a.hard + fr * ((N + 31) / 32), threadIdx.x, [&](int v) { return hb[v] != 0; });
    }
    if (a.soft_out) {
        static_for<0, UMAX>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            int k, t, cw;
            if (unit(q, k, t, cw)) a.soft_out[fr * N + k * M + t] = so[q];
        });
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Integer min-sum (upstream imin_sum_decod_qc_lm, decoders.cpp:5430-5690, decoder id 4), code-specialised, for
// max_data <= 127 (MS_DBITS <= 8) and ialpha <= 16: every value is an int8.  One frame per workgroup of ceil(M/64) waves;
// lane n is check row n of every block row AND variable n of every block column.
//
// Upstream's STATE1 adds the check-to-variable messages into soft[] one edge after the other and SATURATES after every add
// (:5568), so the sum is order dependent and cannot be an atomic scatter.  Here the check lane publishes the messages of a
// block row as bytes (<= 8 edges -> 2 dwords per check) and the VARIABLE lane walks its column's edges in upstream's order
// (rows ascending) with one ds_read_i8 + add + med3 per edge; STATE2 (+ channel value, saturate) follows in registers.
// STATE3 runs on the check lane like the fp64 kernels, with (|v2c| << 3 | slot) as the key of a min / med3 pair, so
// "first minimum wins" (:5650) falls out of the integer order; the message it sends is then built for all 8 slots at once
// with byte-parallel arithmetic (EMIT below).  The message a check sent IS the value upstream recomputes from the old
// record in STATE3 (:5628-5631), so the record itself is never stored.
// Both LDS arrays are stored twice, M elements apart, so that a rotated access is base + lane + immediate.
//   soft2[k][2M]        int8   a-posteriori values
//   msg2[j][2][2M]      dword  4 message bytes each (two's complement), slots 0-3 / 4-7 of block row j
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int med3i(int x, int y, int z) {   // median of three (clang has no builtin for the integer form)
    int d;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
    return d;
}

#ifndef LDPC_IMS_MSG_COPIES
#define LDPC_IMS_MSG_COPIES 1
#endif
#ifndef LDPC_IMS_PACK_IY
#define LDPC_IMS_PACK_IY 1    // quantised channel values four to a VGPR (they are int8): 24 fewer live registers, no scratch spills at 3 waves per SIMD
#endif

template <class C>
struct ColView {   // column view of the code, compile time
    int cw[C::NH] = {};
    int cj[C::NH][C::RH] = {}, cs[C::NH][C::RH] = {}, cc[C::NH][C::RH] = {};   // block row, slot in that row, shift; rows ascending
    constexpr ColView() {
        for (int j = 0; j < C::RH; ++j)
            for (int s = 0; s < C::RW[j]; ++s) {
                const int k = C::COL[j][s];
                cj[k][cw[k]] = j; cs[k][cw[k]] = s; cc[k][cw[k]] = C::SH[j][s];
                ++cw[k];
            }
    }
};

// SMALL: M <= 32, floor(64/M) frames per wavefront (lane = f*M + n); a frame that has converged freezes while the others of
// its wave go on (see ms_small_body).  Otherwise one frame per workgroup of ceil(M/64) waves.
template <class C, bool SMALL>
__device__ __forceinline__ void ims_body_t(const SpecArgs &a) {
    static_assert(C::WMAX <= 8, "ims_body: at most 8 circulants per block row (message bytes in 2 dwords)");
    constexpr ColView<C> V{};
    constexpr int RH = C::RH, NH = C::NH, M = C::M, N = NH * M, W = (M + 63) / 64;
    static_assert(!SMALL || (M <= 32 && W == 1), "ims_small_body: M <= 32");
    extern __shared__ double lds[];
    // MC copies of the message array: 2 = rotation by immediate offset (fewer VALU), 1 = half the LDS (more frames per CU)
    constexpr int MC = LDPC_IMS_MSG_COPIES;
    constexpr int F = SMALL ? 64 / M : 1;                                   // frames per workgroup
    constexpr size_t kSoftBytes = (size_t)2 * N, kMsgBytes = (size_t)RH * 2 * MC * M * 4, kMsgBase = (F * kSoftBytes + 15) & ~(size_t)15;
    char *const lds0 = reinterpret_cast<char *>(lds);
    int *const flag = reinterpret_cast<int *>(lds0 + kMsgBase + F * kMsgBytes);
    const int f = SMALL ? (int)threadIdx.x / M : 0;
    const int n = SMALL ? (int)threadIdx.x - f * M : (int)threadIdx.x;
    const long long fr = SMALL ? (long long)blockIdx.x * F + f : (long long)blockIdx.x;
    const bool valid = SMALL ? (f < F && fr < a.nframes) : ((M % 64 == 0) || n < M);
    const int nv = valid ? n : 0;
    const long long frv = valid ? fr : 0;
    char *const softb = lds0 + (size_t)(valid ? f : 0) * kSoftBytes;        // this frame's soft2[k][2M] int8
    char *const msgb = lds0 + kMsgBase + (size_t)(valid ? f : 0) * kMsgBytes;   // this frame's msg[j][2][MC*M] dwords
    const u64 frame_lanes = SMALL ? ((M >= 32 ? 0xffffffffull : ((1ull << (M & 31)) - 1ull)) << ((valid ? f : 0) * M)) : ~0ull;
    const int md = a.ims_max_data, nmd = -a.ims_max_data;
    // Lanes talk to each other through LDS.  Several waves: a workgroup barrier.  One wave: LDS executes a wave's accesses in
    // program order, but the COMPILER only sees one thread and would move a load of [base + lane + c] above the store to
    // [base + lane] it "cannot alias" -- a compiler-level fence keeps the phases apart.
    auto phase_fence = [&]() {
        if constexpr (W > 1) __syncthreads();
        else asm volatile("" ::: "memory");
    };
    auto sat = [&](int x) { return med3i(x, nmd, md); };   // limit_val :4308
    FrameVote fvote;
    if constexpr (W > 1) fvote.init(flag);
    auto vote = [&](bool fail) -> bool {
        if constexpr (W == 1) return __ballot(fail) != 0ull;
        else return fvote(fail);
    };

    constexpr bool PACK_IY = LDPC_IMS_PACK_IY != 0;
    int iy[PACK_IY ? (NH + 3) / 4 : NH];                                    // :5472-5500 energy-normalised quantiser
    if constexpr (PACK_IY) static_for<0, (NH + 3) / 4>([&](auto Q) { iy[decltype(Q)::value] = 0; });
    {
        const double coef = a.ims_coef[frv];
        const double *yrow = a.llr + frv * N + nv;
        static_for<0, NH>([&](auto K) {
            constexpr int k = decltype(K)::value;
            double val = yrow[k * M];
            int sign = 0;
            if (val < 0) { val = -val; sign = 1; }
            val *= coef;
            if (val > a.ims_thr) val = a.ims_thr;
            const int ival = (int)(short)floor(val * a.ims_max_quant / a.ims_thr + 0.5);
            const int q = sign ? -ival : ival;                              // |q| <= max_quant <= 127: one byte
            if constexpr (PACK_IY) iy[k >> 2] |= (q & 0xff) << (8 * (k & 3));
            else iy[k] = q;
        });
    }
    auto iy_of = [&](auto K) -> int {
        constexpr int k = decltype(K)::value;
        if constexpr (PACK_IY) {
            int w = iy[k >> 2];
            asm volatile("" : "+v"(w));     // opaque per use: otherwise the unpacking is hoisted out of the iteration loop and all NH values stay live
            return __builtin_amdgcn_sbfe(w, 8 * (k & 3), 8);
        } else return iy[k];
    };
    u32 pm[RH][2];                                                          // the messages this check sent last (bytes)
    static_for<0, RH>([&](auto J) {
        constexpr int j = decltype(J)::value;
        pm[j][0] = 0u; pm[j][1] = 0u;                                       // all-zero records :5462-5470 -> zero messages
        if (valid) {
            static_for<0, 2>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                u32 *p = reinterpret_cast<u32 *>(msgb + ((size_t)(j * 2 + q) * MC * M + nv) * 4);
                p[0] = 0u;
                if constexpr (MC == 2) p[M] = 0u;
            });
        }
    });
    phase_fence();

    int res = -a.maxiter;
    bool done = SMALL ? !valid : false;
    for (int iter = 0; iter < a.maxiter; ++iter) {
        u32 failw = 0;
        if (SMALL ? !done : true) {      // one divergent region per iteration for the frames of a wave that are still running
        // ---------------- STATE1 + STATE2 on the variable lane (:5540-5604)
        // Software pipeline: the message bytes of column group g+1 are requested before group g is summed, and soft2 is
        // written only after every column is done (the compiler cannot tell msg2 from soft2, so a store in between would
        // pin all later loads behind it: one exposed LDS round trip per column).
        int softv[NH];
        {
            constexpr int G = 4, NG = (NH + G - 1) / G;
            int mv[2][G][RH];
            auto request = [&](auto GI) {
                constexpr int g = decltype(GI)::value;
                static_for<g * G, (g * G + G < NH ? g * G + G : NH)>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    static_for<0, V.cw[k]>([&](auto X) {
                        constexpr int x = decltype(X)::value, j = V.cj[k][x], s = V.cs[k][x], c = V.cc[k][x];
                        if constexpr (MC == 2) {
                            constexpr int off = ((j * 2 + (s >> 2)) * 2 * M + (M - c) % M) * 4 + (s & 3);
                            mv[g & 1][k % G][x] = (int)*reinterpret_cast<const signed char *>(msgb + off + nv * 4);
                        } else {
                            constexpr int off = (j * 2 + (s >> 2)) * M * 4 + (s & 3), rot = (M - c) % M;
                            u32 pos = (u32)nv + (u32)rot;                                     // (t - c) mod M
                            if constexpr (rot == 0) {}
                            else if constexpr ((M & (M - 1)) == 0) pos &= (u32)(M - 1);
                            else pos = pos < pos - (u32)M ? pos : pos - (u32)M;              // unsigned min: pos - M wraps when pos < M
                            mv[g & 1][k % G][x] = (int)*reinterpret_cast<const signed char *>(msgb + off + pos * 4);
                        }
                    });
                });
            };
            request(IC<0>{});
            static_for<0, NG>([&](auto GI) {
                constexpr int g = decltype(GI)::value;
                if constexpr (g + 1 < NG) request(IC<g + 1>{});
                __builtin_amdgcn_sched_barrier(0);
                static_for<g * G, (g * G + G < NH ? g * G + G : NH)>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    int acc = 0;
                    static_for<0, V.cw[k]>([&](auto X) { acc = sat(acc + mv[g & 1][k % G][decltype(X)::value]); });   // rows ascending, saturate per add :5568
                    softv[k] = sat(iy_of(K) + acc);                                                                       // :5590-5597
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        if (valid) {
            static_for<0, NH>([&](auto K) {
                constexpr int k = decltype(K)::value;
                signed char *p = reinterpret_cast<signed char *>(softb + k * 2 * M + nv);
                p[0] = (signed char)softv[k]; p[M] = (signed char)softv[k];
            });
        }
        phase_fence();
        // ---------------- STATE3 on the check lane (:5610-5678) + the messages of the next STATE1
        int rb[2][8];                                                         // soft values of block row j / j+1 (same pipeline)
        auto request_row = [&](auto JI) {
            constexpr int j = decltype(JI)::value;
            static_for<0, C::RW[j]>([&](auto S) {
                constexpr int s = decltype(S)::value;
                rb[j & 1][s] = (int)*reinterpret_cast<const signed char *>(softb + C::COL[j][s] * 2 * M + C::SH[j][s] + nv);
            });
        };
        request_row(IC<0>{});
        static_for<0, RH>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int RW = C::RW[j];
            if constexpr (j + 1 < RH) request_row(IC<j + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            int r[RW];
            static_for<0, RW>([&](auto S) { r[decltype(S)::value] = rb[j & 1][decltype(S)::value]; });
            u32 k1 = (u32)md << 3, k2 = (u32)md << 3, slo = 0u, shi = 0u, par = 0u, sy = 0u;
            static_for<0, RW>([&](auto S) {
                constexpr int s = decltype(S)::value;
                const int t = __builtin_amdgcn_sbfe((int)pm[j][s >> 2], 8 * (s & 3), 8);
                const int msg = r[s] - t;                                     // :5633
                sy ^= (u32)r[s];
                par ^= (u32)msg;
                // byte 3 of msg is 0x00 / 0xff = its sign: drop it into byte (s & 3) of the row's sign-byte word
                constexpr u32 sel = (0x03020100u & ~(0xffu << (8 * (s & 3)))) | (0x07u << (8 * (s & 3)));
                if constexpr (s < 4) slo = __builtin_amdgcn_perm((u32)msg, slo, sel);
                else shi = __builtin_amdgcn_perm((u32)msg, shi, sel);
                const int neg = t - r[s];
                const u32 key = ((u32)(msg > neg ? msg : neg) << 3) | (u32)s;   // |msg|; the clamp to max_data (:5645) is in the start value
                k2 = (u32)med3i((int)key, (int)k1, (int)k2);
                k1 = key < k1 ? key : k1;
            });
            failw |= sy;
            // EMIT: message of slot s = sign_s ^ parity ? -mag_s : mag_s, mag_s = (s == pos ? min2 : min1) * ialpha >> 4  (:5556-5566)
            const u32 a1 = ((k1 >> 3) * (u32)a.ims_ialpha) >> 4, a2 = ((k2 >> 3) * (u32)a.ims_ialpha) >> 4;
            const u32 A = __builtin_amdgcn_perm(a1, a1, 0u);                  // a1 in all four bytes
            const u64 dd = (u64)(a1 ^ a2) << ((k1 & 7u) * 8u);                // turns byte `pos` into a2
            const u32 mlo = A ^ (u32)dd, mhi = A ^ (u32)(dd >> 32);
            const u32 pw = (u32)((int)par >> 31);
            const u32 nlo = slo ^ pw, nhi = shi ^ pw;                          // 0xff where the message is negative
            constexpr u32 K80 = 0x80808080u;
            const u32 ql = (K80 - mlo) ^ K80, qh = (K80 - mhi) ^ K80;          // -mag per byte (mag <= 127: no borrow between bytes)
            pm[j][0] = (nlo & ql) | (~nlo & mlo);
            pm[j][1] = (nhi & qh) | (~nhi & mhi);
        });
        if (valid) {                                                          // published after ALL rows have read soft2 (see above)
            static_for<0, RH>([&](auto J) {
                constexpr int j = decltype(J)::value;
                u32 *p0 = reinterpret_cast<u32 *>(msgb + ((size_t)(j * 2 + 0) * MC * M + nv) * 4);
                p0[0] = pm[j][0];
                if constexpr (MC == 2) p0[M] = pm[j][0];
                if constexpr (C::RW[j] > 4) {
                    u32 *p1 = reinterpret_cast<u32 *>(msgb + ((size_t)(j * 2 + 1) * MC * M + nv) * 4);
                    p1[0] = pm[j][1];
                    if constexpr (MC == 2) p1[M] = pm[j][1];
                }
            });
        }
        }
        if constexpr (W == 1) phase_fence();                                  // (the vote's barriers do it for several waves)
        if constexpr (SMALL) {
            const u64 failing = __ballot(!done && (failw >> 31) != 0);        // per check, frames still running
            if (!done && (failing & frame_lanes) == 0ull) { done = true; res = iter + 1; }   // :5684-5689, per frame
            if (__ballot(!done) == 0ull) break;
        } else {
            if (!vote(valid && (failw >> 31) != 0)) { res = iter + 1; break; }    // :5684-5689
        }
    }

    if constexpr (SMALL) { if (!valid) return; }
    if ((SMALL ? n == 0 : threadIdx.x == 0) && a.iters) a.iters[fr] = res;
    phase_fence();
    if (a.hard) {
        constexpr int HW = (N + 31) / 32;
        if constexpr (SMALL) {
            for (int w = n; w < HW; w += M) {
                u32 bits = 0;
                for (int b = 0; b < 32; ++b) {
                    const int v = 32 * w + b;
                    if (v < N) bits |= ((u32)(int)*reinterpret_cast<const signed char *>(softb + (v / M) * 2 * M + v % M) >> 31) << b;
                }
                a.hard[fr * HW + w] = bits;
            }
        } else {
            pack_hard<N, W * 64>(a.hard + fr * HW, threadIdx.x, [&](int v) { return *reinterpret_cast<const signed char *>(softb + (v / M) * 2 * M + v % M) < 0; });
        }
    }
    if (a.soft_out && valid) {
        static_for<0, NH>([&](auto K) {
            constexpr int k = decltype(K)::value;
            a.soft_out[fr * N + k * M + n] = (double)*reinterpret_cast<const signed char *>(softb + k * 2 * M + n);
        });
    }
}

template <class C> __device__ __forceinline__ void ims_body(const SpecArgs &a) { ims_body_t<C, false>(a); }
template <class C> __device__ __forceinline__ void ims_small_body(const SpecArgs &a) { ims_body_t<C, true>(a); }

}  // namespace ldpc_spec
