// ldpc_encode.hpp -- the systematic encoder of upstream's dual-diagonal QC-LDPC codes on the device (included by ldpc_hip.hip).
//
// qc_encode (bp_simulation.cpp:22-117; host restatement: include/ldpc/encoder.h): partial syndromes of the information part per
// block row, their sum over the block rows, then back-substitution along the double diagonal.  Every step is a XOR of rotated M-bit
// slices, i.e. byte / bit work with one frame per workgroup: thread h owns position h of every block (the rotations go through
// LDS), a few barriers per frame.  The parity part may consist of several bidiagonal / unidiagonal blocks (bp_simulation.cpp:142-191):
// they are encoded from the last to the first, each against the information part and the parity of the blocks behind it.
//   random_info_kernel      information bits from Philox4x32-10 keyed by (seed, codeword index, word)
//   qc_encode_kernel        information bits -> codewords (bytes 0/1, parity first)
//   cw_channel_order_kernel codewords -> transmit order through the direct interleaver map, zero padded to whole symbols (:570,:575)
//   cw_pack_kernel          codewords -> packed words for the error count (ldpc_frontend.hpp count_errors_kernel)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ldpc_frontend.hpp"

namespace ldpc {

struct RandomInfoArgs {
    uint8_t *info;        // [B][K] bytes 0/1
    long long B, first;   // codeword indices first .. first + B
    int K;
    uint64_t seed;
};

__global__ void __launch_bounds__(256) random_info_kernel(const RandomInfoArgs a) {
    const int groups = (a.K + 127) / 128;   // 128 bits per Philox block
    const long long total = a.B * groups;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long w = i / groups;
        const int g = (int)(i - w * groups);
        const uint64_t idx = (uint64_t)(a.first + w);
        const Philox4 p = philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)g, 2u /* stream tag: information bits */, (uint32_t)a.seed,
                                        (uint32_t)(a.seed >> 32));
        uint8_t *out = a.info + w * a.K + g * 128;
        const int n = a.K - g * 128 < 128 ? a.K - g * 128 : 128;
        for (int b = 0; b < n; ++b) out[b] = (uint8_t)((p.x[b >> 5] >> (b & 31)) & 1u);
    }
}

constexpr int kEncMaxBlocks = 16;

struct EncodeArgs {
    const uint8_t *info;   // [B][(c - b) * M]
    uint8_t *cw;           // [B][c * M]
    long long B;
    const int *hd;         // [b][c] shifts, negative = empty
    int b, c, M;
    // bp_simulation.cpp:142-191: the parity part may consist of several bidiagonal / unidiagonal blocks (:143-156); block q spans block
    // rows (and parity block columns) [off[q], hi[q]) and is encoded against everything to its right -- the information part and the
    // parity of the blocks after it -- so the blocks run from the last to the first (:167-185).  One block is the usual case.
    int nblk;
    int off[kEncMaxBlocks], hi[kEncMaxBlocks];
    int single[kEncMaxBlocks];   // :33 "is_single_diagonal" of the block
    int p[kEncMaxBlocks];        // :36-39 first row of the block with a positive shift in its special parity column hi - 1
};

__global__ void __launch_bounds__(256) qc_encode_kernel(const EncodeArgs a) {
    extern __shared__ uint8_t enc_sm[];   // cword[n] | synd[r] | sum[M]
    const int b = a.b, c = a.c, M = a.M, r = b * M, n = c * M, T = blockDim.x, tid = threadIdx.x;
    uint8_t *cword = enc_sm, *synd = enc_sm + n, *sum = synd + r;
    auto at = [&](int i, int j) { return a.hd[i * c + j]; };
    for (long long fr = blockIdx.x; fr < a.B; fr += gridDim.x) {
        for (int i = tid; i < n; i += T) cword[i] = i < r ? (uint8_t)0 : (uint8_t)(a.info[fr * (n - r) + (i - r)] & 1);
        __syncthreads();
        for (int q = a.nblk - 1; q >= 0; --q) {
            const int off = a.off[q], hi = a.hi[q], rb = hi - off;
            for (int idx = tid; idx < rb * M; idx += T) {                  // :49-62 partial syndromes over everything right of the block
                const int i = off + idx / M, h = idx % M;
                uint8_t s = 0;
                for (int j = hi; j < c; ++j) {
                    const int sh = at(i, j);
                    if (sh >= 0) s ^= cword[j * M + (h + sh) % M];
                }
                synd[idx] = s;
            }
            __syncthreads();
            for (int h = tid; h < M; h += T) {
                uint8_t s = 0;
                for (int i = 0; i < rb; ++i) s ^= synd[i * M + h];
                sum[h] = s;
            }
            __syncthreads();
            if (a.single[q]) {
                for (int i = tid; i < rb * M; i += T) cword[off * M + i] = synd[i];   // :64-68
            } else {
                const int sp = at(off + a.p[q], hi - 1), s0 = at(off, hi - 1);
                for (int h = tid; h < M; h += T) {                          // :70-84 back-substitution along the double diagonal
                    const uint8_t xh = sum[(h + M - sp) % M];
                    cword[(hi - 1) * M + h] = xh;
                    uint8_t v = synd[h];
                    if (s0 == 0) v ^= xh;
                    if (s0 > 0) v ^= sum[h];
                    cword[off * M + h] = v;
                    uint8_t prev = v;
                    for (int i = 1; i < rb - 1; ++i) {
                        const int idx = i * M + h;
                        uint8_t w = synd[idx] ^ prev;
                        const int si = at(off + i, hi - 1);
                        if (si == 0) w ^= xh;
                        if (si > 0) w ^= sum[h];
                        cword[off * M + idx] = w;
                        prev = w;
                    }
                }
            }
            __syncthreads();
        }
        for (int i = tid; i < n; i += T) a.cw[fr * n + i] = cword[i];
        __syncthreads();
    }
}

struct CwOrderArgs {
    const uint8_t *cw;       // [W][N]
    uint8_t *tx;             // [W][ntx]
    const int32_t *direct;   // [N] or null = identity
    long long W;
    int N, ntx;
};
__global__ void __launch_bounds__(256) cw_channel_order_kernel(const CwOrderArgs a) {
    const long long total = a.W * a.ntx;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long w = i / a.ntx;
        const int j = (int)(i - w * a.ntx);
        a.tx[i] = j < a.N ? (uint8_t)(a.cw[w * a.N + (a.direct ? a.direct[j] : j)] & 1) : (uint8_t)0;
    }
}

struct CwPackArgs {
    const uint8_t *cw;   // [W][N]
    uint32_t *packed;    // [W][hard_words]
    long long W;
    int N, hard_words;
};
__global__ void __launch_bounds__(256) cw_pack_kernel(const CwPackArgs a) {
    const long long total = a.W * a.hard_words;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long w = i / a.hard_words;
        const int k = (int)(i - w * a.hard_words);
        uint32_t bits = 0;
        for (int bit = 0; bit < 32; ++bit) {
            const int v = 32 * k + bit;
            if (v < a.N) bits |= (uint32_t)(a.cw[w * a.N + v] & 1) << bit;
        }
        a.packed[i] = bits;
    }
}

}  // namespace ldpc
