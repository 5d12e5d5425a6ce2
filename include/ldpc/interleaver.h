// interleaver.h -- the bit interleavers of upstream's simulation chain as plain index maps (header only, C++11).
//
// Upstream: Permutations_Open / Permutation_Init / Permutation (direct_inverse_perm.cpp:139-900), used by bp_simulation()
// between encoder and mapper (direct, bp_simulation.cpp:573) and between demapper and decoder (inverse, :684).
// Every mode boils down to one gather map per direction, out[i] = in[map[i]], i in [0, N), N = c*M:
//   0 identity            1 random permutation of all N positions
//   2 deterministic ("KBD"): spreads the block columns over the bits of a QAM symbol so that the columns of maximal
//     weight land on fixed bit positions; depends on the base matrix and on halfmlog = log2(Q)/2   (:486-700)
//   3 block random: one random permutation of `block_size` positions applied inside every block (+ one for the short tail)
//   4 interleaved random: one random permutation applied to each of the `step_size` interleaved sub-sequences
// Random permutations come from upstream's own LCG (`myrand`, :130-135, state reset to 1 by every Permutations_Open) with
// rejection of repeats (:293-311), so a given configuration always yields the same map; reproduced here bit for bit
// (tests/test_host_cpu.py checks against maps recorded from the compiled upstream code).
// The maps feed ldpc_hip_permute_dev (device gather) and the exact-replay harness (ldpc/bp_simulation.h).
#ifndef LDPC_INTERLEAVER_H
#define LDPC_INTERLEAVER_H

#include <string>
#include <vector>

namespace ldpc {

struct Interleaver {
    std::vector<int> direct, inverse;   // out[i] = in[direct[i]] (direction 0) / in[inverse[i]] (direction 1)
};

namespace detail {

struct UpstreamLcg {                     // direct_inverse_perm.cpp:130-135
    unsigned next = 1;
    int operator()() { next = next * 1103515245u + 12345u; return (int)(next & 0x3FFFFFFFu); }
};

// :293-311 draw until `size` distinct values have been seen, keep them in the order of first appearance
inline std::vector<int> random_perm(UpstreamLcg &rnd, int size) {
    std::vector<int> perm;
    std::vector<char> seen((size_t)(size > 0 ? size : 0), 0);
    perm.reserve(seen.size());
    while ((int)perm.size() < size) {
        const int rn = rnd() % size;
        if (seen[(size_t)rn]) continue;
        seen[(size_t)rn] = 1;
        perm.push_back(rn);
    }
    return perm;
}

inline bool invert(const std::vector<int> &perm, std::vector<int> &inv) {   // :702-716 (sort by value = inverse of a permutation)
    inv.assign(perm.size(), -1);
    for (size_t i = 0; i < perm.size(); ++i) {
        if (perm[i] < 0 || (size_t)perm[i] >= perm.size() || inv[(size_t)perm[i]] != -1) return false;
        inv[(size_t)perm[i]] = (int)i;
    }
    return true;
}

}  // namespace detail

// hd: base matrix, row-major b x c, negative = empty block.  halfmlog: 1 for BPSK / QAM4, 2 / 3 / 4 for QAM16 / 64 / 256
// (bp_simulation.cpp:402-411).  Returns false with a message for configurations upstream itself cannot interleave.
inline bool build_interleaver(int b, int c, int M, int halfmlog, int mode, int block_size, int step_size, const int *hd,
                              Interleaver &out, std::string &err) {
    const int N = c * M;
    if (b <= 0 || c <= 0 || M <= 0 || halfmlog < 1 || halfmlog > 4) { err = "interleaver: bad code shape"; return false; }
    detail::UpstreamLcg rnd;
    std::vector<int> perm((size_t)N, 0);
    out.direct.assign((size_t)N, 0);
    out.inverse.assign((size_t)N, 0);
    switch (mode) {
    case 0:
        for (int i = 0; i < N; ++i) perm[(size_t)i] = i;
        break;
    case 1:
        perm = detail::random_perm(rnd, N);
        break;
    case 2: {
        const int c0 = c / halfmlog, n0 = N / halfmlog, c0_mod = c % halfmlog;
        std::vector<int> cw((size_t)c, 0), V((size_t)c + 8, 0), q;
        for (int j = 0; j < c; ++j)
            for (int i = 0; i < b; ++i) cw[(size_t)j] += hd[i * c + j] >= 0;                       // :337-344
        int mw = cw[0];
        for (int j = 1; j < c; ++j) mw = cw[(size_t)j] > mw ? cw[(size_t)j] : mw;
        for (int j = 0, k = 0; j < halfmlog; ++j, ++k)                                             // :359-367 columns dealt round-robin
            for (int i = 0; i * halfmlog < c; ++i) V[(size_t)(j * c0 + i)] = k + i * halfmlog;    //          to the halfmlog bit positions
        if (c0_mod) for (int k = c0 * halfmlog; k < c; ++k) V[(size_t)k] = k;                      // :376-380
        int mrp = 0;
        for (int i = 0; i < c; ++i) {
            if (V[(size_t)i] < 0 || V[(size_t)i] >= c) { err = "interleaver: column order is not a permutation"; return false; }
            mrp += cw[(size_t)V[(size_t)i]] == mw;                                                 // :382-386 number of maximal-weight columns
        }
        for (int i = 0; i < c; ++i) if (V[(size_t)i] >= mrp) q.push_back(V[(size_t)i]);           // :389-396
        if ((int)q.size() != c - mrp) { err = "interleaver: column order is not a permutation"; return false; }
        for (int i = 0; i < c - mrp; ++i) V[(size_t)i] = q[(size_t)i];                             // :399-403 the first mrp columns go last
        for (int i = 0; i < mrp; ++i) V[(size_t)(c - mrp + i)] = i;
        if (c0_mod == 0) {                                                                         // :488-506 (then N % halfmlog == 0 as well)
            std::vector<int> s((size_t)n0);
            for (int i = 0, h = 0; i < halfmlog; ++i, h += c0) {
                for (int j = 0; j < c0; ++j)
                    for (int k = 0; k < M; ++k) s[(size_t)(j * M + k)] = V[(size_t)(j + h)] * M + k;
                for (int j = 0, k = 0; k < n0; ++k, j += halfmlog) perm[(size_t)(i + j)] = s[(size_t)k];
            }
        } else {                                                                                   // :508-654
            const int ibad = c - mrp, imix1 = ibad - mrp, imix2 = imix1 - mrp, imix3 = imix2 - mrp;
            const int limit = halfmlog == 2 ? imix1 : halfmlog == 3 ? imix2 : imix3;
            if (limit < 0) { err = "interleaver: more maximal-weight columns than the deterministic mode can place"; return false; }
            for (int k = 0; k < M; ++k)
                for (int i = 0; i < limit; ++i) perm[(size_t)(i * M + k)] = V[(size_t)i] * M + k;
            const int rem = (limit * M) % halfmlog;
            // which column group supplies bit position 0, 1, ... of the remaining symbols: 0 = bad (maximal weight), 1..3 = mix1..mix3
            static const int order2[2][2] = {{1, 0}, {0, 1}};
            static const int order3[3][3] = {{1, 2, 0}, {1, 0, 2}, {0, 1, 2}};
            static const int order4[4][4] = {{1, 2, 3, 0}, {1, 2, 0, 3}, {3, 0, 1, 2}, {0, 3, 1, 2}};
            const int base[4] = {ibad, imix1, imix2, imix3};
            int l = limit * M;
            for (int i = 0; i < mrp; ++i)
                for (int k = 0; k < M; ++k)
                    for (int t = 0; t < halfmlog; ++t) {
                        const int grp = halfmlog == 2 ? order2[rem][t] : halfmlog == 3 ? order3[rem][t] : order4[rem][t];
                        perm[(size_t)l++] = V[(size_t)(base[grp] + i)] * M + k;
                    }
        }
        break;
    }
    case 3:
    case 4: {
        const int bs = mode == 3 ? block_size : (step_size > 0 ? N / step_size : 0);               // :160-172
        if (bs <= 0) { err = "interleaver: block size must be positive"; return false; }
        const int nblocks = N / bs, tail = N % bs;
        const std::vector<int> pb = detail::random_perm(rnd, bs), pt = detail::random_perm(rnd, tail);   // :692-700 (same LCG stream)
        std::vector<int> ib, it;
        detail::invert(pb, ib);
        detail::invert(pt, it);
        if (mode == 3) {                                                                           // :812-857 contiguous blocks
            for (int k = 0; k < nblocks; ++k)
                for (int i = 0; i < bs; ++i) { out.direct[(size_t)(k * bs + i)] = k * bs + pb[(size_t)i]; out.inverse[(size_t)(k * bs + i)] = k * bs + ib[(size_t)i]; }
            for (int i = 0; i < tail; ++i) { out.direct[(size_t)(nblocks * bs + i)] = nblocks * bs + pt[(size_t)i]; out.inverse[(size_t)(nblocks * bs + i)] = nblocks * bs + it[(size_t)i]; }
        } else {                                                                                   // :858-897 sub-sequences k, k+step, k+2*step, ...
            if (tail != 0 || nblocks != step_size) { err = "interleaver: step size must divide the code length"; return false; }   // upstream leaves positions unwritten otherwise
            for (int k = 0; k < nblocks; ++k)
                for (int i = 0; i < bs; ++i) { out.direct[(size_t)(k + i * step_size)] = k + pb[(size_t)i] * step_size; out.inverse[(size_t)(k + i * step_size)] = k + ib[(size_t)i] * step_size; }
        }
        return true;
    }
    default:
        err = "interleaver: unknown permutation type";
        return false;
    }
    out.direct = perm;
    if (!detail::invert(perm, out.inverse)) { err = "interleaver: the generated map is not a permutation"; return false; }
    return true;
}

}  // namespace ldpc

#endif
