// encoder.h -- systematic encoder for the dual-diagonal QC-LDPC codes upstream's search produces (header only, C++11).
//
// Own restatement of qc_encode / random_codeword (bp_simulation.cpp:22-191).  Upstream's bp_simulation() calls
// random_codeword() and then overwrites the result with zeros (:568), so the hot path never transmits a non-zero codeword;
// this header exists so that the GPU decoders can be exercised on non-zero codewords (tests: the decoders' sign symmetry) and
// for callers that want real codewords.  Information bits occupy block columns b..c-1 (positions [b*M, c*M)),
// parity bits block columns 0..b-1.  For a given information word the codeword is unique whenever the parity part of H is
// invertible, so any correct encoder returns exactly upstream's codeword; correctness is checked by H*c = 0 (here, like
// upstream :83-116, and in the tests).
#ifndef LDPC_ENCODER_H
#define LDPC_ENCODER_H

#include <vector>

namespace ldpc {

typedef std::vector<unsigned char> BitVec;

// bp_simulation.cpp:22-117.  mx: b x c row-major, negative = empty block; cword: c*M bits, information part filled in.
// Returns 0 ok, -1 the special parity column has no positive shift, 1 the result is not a codeword (structure not encodable).
inline int qc_encode(const int *mx, int b, int c, int M, BitVec &cword) {
    const int r = b * M;
    auto at = [&](int i, int j) { return mx[i * c + j]; };
    const bool single = b > 1 ? at(1, 0) < 0 : true;                                  // :33 "is_single_diagonal"
    int p = 0;
    while (p < b && at(p, b - 1) <= 0) ++p;                                             // :36-39
    if (p >= b && !single) return -1;
    BitVec synd((size_t)r, 0), sum((size_t)M, 0);
    for (int i = 0; i < b; ++i) {                                                       // :49-62 partial syndromes of the information part
        for (int j = b; j < c; ++j)
            if (at(i, j) >= 0)
                for (int h = 0; h < M; ++h) synd[(size_t)(i * M + h)] ^= cword[(size_t)(j * M + (h + at(i, j)) % M)];
        for (int h = 0; h < M; ++h) sum[(size_t)h] ^= synd[(size_t)(i * M + h)];
    }
    if (single) {
        for (int i = 0; i < r; ++i) cword[(size_t)i] = synd[(size_t)i];                 // :64-68
    } else {
        for (int h = 0; h < M; ++h) {                                                   // :70-84 back-substitution along the double diagonal
            const unsigned char xh = sum[(size_t)((h + M - at(p, b - 1)) % M)];
            cword[(size_t)((b - 1) * M + h)] = xh;
            unsigned char v = synd[(size_t)h];
            if (at(0, b - 1) == 0) v ^= xh;
            if (at(0, b - 1) > 0) v ^= sum[(size_t)h];
            cword[(size_t)h] = v;
            for (int i = 1; i < b - 1; ++i) {
                const int idx = i * M + h;
                unsigned char w = synd[(size_t)idx] ^ cword[(size_t)(idx - M)];
                if (at(i, b - 1) == 0) w ^= cword[(size_t)((b - 1) * M + h)];
                if (at(i, b - 1) > 0) w ^= sum[(size_t)h];
                cword[(size_t)idx] = w;
            }
        }
    }
    for (int i = 0; i < b; ++i)                                                         // :87-116 is it a codeword?
        for (int h = 0; h < M; ++h) {
            unsigned char s = 0;
            for (int j = 0; j < c; ++j)
                if (at(i, j) >= 0) s ^= cword[(size_t)(j * M + (h + at(i, j)) % M)];
            if (s) return 1;
        }
    return 0;
}

// bp_simulation.cpp:142-191 with the information bits given instead of drawn: the base matrix may consist of several
// bidiagonal / unidiagonal blocks (:143-156), encoded from the last block to the first (:167-185).
// info: (c - last_block_end) * M bits for block columns last_block_end..c-1 -- for the usual single-block matrix that is
// (c - b) * M.  Returns 0, or qc_encode's code, or 2 if the final validation (:119-140) fails.
inline int encode(const int *mx, int b, int c, int M, const unsigned char *info, BitVec &cword) {
    auto at = [&](int i, int j) { return mx[i * c + j]; };
    std::vector<int> brk(1, 0);
    for (int i = 1; i + 1 < b; ++i) {
        if (at(i, i) >= 0 && at(i + 1, i) >= 0 && at(i, i - 1) < 0) brk.push_back(i);                                        // a bidiagonal block begins
        if (i > 1 && at(i, i) >= 0 && at(i + 1, i) < 0 && at(i, i - 1) < 0 && at(i - 1, i - 1) >= 0 && at(i - 1, i - 2) >= 0)
            brk.push_back(i);                                                                                                // a unidiagonal block begins
    }
    brk.push_back(b);
    const int n = c * M;
    cword.assign((size_t)n, 0);
    for (int i = brk.back() * M; i < n; ++i) cword[(size_t)i] = info[i - brk.back() * M] & 1;
    for (int bi = (int)brk.size() - 2; bi >= 0; --bi) {
        const int hi = brk[(size_t)bi + 1], off = brk[(size_t)bi];
        const int rb = hi - off, cb = c - off;
        std::vector<int> sub((size_t)rb * cb);
        for (int rr = 0; rr < rb; ++rr)
            for (int cc = 0; cc < cb; ++cc) sub[(size_t)(rr * cb + cc)] = at(rr + off, cc + off);
        BitVec local(cword.begin() + (size_t)off * M, cword.end());
        const int rc = qc_encode(sub.data(), rb, cb, M, local);
        if (rc != 0) return rc;
        for (size_t k = 0; k < local.size(); ++k) cword[(size_t)off * M + k] = local[k];
    }
    for (int i = 0; i < b; ++i)                                                         // independent_validation :119-140
        for (int h = 0; h < M; ++h) {
            unsigned char s = 0;
            for (int j = 0; j < c; ++j)
                if (at(i, j) >= 0) s ^= cword[(size_t)(j * M + (h + at(i, j)) % M)];
            if (s) return 2;
        }
    return 0;
}

}  // namespace ldpc

#endif
