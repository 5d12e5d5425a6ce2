// compat_driver.cpp -- exercises the upstream call surface (ldpc/decoders.h) exactly the way bp_simulation.cpp does:
// decod_open -> fill hd -> decod_init -> per frame: copy LLRs into st->y, call the decoder on (st->y, st->decword).
// usage: compat_driver <in.bin> <out.bin>
//   in : int32 dec_id, rh, nh, M, B, maxiter, decision ; int16 hd[rh*nh] ; double llr[B*N]
//   out: int32 iters[B] ; double decword[B*N] ; double y_after[B*N]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifdef LDPC_COMPAT_UPSTREAM_HEADERS
#include "decoders.h"       // UPSTREAM'S header and DEC_STATE layout (oracle/Makefile `ref`: _ref/compat_driver_upstream)
#else
#include "ldpc/decoders.h"
#endif

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 3;
    int hdr[7];
    if (fread(hdr, sizeof(int), 7, f) != 7) return 4;
    const int dec_id = hdr[0], rh = hdr[1], nh = hdr[2], M = hdr[3], B = hdr[4], maxiter = hdr[5], decision = hdr[6];
    const int N = nh * M;
    std::vector<short> hd((size_t)rh * nh);
    std::vector<double> llr((size_t)B * N), dec((size_t)B * N), after((size_t)B * N);
    std::vector<int> iters(B);
    if (fread(hd.data(), sizeof(short), hd.size(), f) != hd.size()) return 5;
    if (fread(llr.data(), sizeof(double), llr.size(), f) != llr.size()) return 6;
    fclose(f);

    if (decod_open(FHT_DEC, 4, rh, nh, M) != NULL) return 10;          // not built -> NULL like an unknown id
    if (decod_init(NULL) != 1) return 11;                               // sic (decoders.cpp:1014-1015)
    DEC_STATE *st = decod_open(dec_id, 1, rh, nh, M);
    if (!st) return 12;
    for (int i = 0; i < rh; i++) for (int j = 0; j < nh; j++) st->hd[i][j] = hd[(size_t)i * nh + j];
    if (!decod_init(st)) return 13;
    for (int b = 0; b < B; b++) {
        memcpy(st->y, &llr[(size_t)b * N], sizeof(double) * N);
        int it;
        switch (dec_id) {                                               // bp_simulation.cpp:716-729
        case SP_DEC: it = sum_prod_decod_qc_lm(st, st->y, st->decword, maxiter, decision); break;
        case MS_DEC: it = min_sum_decod_qc_lm(st, st->y, st->decword, maxiter, decision, MS_ALPHA); break;
        case BP_DEC: it = bp_decod_qc_lm(st, st->y, st->decword, maxiter, decision); break;
        case ASP_DEC: it = sum_prod_gf2_decod_qc_lm(st, st->y, st->decword, maxiter, decision); break;
        case TASP_DEC: it = tdmp_sum_prod_gf2_decod_qc_lm(st, st->y, st->decword, maxiter, decision); break;
        case IMS_DEC: it = imin_sum_decod_qc_lm(st, st->y, st->decword, maxiter, decision, MS_ALPHA, MS_THR, MS_QBITS, MS_DBITS); break;
        default:     it = lmin_sum_decod_qc_lm(st, st->y, st->decword, maxiter, decision, MS_ALPHA, MS_BETA); break;
        }
        iters[b] = it;
        memcpy(&dec[(size_t)b * N], st->decword, sizeof(double) * N);
        memcpy(&after[(size_t)b * N], st->y, sizeof(double) * N);
    }
#ifndef LDPC_COMPAT_UPSTREAM_HEADERS
    // the batched extension (not part of upstream's header) must agree with the per-frame calls
    std::vector<double> llr2 = llr, dec2((size_t)B * N);
    std::vector<int> it2(B);
    if (dec_id == BP_DEC) {   // BP carries the last frame's syndrome in the state (decoders.cpp:1742-1762): start over like the loop above did
        decod_close(st);
        st = decod_open(dec_id, 1, rh, nh, M);
        if (!st) return 14;
        for (int i = 0; i < rh; i++) for (int j = 0; j < nh; j++) st->hd[i][j] = hd[(size_t)i * nh + j];
        if (!decod_init(st)) return 15;
    }
    ldpc_decod_batch(st, llr2.data(), dec2.data(), it2.data(), B, maxiter, decision);
    if (memcmp(it2.data(), iters.data(), sizeof(int) * B) != 0) return 20;
    if (dec_id != BP_DEC && dec_id != SP_DEC && dec_id != TASP_DEC && dec_id != ASP_DEC && memcmp(dec2.data(), dec.data(), sizeof(double) * dec.size()) != 0) return 21;
#endif
    decod_close(st);

    f = fopen(argv[2], "wb");
    fwrite(iters.data(), sizeof(int), B, f);
    fwrite(dec.data(), sizeof(double), dec.size(), f);
    fwrite(after.data(), sizeof(double), after.size(), f);
    fclose(f);
    return 0;
}
