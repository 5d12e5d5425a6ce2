/* simulate_multi.c -- the multi-GPU C-ABI from plain C: one FER point of a QC-LDPC code with real codewords, 16-QAM and a block
 * interleaver, the frames sharded over every GPU of the node (one host thread + stream + context per GPU inside the library, RCCL
 * all-reduce of the five counters).  INTEGRATION.md section 3.
 *
 *   gcc -O2 -Iinclude examples/simulate_multi.c -o simulate_multi -Lldpc-lib_amd -lldpc_hip -Wl,-rpath,$PWD/ldpc-lib_amd
 *   ./simulate_multi tests/golden/h16x32_m126.txt 64 3 50 5.2 1000000
 *                    base-matrix file            M  decoder max-iterations  Eb/N0  frames
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ldpc_hip.h"

int main(int argc, char **argv) {
    if (argc < 7) {
        fprintf(stderr, "usage: %s <base-matrix.txt> <M> <decoder-id> <max-iterations> <Eb/N0 dB> <frames>\n", argv[0]);
        return 2;
    }
    const int M = atoi(argv[2]), dec = atoi(argv[3]), maxit = atoi(argv[4]);
    const double snr = atof(argv[5]);
    const long long frames = atoll(argv[6]);
    FILE *f = fopen(argv[1], "rt");
    if (!f) { perror(argv[1]); return 1; }
    static int16_t raw[64 * 256], hd[64 * 256];
    char line[8192];
    int rh = 0, nh = 0;
    while (fgets(line, sizeof line, f)) {
        int n = 0;
        for (char *tok = strtok(line, " \t\r\n"); tok; tok = strtok(NULL, " \t\r\n")) raw[rh * 256 + n++] = (int16_t)atoi(tok);
        if (n == 0) continue;
        if (nh == 0) nh = n;
        if (n != nh || nh > 256 || rh >= 64) { fprintf(stderr, "bad matrix file\n"); return 1; }
        rh++;
    }
    fclose(f);
    for (int i = 0; i < rh; i++)            /* re-lift to M: main_simulation.cpp:400-414 */
        for (int j = 0; j < nh; j++) {
            int v = raw[i * 256 + j];
            if (v > 0) { v %= M; if (j == rh - 1 && v == 0) v = 1; }
            hd[i * nh + j] = (int16_t)v;
        }

    int ndev = ldpc_hip_device_count();
    if (ndev < 1) { fprintf(stderr, "no HIP device\n"); return 1; }
    int devices[64];
    if (ndev > 64) ndev = 64;
    for (int i = 0; i < ndev; i++) devices[i] = i;
    ldpc_hip_multi *m = NULL;
    if (ldpc_hip_open_multi(dec, rh, nh, M, hd, devices, ndev, &m) != 0) { fprintf(stderr, "%s\n", ldpc_hip_last_error()); return 1; }

    /* sixteen random codewords from the library's encoder; frame f transmits codeword f % 16 */
    const int N = nh * M, K = (nh - rh) * M, ncw = 16;
    uint8_t *info = malloc((size_t)K), *cw = malloc((size_t)ncw * N);
    srand(1);
    for (int w = 0; w < ncw; w++) {
        for (int i = 0; i < K; i++) info[i] = (uint8_t)(rand() & 1);
        if (ldpc_hip_encode_host(rh, nh, M, hd, info, cw + (size_t)w * N) != 0) { fprintf(stderr, "%s\n", ldpc_hip_last_error()); return 1; }
    }
    if (ldpc_hip_multi_set_codewords(m, cw, ncw) != 0 || ldpc_hip_multi_set_interleaver(m, 3, 64, 1) != 0) {
        fprintf(stderr, "%s\n", ldpc_hip_last_error());
        return 1;
    }
    unsigned long long cnt[4], sum_it = 0;
    if (ldpc_hip_simulate_multi(m, snr, /*modulation: QAM16*/ 2, /*punctured blocks*/ 0, maxit, 0.8, /*seed*/ 1, /*first frame*/ 0, frames,
                                /*frames per batch and GPU*/ 16384, cnt, &sum_it) != 0) {
        fprintf(stderr, "%s\n", ldpc_hip_last_error());
        return 1;
    }
    printf("# %d GPU(s), counters reduced by %s\n", ldpc_hip_multi_shards(m), ldpc_hip_multi_reduction(m));
    printf("Eb/N0 %.2f dB  frames %llu  FER %.6f  BER %.3e  undetected %llu  mean iterations %.2f\n", snr, cnt[3], (double)cnt[1] / (double)cnt[3],
           (double)cnt[0] / (double)cnt[3] / K, cnt[2], (double)sum_it / (double)cnt[3]);
    ldpc_hip_close_multi(m);
    free(info); free(cw);
    return 0;
}
