/* simulate.c -- the C-ABI from plain C: FER/BER of a QC-LDPC code over an Eb/N0 sweep on one MI355X.
 *
 *   gcc -O2 -Iinclude examples/simulate.c -o simulate -Lldpc-lib_amd -lldpc_hip -Wl,-rpath,$PWD/ldpc-lib_amd
 *   ./simulate tests/golden/h16x32_m126.txt 64 3 50 1.0 3.0 0.5 100000
 *               base-matrix file            M  decoder(1 SP,3 MS,4 IMS,7 TASP,8 LMS) max-iterations  snr-from snr-to step frames
 *
 * The base matrix file holds rh rows of nh integers (-1 = empty circulant); shifts are re-lifted to M with upstream's
 * rule (main_simulation.cpp:400-414).  Noise is generated on the device (counter-based Philox), so this is the
 * throughput mode of INTEGRATION.md section 3.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ldpc_hip.h"

int main(int argc, char **argv) {
    if (argc < 9) {
        fprintf(stderr, "usage: %s <base-matrix.txt> <M> <decoder-id> <max-iterations> <snr-from> <snr-to> <snr-step> <frames>\n", argv[0]);
        return 2;
    }
    const int M = atoi(argv[2]), dec = atoi(argv[3]), maxit = atoi(argv[4]);
    const double s0 = atof(argv[5]), s1 = atof(argv[6]), ds = atof(argv[7]);
    const long long frames = atoll(argv[8]);

    /* read the matrix: count columns from the first line */
    FILE *f = fopen(argv[1], "rt");
    if (!f) { perror(argv[1]); return 1; }
    static int16_t hd[64 * 256];
    char line[8192];
    int rh = 0, nh = 0;
    while (fgets(line, sizeof line, f)) {
        int n = 0;
        for (char *tok = strtok(line, " \t\r\n"); tok; tok = strtok(NULL, " \t\r\n")) hd[rh * 256 + n++] = (int16_t)atoi(tok);
        if (n == 0) continue;
        if (nh == 0) nh = n;
        if (n != nh || nh > 256 || rh >= 64) { fprintf(stderr, "bad matrix file\n"); return 1; }
        rh++;
    }
    fclose(f);
    int16_t *h = (int16_t *)malloc(sizeof(int16_t) * rh * nh);
    for (int i = 0; i < rh; i++)
        for (int j = 0; j < nh; j++) {
            int v = hd[i * 256 + j];
            if (v > 0) {                        /* main_simulation.cpp:400-414 */
                v %= M;
                if (j == rh - 1 && v == 0) v = 1;
            }
            h[i * nh + j] = (int16_t)v;
        }

    ldpc_hip_ctx *ctx = NULL;
    if (ldpc_hip_open(dec, rh, nh, M, h, 0, &ctx) != 0) { fprintf(stderr, "ldpc_hip_open: %s\n", ldpc_hip_last_error()); return 1; }
    printf("# (%d,%d) code, M=%d, decoder %d [%s], %d iterations, %lld frames per point\n", ldpc_hip_n(ctx), ldpc_hip_n(ctx) - ldpc_hip_r(ctx), M,
           dec, ldpc_hip_kernel_name(ctx), maxit, frames);
    printf("# Eb/N0[dB]        FER          BER   mean-iters   frames/s\n");
    for (double snr = s0; snr <= s1 + 1e-9; snr += ds) {
        unsigned long long cnt[4], sit = 0;
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        if (ldpc_hip_simulate(ctx, snr, 0, 0, maxit, 0.8, /*seed*/ 1, /*first_frame*/ 0, frames, cnt, &sit) != 0) {
            fprintf(stderr, "ldpc_hip_simulate: %s\n", ldpc_hip_last_error());
            return 1;
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double sec = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
        printf("%9.3f %12.5e %12.5e %10.2f %12.0f\n", snr, (double)cnt[1] / cnt[3], (double)cnt[0] / cnt[3] / (ldpc_hip_n(ctx) - ldpc_hip_r(ctx)),
               (double)sit / cnt[3], cnt[3] / sec);
    }
    ldpc_hip_close(ctx);
    free(h);
    return 0;
}
