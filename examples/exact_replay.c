/* exact_replay.c -- upstream's bp_simulation frame loop from plain C, with upstream's own noise stream continued on the GPU.
 *
 *   gcc -O2 -Iinclude examples/exact_replay.c -o exact_replay -Lldpc-lib_amd -lldpc_hip -Wl,-rpath,$PWD/ldpc-lib_amd
 *   ./exact_replay tests/golden/h16x32_m126.txt 64 3 50 2.0 4000 1
 *                  base-matrix file            M  decoder max-iterations snr n_experiments seed
 *
 * What main_simulation.cpp:492-500 does for one (code, SNR) point: `initial_random_seed = seed; reset_random();` then
 * bp_simulation(..., n_frame_errors = huge, n_experiments, snr, reference_frame_error = 1, ...).  The generator is
 * std::mt19937(seed) (commons_portable.cpp:140-158); random_codeword() first draws (nh - rh) * M values of next_random_int(0, 2), one
 * generator word each (bp_simulation.cpp:512,160-162); then every frame draws N samples of next_random_gaussian().  Here the 624
 * state words are produced by the public seeding recurrence, walked past the codeword draws on the host, and handed to the
 * device (ldpc_hip_mt_set_state); ldpc_hip_mt_frames then returns the per-frame records of exactly upstream's frames.  For the
 * example code at 2.0 dB, seed 1, 4000 experiments this prints 170 errored frames in 4001 -- the upstream binary's count
 * (BASELINE.md section 2).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ldpc_hip.h"

/* std::mt19937: seeding (init_genrand) and one step of the recurrence -- the public algorithm (Matsumoto & Nishimura 1998) */
static void mt_seed(uint32_t x[624], uint32_t seed) {
    x[0] = seed;
    for (int i = 1; i < 624; i++) x[i] = 1812433253u * (x[i - 1] ^ (x[i - 1] >> 30)) + (uint32_t)i;
}
static void mt_regenerate(uint32_t x[624]) {   /* the next block of 624 words, in place */
    for (int i = 0; i < 624; i++) {
        const uint32_t y = (x[i] & 0x80000000u) | (x[(i + 1) % 624] & 0x7fffffffu);
        x[i] = x[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
}

int main(int argc, char **argv) {
    if (argc < 8) {
        fprintf(stderr, "usage: %s <base-matrix.txt> <M> <decoder-id> <max-iterations> <snr> <n_experiments> <seed>\n", argv[0]);
        return 2;
    }
    const int M = atoi(argv[2]), dec = atoi(argv[3]), maxit = atoi(argv[4]);
    const double snr = atof(argv[5]);
    const long long n_experiments = atoll(argv[6]);
    const uint32_t seed = (uint32_t)strtoul(argv[7], NULL, 10);

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

    /* the generator after reset_random() and the codeword draws: position = words drawn so far inside the current block */
    uint32_t x[624];
    mt_seed(x, seed);
    long long pos = 624;                                  /* a freshly seeded std::mt19937 regenerates at its first draw */
    for (long long burn = (long long)(nh - rh) * M; burn > 0;) {
        if (pos == 624) { mt_regenerate(x); pos = 0; }
        const long long take = burn < 624 - pos ? burn : 624 - pos;
        pos += take;
        burn -= take;
    }
    if (ldpc_hip_mt_set_state(ctx, x, (int)pos) != 0) { fprintf(stderr, "%s\n", ldpc_hip_last_error()); return 1; }

    /* bp_simulation.cpp:591: `experiment <= n_experiments` admits n_experiments + 1 frames; no early stop in this example */
    const long long B = n_experiments + 1;
    int32_t *info = (int32_t *)malloc(sizeof(int32_t) * (size_t)B), *iters = (int32_t *)malloc(sizeof(int32_t) * (size_t)B);
    if (ldpc_hip_mt_frames(ctx, snr, /*modulation*/0, /*punctured*/0, maxit, /*alpha*/0.8, B, info, iters) != 0) {
        fprintf(stderr, "ldpc_hip_mt_frames: %s\n", ldpc_hip_last_error());
        return 1;
    }
    long long nse = 0, nde = 0, nue = 0;
    for (long long i = 0; i < B; i++)
        if (info[i] != 0) {                               /* :805-810 */
            nse += info[i] & ((1 << 30) - 1);
            nde++;
            if (iters[i] >= 0) nue++;
        }
    uint32_t st[624];
    int p = 0;
    ldpc_hip_mt_get_state(ctx, st, &p);                   /* what upstream's `generator` object holds afterwards */
    printf("frames %lld errored %lld undetected %lld FER %.6g BER %.6g generator-position %d first-state-word %u\n", B, nde, nue, (double)nde / (double)B,
           (double)nse / (double)B / (double)(ldpc_hip_n(ctx) - ldpc_hip_r(ctx)), p, st[0]);
    ldpc_hip_close(ctx);
    free(info); free(iters); free(h);
    return 0;
}
