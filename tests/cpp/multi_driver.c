/* multi_driver.c -- test program for the multi-device layer behind the C-ABI (csrc/ldpc_multi.hpp), plain C, no torch in the process.
 *
 *   multi_driver <base-matrix.txt> <M> <n_shards>
 *
 * All shards sit on device 0.  Prints one line per call with the counters; the Python test runs it for n = 1 (host sum) and, with
 * tests/cpp/rccl_stub.cpp loaded through LDPC_HIP_RCCL_PATH and LDPC_HIP_RCCL_ALLOW_DUPLICATE=1, for n = 2, 3, 8 through the
 * communicator path, and compares the lines.  With LDPC_HIP_TEST_FAIL_SHARD=<i> every counting call must FAIL (exit code 3 after
 * printing the messages) instead of hanging. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ldpc_hip.h"

static unsigned long long fnv(const int32_t *a, const int32_t *b, long long n) {
    unsigned long long h = 1469598103934665603ull;
    for (long long i = 0; i < n; i++) {
        h = (h ^ (unsigned)a[i]) * 1099511628211ull;
        h = (h ^ (unsigned)b[i]) * 1099511628211ull;
    }
    return h;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <base-matrix.txt> <M> <n_shards>\n", argv[0]); return 2; }
    const int M = atoi(argv[2]), n = atoi(argv[3]);
    FILE *f = fopen(argv[1], "rt");
    if (!f) { perror(argv[1]); return 1; }
    static int16_t raw[64 * 256], hd[64 * 256];
    char line[8192];
    int rh = 0, nh = 0;
    while (fgets(line, sizeof line, f)) {
        int k = 0;
        for (char *tok = strtok(line, " \t\r\n"); tok; tok = strtok(NULL, " \t\r\n")) raw[rh * 256 + k++] = (int16_t)atoi(tok);
        if (k == 0) continue;
        if (nh == 0) nh = k;
        if (k != nh || nh > 256 || rh >= 64) { fprintf(stderr, "bad matrix file\n"); return 1; }
        rh++;
    }
    fclose(f);
    for (int i = 0; i < rh; i++)            /* re-lift to M: main_simulation.cpp:400-414 */
        for (int j = 0; j < nh; j++) {
            int v = raw[i * 256 + j];
            if (v > 0) { v %= M; if (j == rh - 1 && v == 0) v = 1; }
            hd[i * nh + j] = (int16_t)v;
        }
    int devices[64];
    for (int i = 0; i < n && i < 64; i++) devices[i] = 0;
    const int expect_failure = getenv("LDPC_HIP_TEST_FAIL_SHARD") != NULL;
    int failures = 0;

    ldpc_hip_multi *m = NULL;
    if (ldpc_hip_open_multi(LDPC_HIP_MS_DEC, rh, nh, M, hd, devices, n, &m) != 0) { fprintf(stderr, "open: %s\n", ldpc_hip_last_error()); return 1; }
    printf("shards %d reduction %s comm_inits %lld\n", ldpc_hip_multi_shards(m), ldpc_hip_multi_reduction(m), ldpc_hip_multi_comm_inits());

    /* 1. counters only */
    const long long B = 6000, batch = 1000;
    unsigned long long cnt[4] = {0, 0, 0, 0}, sit = 0;
    int rc = ldpc_hip_simulate_multi(m, 2.0, 0, 0, 50, 0.8, 1, 0, B, batch, cnt, &sit);
    if (rc != 0) { printf("simulate_multi failed: %s\n", ldpc_hip_last_error()); failures++; }
    else printf("simulate %llu %llu %llu %llu %llu\n", cnt[0], cnt[1], cnt[2], cnt[3], sit);

    /* 2. ordered records */
    int32_t *info = malloc(sizeof(int32_t) * (size_t)B), *its = malloc(sizeof(int32_t) * (size_t)B);
    rc = ldpc_hip_frames_multi(m, 2.0, 0, 0, 50, 0.8, 1, 0, B, batch, info, its, cnt, &sit);
    if (rc != 0) { printf("frames_multi failed: %s\n", ldpc_hip_last_error()); failures++; }
    else printf("frames %llu %llu %llu %llu %llu records %016llx\n", cnt[0], cnt[1], cnt[2], cnt[3], sit, fnv(info, its, B));

    /* 3. resident batches: 24 * 250 frames dealt to the shards in contiguous runs (n divides 24), the same global frames for every n */
    const int N = ldpc_hip_n(ldpc_hip_multi_ctx(m, 0));
    const long long total = 6000, per = total / n;
    if (total % n == 0) {
        const double *ptr[64];
        double *buf[64];
        for (int i = 0; i < n; i++) {
            if (hipMalloc((void **)&buf[i], sizeof(double) * (size_t)per * N) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
            ptr[i] = buf[i];
            if (ldpc_hip_channel_llr_dev(ldpc_hip_multi_ctx(m, i), 2.0, 0, 0, 26.0, 1, (long long)i * per, per, buf[i], ldpc_hip_multi_stream(m, i)) != 0) {
                fprintf(stderr, "channel: %s\n", ldpc_hip_last_error());
                return 1;
            }
        }
        for (int rep = 0; rep < 2; rep++) {   /* twice: buffers are reused, counters must not accumulate across calls */
            rc = ldpc_hip_decode_count_multi(m, ptr, per, 0, 50, 0.8, cnt, &sit);
            if (rc != 0) { printf("decode_count_multi failed: %s\n", ldpc_hip_last_error()); failures++; }
            else printf("decode_count %llu %llu %llu %llu %llu\n", cnt[0], cnt[1], cnt[2], cnt[3], sit);
        }
        for (int i = 0; i < n; i++) (void)hipFree(buf[i]);
    }

    /* 4. a second multi context on the same device list reuses the communicators */
    ldpc_hip_multi *m2 = NULL;
    if (ldpc_hip_open_multi(LDPC_HIP_LMS_DEC, rh, nh, M, hd, devices, n, &m2) != 0) { fprintf(stderr, "open 2: %s\n", ldpc_hip_last_error()); return 1; }
    rc = ldpc_hip_simulate_multi(m2, 1.6, 0, 0, 50, 0.8, 1, 0, 2000, 500, cnt, &sit);
    if (rc != 0) { printf("simulate_multi (2nd context) failed: %s\n", ldpc_hip_last_error()); failures++; }
    else printf("second %llu %llu %llu %llu %llu\n", cnt[0], cnt[1], cnt[2], cnt[3], sit);
    printf("comm_inits %lld\n", ldpc_hip_multi_comm_inits());
    ldpc_hip_close_multi(m2);
    ldpc_hip_close_multi(m);
    ldpc_hip_multi_release_comms();
    free(info); free(its);
    if (expect_failure) return failures == 3 + (total % n == 0 ? 2 : 0) ? 3 : 4;   /* every counting call must have failed, none hung */
    return failures ? 1 : 0;
}
