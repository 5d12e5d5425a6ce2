// ubench_fetch.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access widths this library uses.
// MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reads exactly half the bytes of a 16 B/lane streaming read; other
// widths are uncalibrated.  The flagship decoder reads its channel LLRs 8 B/lane (global_load_dwordx2, one fp64 per
// lane, consecutive lanes -> consecutive doubles), so this program streams a KNOWN number of bytes once with each width:
//     read8_kernel   8 B / lane (double)         read16_kernel  16 B / lane (double2)
//     write8_kernel  8 B / lane stores           (WRITE_SIZE of the packed-bit / soft-value outputs)
// Run it under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); the ratio bytes / (counter * 1024)
// per kernel is the correction factor bench.py applies (profiles/r02_fetch_calibration.txt).
// The buffer (1 GiB) is larger than the 256 MiB Infinity Cache and is touched exactly once per kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void read8_kernel(const double *p, size_t n, double *sink) {
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 12345.678) sink[0] = acc;
}
__global__ void read16_kernel(const double2 *p, size_t n, double *sink) {
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 12345.678) sink[0] = acc;
}
// the decoder's own pattern: one wave per 16 KiB frame, lane reads frame[k*64 + lane] for k = 0..31, repeated `reps` times
__global__ void read8_frames_kernel(const double *p, int reps, double *sink) {
    const double *row = p + (size_t)blockIdx.x * 2048 + threadIdx.x;
    double acc = 0;
    for (int r = 0; r < reps; ++r) {
        int o = 0;
        asm volatile("" : "+v"(o));
        for (int k = 0; k < 32; ++k) acc += row[o + k * 64];
    }
    if (acc == 12345.678) sink[0] = acc;
}
__global__ void write8_kernel(double *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}

int main() {
    const size_t bytes = (size_t)1 << 30, n8 = bytes / 8, n16 = bytes / 16;
    double *buf = nullptr, *sink = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, bytes));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(read8_kernel, dim3(256 * 16), dim3(256), 0, 0, buf, n8, sink);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(read16_kernel, dim3(256 * 16), dim3(256), 0, 0, (const double2 *)buf, n16, sink);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(read8_frames_kernel, dim3(65536), dim3(64), 0, 0, buf, 1, sink);   // 65536 frames x 16 KiB = 1 GiB, once
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(read8_frames_kernel, dim3(65536), dim3(64), 0, 0, buf, 50, sink);  // the same frames re-read 50 times (L2 / MALL hits expected)
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(write8_kernel, dim3(256 * 16), dim3(256), 0, 0, buf, n8);
    CK(hipDeviceSynchronize());
    printf("bytes per kernel: read8 %zu, read16 %zu, read8_frames(x1) %zu, read8_frames(x50) first-touch %zu, write8 %zu\n", bytes, bytes, bytes, bytes, bytes);
    return 0;
}
