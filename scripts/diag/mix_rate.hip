// Do the FP64 matrix pipe and the FP64 vector ALU of a gfx950 SIMD run CONCURRENTLY?  One workgroup of 512 threads (two waves per SIMD: waves w and w + 4
// share SIMD w): waves 0..3 issue a stream of independent v_mfma_f64_4x4x4, waves 4..7 a stream of independent v_fma_f64 (or nothing, or the same as 0..3).
// Cycles of each half alone and of both together: if the two streams overlap, "together" is the longer of the two, not their sum.
// build: hipcc --offload-arch=gfx950 -O3 -o mix_rate scripts/diag/mix_rate.hip ; run: ./mix_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE_LO, int MODE_HI>      // 0 idle, 1 mfma stream (8 chains), 2 fma64 stream (16 chains)
__global__ __launch_bounds__(512) void k_mix(double *out, unsigned long long *cyc, int reps) {
    const int wv = threadIdx.x >> 6;
    const int mode = wv < 4 ? MODE_LO : MODE_HI;
    double acc[16];
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = i;
    __syncthreads();
    unsigned long long t0 = clock64();
    if (mode == 1) {
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
        }
    } else if (mode == 2) {
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = fma(a, acc[i], b);
        }
    }
    unsigned long long t1 = clock64();
    double s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[wv] = t1 - t0;
}
template <class KF> void run(const char *name, KF kern) {
    double *out; unsigned long long *cyc, h[8];
    hipMalloc(&out, 8 * 512); hipMalloc(&cyc, 64);
    const int reps = 2000;
    hipLaunchKernelGGL(kern, dim3(1), dim3(512), 0, 0, out, cyc, reps);
    hipLaunchKernelGGL(kern, dim3(1), dim3(512), 0, 0, out, cyc, reps);
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-34s wave 0: %8.0f cycles (%.1f per mfma of 8 x %d)   wave 4: %8.0f cycles (%.1f per fma64 of 16 x %d)\n", name, (double)h[0], (double)h[0] / reps / 8, reps, (double)h[4], (double)h[4] / reps / 16, reps);
    hipFree(out); hipFree(cyc);
}
int main() {
    run("mfma | idle", k_mix<1, 0>);
    run("idle | fma64", k_mix<0, 2>);
    run("mfma | fma64", k_mix<1, 2>);
    run("mfma | mfma", k_mix<1, 1>);
    run("fma64 | fma64", k_mix<2, 2>);
    return 0;
}
