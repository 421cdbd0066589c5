// Issue rate of v_mfma_f64_4x4x4_4b_f64 on gfx950: K independent accumulator chains, one wave (or four) per workgroup.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate scripts/diag/mfma_rate.hip ; run: ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K>
__global__ void k_rate(double *out, unsigned long long *cyc, int reps) {
    double acc[K];
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5;
#pragma unroll
    for (int i = 0; i < K; ++i) acc[i] = i;
    __syncthreads();
    unsigned long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < K; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = clock64();
    double s = 0; for (int i = 0; i < K; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int K>
__global__ void k_fma(double *out, unsigned long long *cyc, int reps) {
    double acc[K];
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
#pragma unroll
    for (int i = 0; i < K; ++i) acc[i] = i;
    __syncthreads();
    unsigned long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < K; ++i) acc[i] = fma(a, acc[i], b);
    }
    unsigned long long t1 = clock64();
    double s = 0; for (int i = 0; i < K; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <class KF> void run(const char *name, KF kern, int K, int threads) {
    double *out; unsigned long long *cyc, h;
    hipMalloc(&out, 8 * 1024 * 64); hipMalloc(&cyc, 8);
    const int reps = 2000;
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, cyc, reps);
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, cyc, reps);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-10s K=%2d threads=%3d : %.1f cycles per instruction (per wave)\n", name, K, threads, (double)h / reps / K);
    hipFree(out); hipFree(cyc);
}
int main() {
    run("mfma", k_rate<1>, 1, 64); run("mfma", k_rate<2>, 2, 64); run("mfma", k_rate<4>, 4, 64); run("mfma", k_rate<8>, 8, 64); run("mfma", k_rate<16>, 16, 64);
    run("mfma", k_rate<8>, 8, 256);
    run("fma64", k_fma<1>, 1, 64); run("fma64", k_fma<4>, 4, 64); run("fma64", k_fma<8>, 8, 64); run("fma64", k_fma<16>, 16, 64); run("fma64", k_fma<16>, 16, 256);
    // two and four waves per SIMD (one workgroup of 512 / 1024 threads): does a second wave fill the dependency stalls of the first?
    run("fma64", k_fma<1>, 1, 256); run("fma64", k_fma<1>, 1, 512); run("fma64", k_fma<1>, 1, 1024);
    run("fma64", k_fma<4>, 4, 256); run("fma64", k_fma<4>, 4, 512); run("fma64", k_fma<4>, 4, 1024);
    run("mfma", k_rate<1>, 1, 256); run("mfma", k_rate<1>, 1, 512); run("mfma", k_rate<2>, 2, 512);
    return 0;
}
