// Probe the operand layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 (development tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double *out) {       // block = pair (la, lb): A one-hot at lane la, B one-hot at lane lb
    int la = blockIdx.x / 64, lb = blockIdx.x % 64, l = threadIdx.x;
    double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[(size_t)blockIdx.x * 64 + l] = d;
}
__global__ void lat(double *out, int reps) {
    double acc = 1.0, a = 1e-3 * threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; ++i) { acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, acc, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, acc, acc, 0, 0, 0);
                                     acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, acc, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, acc, acc, 0, 0, 0); }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = (double)(t1 - t0) / (4.0 * reps);
    out[1 + threadIdx.x] = acc;
}
int main() {
    double *out; hipMalloc(&out, 64 * 64 * 64 * sizeof(double));
    hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, out);
    std::vector<double> h(64 * 64 * 64);
    hipMemcpy(h.data(), out, h.size() * sizeof(double), hipMemcpyDeviceToHost);
    // for each (la, lb) list destination lanes
    printf("pairs (la,lb)->ld with nonzero D (first 24 la, compact):\n");
    for (int la = 0; la < 64; ++la) {
        printf("la=%2d:", la);
        for (int lb = 0; lb < 64; ++lb) for (int ld = 0; ld < 64; ++ld) if (h[((size_t)la * 64 + lb) * 64 + ld] != 0.0) printf(" (%d->%d)", lb, ld);
        printf("\n");
    }
    hipLaunchKernelGGL(lat, dim3(1), dim3(64), 0, 0, out, 1000);
    double t; hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
    printf("dependent v_mfma_f64_4x4x4: %.1f cycles each\n", t);
    return 0;
}
