// Micro-benchmarks of the sweep building blocks (development tool, not part of the product).
#include "../pympc_amd/csrc/mpcqp.hip"
#include <cstdio>
#include <vector>

__global__ void ub_mfma_chain(double *out, int reps) {
    d4 acc = {1.0, 2.0, 3.0, 4.0};
    double a = 1e-3 * threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; ++i) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[3], acc, 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = (double)(t1 - t0) / (4.0 * reps); }
    out[1 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int WHICH>
__global__ __launch_bounds__(NT) void ub_sweep(Lay L, const double *F, double *out, int reps) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double *Tc = sh, *ring = sh + L.N * 16;
    for (int i = threadIdx.x; i < L.N * 16; i += NT) Tc[i] = 1e-3 * i;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    long long w0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
        if (WHICH == 0) { if (threadIdx.x < 64) chain_sweep<16>(0, +1, L.N - 1, 0, L.fstage, F + (size_t)blockIdx.x * L.N * L.fstage, Tc); }
        if (WHICH == 1) sinv_apply<16>(L.N, L.N / 2, L.fstage, F + (size_t)blockIdx.x * L.N * L.fstage, Tc);
        if (WHICH == 2) kkt_core<16>(L.N, L.fstage, F + (size_t)blockIdx.x * L.N * L.fstage, Tc);
        __syncthreads();
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (double)(t1 - t0) / reps; out[1] = (double)(w1 - w0) / reps; }
    if (threadIdx.x == 0) out[2 + blockIdx.x] = Tc[5];
}

int main() {
    Lay L = make_layout(12, 4, 30, 30);
    int nblk = 1024;
    size_t fs = (size_t)L.N * L.fstage;
    std::vector<double> hF(fs * 4);
    for (size_t i = 0; i < hF.size(); ++i) hF[i] = 1e-3 * ((i * 7919) % 101) - 0.05;
    double *F, *out;
    hipMalloc(&F, fs * nblk * sizeof(double));
    for (int b = 0; b < nblk; ++b) hipMemcpy(F + b * fs, hF.data() + (b % 4) * 0, fs * sizeof(double), hipMemcpyHostToDevice);
    hipMalloc(&out, 4096 * sizeof(double));
    double h[3];
    hipLaunchKernelGGL(ub_mfma_chain, dim3(1), dim3(64), 0, 0, out, 1000);
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("dependent f64 16x16x4 MFMA (reference shape, not used by the sweeps any more): %.1f shader-clock ticks each (s_memtime)\n", h[0]);
    size_t smem = (L.N * 16 + 2048 + 64) * sizeof(double);
    const char *names[3] = {"chain_sweep fwd (1 wave)", "sinv_apply (4 waves)", "kkt_core"};
    for (int grid : {1, 256, 1024}) {
        for (int w = 0; w < 3; ++w) {
            for (int rep = 0; rep < 2; ++rep) {
                if (w == 0) hipLaunchKernelGGL(ub_sweep<0>, dim3(grid), dim3(NT), smem, 0, L, F, out, 50);
                if (w == 1) hipLaunchKernelGGL(ub_sweep<1>, dim3(grid), dim3(NT), smem, 0, L, F, out, 50);
                if (w == 2) hipLaunchKernelGGL(ub_sweep<2>, dim3(grid), dim3(NT), smem, 0, L, F, out, 50);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            printf("grid %4d  %-26s %9.0f memtime ticks  %8.2f us (wall_clock 100MHz)  per call; per stage %.0f ticks\n", grid, names[w], h[0], h[1] / 100.0, h[0] / 30.0);
        }
    }
    return 0;
}
