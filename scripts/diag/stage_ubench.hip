// Development tool (not part of the product): cycles per stage of the sweep building blocks, one sweeping wave per chain
// as in the solver, under different memory conditions.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o /tmp/stage_ubench scripts/diag/stage_ubench.hip
#include "../../pympc_amd/csrc/mpcqp.hip"

// MODE 0: chain_sweep forward; 2 / 4: hybrid back substitution (so_sweep) with G' / G
// HELP 0: waves 2, 3 idle; 1: they touch one dword per cache line of the chain's stages (L2 warm-up) while the sweepers run
template <int MODE, int HELP>
__global__ __launch_bounds__(NT) void ub(const double *F, const double *G, const double *om, double *out, int nst, int fstride, int wgstride, int reps) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double *Tc = sh;
    for (int i = threadIdx.x; i < 2 * (nst + 1) * 16; i += NT) Tc[i] = 1e-3 * (i % 97);
    __syncthreads();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double *Fb = F + (size_t)blockIdx.x * wgstride;
    CoreArgs a; a.N = 2 * (nst + 1); a.fstage = fstride; a.nx = 12; a.nu = 4; a.NcT = a.N; a.rdu = (a.N + 1) * 12; a.F = Fb; a.G = G; a.om = om;
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        if (wv < 2) {
            const double *Fc = Fb + (size_t)wv * (nst + 1) * fstride;      // the wave's half of the instance's stages
            double *Tw = Tc + wv * (nst + 1) * 16;
            if (MODE == 0) chain_sweep<16>(0, +1, nst, fstride, Fc, Tw);
            if (MODE == 2) { CoreArgs b = a; b.F = Fc; so_sweep<16, true, false>(b, Tw, nst, -1, 1, nst); }
            if (MODE == 4) { CoreArgs b = a; b.F = Fc; so_sweep<16, true, true>(b, Tw, 0, +1, 1, nst); }
        } else if (HELP) {
            const double *Fc = Fb + (size_t)(wv - 2) * (nst + 1) * fstride;
            const int doubles = (nst + 1) * fstride;                       // one dword per 128-byte line
            double acc = 0.0;
            for (int o = lane * 16; o < doubles; o += 64 * 16) acc += ((cgdouble *)Fc)[o];
            if (acc == 1.2345e300) Tc[0] = acc;
        }
        __syncthreads();
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = (double)(t1 - t0) / ((double)reps * nst);
}

template <int MODE, int HELP>
static void run(const char *name, int grid, const double *F, const double *G, const double *om, double *out, int nst, int fstride, int wgstride, size_t smem) {
    const int reps = 100;
    hipFuncSetAttribute((const void *)ub<MODE, HELP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((ub<MODE, HELP>), dim3(grid), dim3(NT), smem, 0, F, G, om, out, nst, fstride, wgstride, reps);
    std::vector<double> h(grid);
    hipMemcpy(h.data(), out, grid * sizeof(double), hipMemcpyDeviceToHost);
    double s = 0, mx = 0; for (double v : h) { s += v; mx = std::max(mx, v); }
    printf("%-28s grid %4d  stage stride %4d: %7.0f cycles/stage (mean over workgroups; max %.0f), sweep of %d stages incl. start-up\n", name, grid, fstride, s / grid, mx, nst);
}

int main(int argc, char **argv) {
    const int nst = argc > 1 ? atoi(argv[1]) : 14, fst = FactorFmt<16>::STAGE, per_wg = 2 * (nst + 1) * fst + 512;
    const int maxgrid = 1024;
    double *F, *G, *om, *out;
    hipMalloc(&F, sizeof(double) * (size_t)per_wg * maxgrid); hipMalloc(&G, sizeof(double) * 512); hipMalloc(&om, sizeof(double) * 4096); hipMalloc(&out, sizeof(double) * maxgrid);
    std::vector<double> h((size_t)per_wg * maxgrid);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * ((i * 7919) % 101) - 0.05;
    hipMemcpy(F, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
    hipMemcpy(G, h.data(), 512 * sizeof(double), hipMemcpyHostToDevice); hipMemcpy(om, h.data(), 4096 * sizeof(double), hipMemcpyHostToDevice);
    const size_t smem = 38 * 1024 + (nst > 14 ? 2 * (nst - 14) * 16 * 8 : 0);      // the solver's footprint: four workgroups per CU
    for (int grid : {1, 256, 1024}) {
        for (int fstride : {0, fst}) {
            run<0, 0>("forward", grid, F, G, om, out, nst, fstride, per_wg, smem);
            run<2, 0>("hybrid back (G')", grid, F, G, om, out, nst, fstride, per_wg, smem);
            run<4, 0>("hybrid back (G)", grid, F, G, om, out, nst, fstride, per_wg, smem);
        }
        if (grid == 1) run<0, 1>("forward + L2 touch", grid, F, G, om, out, nst, fst, per_wg, smem);
    }
    return 0;
}
