// Development tool: what one stage of a sweep costs a single wave, piece by piece (registers only -> LDS -> global ring).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o /tmp/chain_ubench scripts/diag/chain_ubench.hip
#include "../../pympc_amd/csrc/mpcqp.hip"

// one 16 x 16 fragment times a stage vector as the two partial sums of the MFMA pairs (what a stage of so_sweep issues)
__device__ __forceinline__ void frag_matvec_halves16(const d4 a, const double in, double &p, double &s) {
    const double r1 = rot_blocks<1>(in), r2 = rot_blocks<2>(in), r3 = rot_blocks<3>(in);
    p = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], in, 0.0, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f64_4x4x4f64(a[2], r2, 0.0, 0, 0, 0);
    p = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], r1, p, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f64_4x4x4f64(a[3], r3, s, 0, 0, 0);
}

template <int V>
__global__ __launch_bounds__(256) void ub(const double *F, double *out, int reps) {
    __shared__ double Tc[4096];
    const int lane = opaque_lane(threadIdx.x & 63);
    for (int i = threadIdx.x; i < 4096; i += 256) Tc[i] = 1e-3 * (i % 97);
    __syncthreads();
    if (threadIdx.x >= 64) return;
    d4 A = *(const d4 *)(F + lane * 4);
    double *tb = Tc + vec_lane_offset(lane);
    const bool writer = vec_lane_writer(lane);
    double x = 1e-3 * lane, own = tb[0];
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        const int k = r & 127;
        if (V == 0) {                   // four dependent MFMAs
            const double r1 = rot_blocks<1>(x), r2 = rot_blocks<2>(x), r3 = rot_blocks<3>(x);
            double o = 0.5;
            o = __builtin_amdgcn_mfma_f64_4x4x4f64(A[0], x, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f64_4x4x4f64(A[1], r1, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f64_4x4x4f64(A[2], r2, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f64_4x4x4f64(A[3], r3, o, 0, 0, 0);
            x = o;
        } else if (V == 1) {            // two pairs + add
            double p, s; frag_matvec_halves16(A, x, p, s); x = p + s;
        } else if (V == 2) {            // one MFMA
            x = __builtin_amdgcn_mfma_f64_4x4x4f64(A[0], x, 0.5, 0, 0, 0);
        } else if (V == 3) {            // one MFMA + one rotation of its result
            x = rot_blocks<1>(__builtin_amdgcn_mfma_f64_4x4x4f64(A[0], x, 0.5, 0, 0, 0));
        } else if (V == 4) {            // one rotation + add (VALU only)
            x = rot_blocks<1>(x) + 0.5;
        } else if (V == 5) {            // pairs + LDS own one stage ahead + conditional store
            const double nxt = tb[((k + 1) & 127) * 16];
            double p, s; frag_matvec_halves16(A, x, p, s); x = (p + s) + own;
            if (writer) tb[k * 16] = x;
            own = nxt;
        } else if (V == 6) {            // pairs + unconditional store, no own
            double p, s; frag_matvec_halves16(A, x, p, s); x = p + s;
            tb[k * 16] = x;
        } else if (V == 7) {            // two MFMA, independent accumulators, then add
            const double r2 = rot_blocks<2>(x);
            const double p = __builtin_amdgcn_mfma_f64_4x4x4f64(A[0], x, 0.0, 0, 0, 0);
            const double s = __builtin_amdgcn_mfma_f64_4x4x4f64(A[2], r2, 0.0, 0, 0, 0);
            x = p + s;
        } else if (V == 8) {            // 16x16x4 MFMA, one
            d4 acc = {0.5, 0.5, 0.5, 0.5};
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[0], x, acc, 0, 0, 0);
            x = acc[0];
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = (double)(t1 - t0) / reps;
    out[1 + threadIdx.x] = x;
}
template <int V> static void run(const char *name, const double *F, double *out) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(ub<V>, dim3(1), dim3(256), 0, 0, F, out, 4000);
    double h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    printf("%-64s %6.1f cycles per stage\n", name, h);
}
int main() {
    double *F, *out; hipMalloc(&F, 8 * 1024); hipMalloc(&out, 8 * 1024);
    std::vector<double> h(1024); for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * ((i * 7919) % 101) - 0.05;
    hipMemcpy(F, h.data(), 8 * 1024, hipMemcpyHostToDevice);
    run<2>("one 4x4x4 f64 MFMA, dependent", F, out);
    run<3>("one MFMA + one DPP rotation of its result", F, out);
    run<4>("one DPP rotation + add (vector ALU only)", F, out);
    run<7>("two independent MFMAs + add", F, out);
    run<0>("3 rotations + four dependent MFMAs", F, out);
    run<1>("3 rotations + two MFMA pairs + add", F, out);
    run<6>("pairs + unconditional LDS store", F, out);
    run<5>("pairs + own vector from LDS one stage ahead + conditional store", F, out);
    run<8>("one 16x16x4 f64 MFMA, dependent", F, out);
    return 0;
}
