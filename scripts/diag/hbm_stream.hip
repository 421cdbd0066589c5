// What does the memory system deliver to a READ-ONLY whole-chip stream shaped like the sweeps' factor stream -- the denominator bench.py's
// frac_of_achievable has been missing (the guide's 6.29 TB/s is a float4 COPY: half reads, half writes)?
// 1024 workgroups x 4 waves (the headline launch) or 2048 / 4096 (refill), every lane 32 bytes per block (two dwordx4: the MFMA fragment
// layout), U blocks in flight per wave, each workgroup walking its own contiguous slice once per pass; buffers 128 MB (inside the 256 MB
// Infinity Cache after the first pass), 1 GB and 4 GB (beyond it); beside it a copy of the same shape and a read of 16 bytes per lane.
// HIP events around `passes` launches; GB/s = bytes / time.
// build: hipcc --offload-arch=gfx950 -O3 -o hbm_stream scripts/diag/hbm_stream.hip ; run: ./hbm_stream > profiles/r5_hbm_stream.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const d4 cgd4;
typedef __attribute__((address_space(1))) const d2 cgd2;
typedef __attribute__((address_space(1))) d4 gd4;

template <int U, int MODE>      // MODE 0: read 32 B per lane, 1: read 16 B per lane, 2: copy 32 B per lane
__global__ __launch_bounds__(256) void k_stream(const double *src, double *dst, size_t doubles_per_wg, double *sink) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int BLK = MODE == 1 ? 128 : 256;               // doubles per wave-block
    const size_t per = doubles_per_wg / 4 / BLK / U * U;     // blocks per wave
    const double *base = src + (size_t)blockIdx.x * doubles_per_wg + (size_t)wv * per * BLK;
    double *obase = dst + (size_t)blockIdx.x * doubles_per_wg + (size_t)wv * per * BLK;
    double acc = 0.0;
    for (size_t b = 0; b < per; b += U) {
        d4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 1) { const d2 t = *(cgd2 *)(base + (b + u) * BLK + lane * 2); v[u] = d4{t[0], t[1], 0.0, 0.0}; }
            else v[u] = *(cgd4 *)(base + (b + u) * BLK + lane * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 2) *(gd4 *)(obase + (b + u) * BLK + lane * 4) = v[u];
            else acc += v[u][0] + v[u][1] + v[u][2] + v[u][3];
        }
    }
    if (MODE != 2 && acc == 12345.678) sink[0] = acc;       // (never true: keeps the loads)
}

template <int U, int MODE>
static void run(const char *what, size_t mbytes, int wgs, int passes) {
    const size_t doubles = mbytes * 1024 * 1024 / 8;
    double *src = nullptr, *dst = nullptr, *sink = nullptr;
    if (hipMalloc(&src, doubles * 8) != hipSuccess) { printf("%-28s %5zu MB: allocation failed\n", what, mbytes); return; }
    hipMemset(src, 0, doubles * 8); hipMalloc(&sink, 64);
    if (MODE == 2) { hipMalloc(&dst, doubles * 8); hipMemset(dst, 0, doubles * 8); }
    const size_t per_wg = doubles / wgs / 4096 * 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_stream<U, MODE>), dim3(wgs), dim3(256), 0, 0, src, dst, per_wg, sink);      // warm (page tables, cache)
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int p = 0; p < passes; ++p) hipLaunchKernelGGL((k_stream<U, MODE>), dim3(wgs), dim3(256), 0, 0, src, dst, per_wg, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    constexpr int BLK = MODE == 1 ? 128 : 256;
    const size_t per = per_wg / 4 / BLK / U * U;
    const double moved = (double)per * BLK * 8 * 4 * wgs * passes * (MODE == 2 ? 2 : 1);
    printf("%-28s U=%d %5zu MB  %4d workgroups x 4 waves  %2d passes  %8.3f ms per pass  %7.1f GB/s%s\n", what, U, mbytes, wgs, passes, ms / passes, moved / (ms * 1e-3) / 1e9,
           MODE == 2 ? " (read + write)" : "");
    hipFree(src); if (dst) hipFree(dst); hipFree(sink);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("# %s, %d CUs; read-only whole-chip streams (scripts/diag/hbm_stream.hip)\n", p.name, p.multiProcessorCount);
    for (size_t mb : {128, 1024, 4096}) {
        run<4, 0>("read 32 B/lane", mb, 1024, mb <= 128 ? 40 : 10);
        run<4, 0>("read 32 B/lane", mb, 4096, mb <= 128 ? 40 : 10);
        run<8, 0>("read 32 B/lane", mb, 1024, mb <= 128 ? 40 : 10);
        run<2, 0>("read 32 B/lane", mb, 1024, mb <= 128 ? 40 : 10);
        run<8, 1>("read 16 B/lane", mb, 1024, mb <= 128 ? 40 : 10);
        run<4, 2>("copy 32 B/lane", mb / 2, 1024, mb <= 128 ? 40 : 10);
    }
    return 0;
}
