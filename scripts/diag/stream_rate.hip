// How fast can ONE wave (or a few) of one workgroup stream a buffer that sits in L2 -- the situation of a sweep wave reading its factor fragments?
// Per-lane 32-byte records (two dwordx4 loads, the MFMA fragment layout of the sweeps) or 16-byte ones (one dwordx4, fully coalesced),
// U independent 2 KB (1 KB) blocks requested before the first is used, buffers of 32 KB .. 64 MB read over and over.
// build: hipcc --offload-arch=gfx950 -O3 -o stream_rate scripts/diag/stream_rate.hip ; run: ./stream_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const d4 cgd4;
typedef __attribute__((address_space(1))) const d2 cgd2;

// each wave walks its own quarter of the buffer in blocks of 64 lanes x (WIDE ? 32 : 16) bytes, U blocks in flight
template <int U, bool WIDE>
__global__ void k_stream(const double *buf, size_t doubles, int reps, double *out, unsigned long long *cyc) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    constexpr int BLK = WIDE ? 256 : 128;                    // doubles per block
    const size_t per = doubles / nw / BLK / U * U;           // blocks per wave
    const double *base = buf + (size_t)wv * per * BLK;
    double acc = 0.0;
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        for (size_t b = 0; b < per; b += U) {
            d4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (WIDE) v[u] = *(cgd4 *)(base + (b + u) * BLK + lane * 4);
                else { const d2 t = *(cgd2 *)(base + (b + u) * BLK + lane * 2); v[u] = d4{t[0], t[1], 0.0, 0.0}; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u][0] + v[u][1] + v[u][2] + v[u][3];
        }
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class KF> void run(const char *name, KF kern, int U, bool wide, size_t kb, int threads, int blocks = 1) {
    double *buf, *out; unsigned long long *cyc, h;
    const size_t doubles = kb * 128;
    hipMalloc(&buf, doubles * 8); hipMemset(buf, 0, doubles * 8); hipMalloc(&out, 8 * 1024 * 1024); hipMalloc(&cyc, 8 * 1024);
    const int reps = (int)(65536 / kb > 4 ? 65536 / kb : 4);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, buf, doubles, 2, out, cyc);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, buf, doubles, reps, out, cyc);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const int nw = threads / 64, BLK = wide ? 256 : 128;
    const size_t per = doubles / nw / BLK / U * U;
    const double bytes = (double)per * BLK * 8 * nw * reps;
    printf("%-6s U=%d buffer %6zu KB  waves %d x %3d workgroups : %6.1f bytes/cycle per workgroup, %6.1f per wave  (%.0f cycles per %d-byte block and wave)\n", name, U, kb, nw, blocks,
           bytes / (double)h, bytes / (double)h / nw, (double)h / ((double)per * reps), BLK * 8);
    hipFree(buf); hipFree(out); hipFree(cyc);
}
#define RUN(U, W, KB, T) run(W ? "32B/ln" : "16B/ln", k_stream<U, W>, U, W, KB, T)
int main() {
    RUN(1, true, 256, 64); RUN(2, true, 256, 64); RUN(3, true, 256, 64); RUN(4, true, 256, 64); RUN(8, true, 256, 64);
    RUN(4, false, 256, 64); RUN(8, false, 256, 64); RUN(16, false, 256, 64);
    RUN(4, true, 16, 64); RUN(4, true, 2048, 64); RUN(4, true, 16384, 64); RUN(4, true, 262144, 64);
    RUN(4, true, 256, 128); RUN(4, true, 256, 256); RUN(8, true, 256, 256); RUN(4, true, 256, 512); RUN(4, true, 256, 1024);
    run("32B/ln", k_stream<4, true>, 4, true, 256, 256, 256); run("32B/ln", k_stream<4, true>, 4, true, 262144, 256, 1024);
    return 0;
}
