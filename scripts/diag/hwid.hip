// Diagnostic: where do the four waves of each 256-thread workgroup land (XCC, SE, CU, SIMD, workgroup slot)?
// Build: hipcc --offload-arch=gfx950 -O2 -o hwid hwid.hip ; run: ./hwid [nwg] [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out, int spin) {
  extern __shared__ double sm[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  sm[threadIdx.x] = threadIdx.x;
  __syncthreads();
  double a = sm[(threadIdx.x + 1) & 255];
  for (int i = 0; i < spin; ++i) a = a * 1.0000001 + 1e-9;     // keep every workgroup resident for a while
  if (a == 123.0) out[0] = 1;
  if ((threadIdx.x & 63) == 0) { out[2 * (blockIdx.x * 4 + threadIdx.x / 64)] = hw; out[2 * (blockIdx.x * 4 + threadIdx.x / 64) + 1] = xcc; }
}
int main(int argc, char** argv) {
  int nwg = argc > 1 ? atoi(argv[1]) : 1024; int lds = argc > 2 ? atoi(argv[2]) : 38912;
  unsigned* d; hipMalloc(&d, nwg * 8 * sizeof(unsigned)); std::vector<unsigned> h(nwg * 8);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, d, 200000);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  int hist[4][4] = {}; std::map<unsigned, int> percu; std::map<unsigned, int> simd_of_pair[2];
  for (int w = 0; w < nwg; ++w) for (int j = 0; j < 4; ++j) {
    unsigned hw = h[2 * (w * 4 + j)], xcc = h[2 * (w * 4 + j) + 1] & 15;
    unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, tg = (hw >> 16) & 15;
    hist[j][simd]++;
    unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    if (j == 0) percu[key]++;
    if (j < 2) simd_of_pair[0][(key << 2) | simd]++;
    if (w < 12) printf("wg %4d wave %d: xcc %u se %u sh %u cu %2u simd %u tg %2u waveid %u\n", w, j, xcc, se, sh, cu, simd, tg, hw & 15);
  }
  printf("wave-in-workgroup x SIMD histogram:\n");
  for (int j = 0; j < 4; ++j) printf("  wave %d: %5d %5d %5d %5d\n", j, hist[j][0], hist[j][1], hist[j][2], hist[j][3]);
  std::map<int, int> hh; for (auto& p : percu) hh[p.second]++;
  printf("CUs used %zu; workgroups per CU histogram:", percu.size()); for (auto& p : hh) printf(" %dx%d", p.second, p.first); printf("\n");
  std::map<int, int> h2; for (auto& p : simd_of_pair[0]) h2[p.second]++;
  printf("waves 0/1 per (CU, SIMD) histogram:"); for (auto& p : h2) printf(" %dx%d", p.second, p.first); printf("\n");
  return 0;
}
