#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
    int x = threadIdx.x;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(0, x, 0x124, 0xF, 0xF, false);       // row_ror:4
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, false);  // row_ror:8
    out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, x, 0x12C, 0xF, 0xF, false); // row_ror:12
}
int main() { int *o; hipMalloc(&o, 192 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o); int h[192]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  for (int r = 0; r < 3; ++r) { printf("ror:%d lanes 0..19:", 4 * (r + 1)); for (int i = 0; i < 20; ++i) printf(" %d", h[r * 64 + i]); printf("\n"); } return 0; }
