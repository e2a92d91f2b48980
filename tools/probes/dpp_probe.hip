// probe: semantics of DPP wave_shr/wave_shl/row_shr on gfx950 (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(-1, v, CTRL, 0xf, 0xf, false); }
__global__ void k(int* out) {
  int l = threadIdx.x;
  out[l] = dppi<0x138>(l);        // wave_shr:1
  out[64 + l] = dppi<0x130>(l);   // wave_shl:1
  out[128 + l] = dppi<0x111>(l);  // row_shr:1
  out[192 + l] = dppi<0x101>(l);  // row_shl:1
}
int main() {
  int* d; hipMalloc(&d, 256 * sizeof(int)); k<<<1, 64>>>(d); int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[4] = {"wave_shr:1", "wave_shl:1", "row_shr:1", "row_shl:1"};
  for (int t = 0; t < 4; t++) { printf("%s:", nm[t]); for (int l = 0; l < 64; l++) printf(" %d", h[64 * t + l]); printf("\n"); }
  return 0;
}
