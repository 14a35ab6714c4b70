// Which lane does a DPP row rotate read?  (settles the control codes used by hv_refine_kernel's neighbour exchange)
// hipcc --offload-arch=gfx950 -O2 dpp_check.hip -o dpp_check.bin && ./dpp_check.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ int dpp(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
__global__ void k(int* out) {
  const int l = threadIdx.x;
  out[l] = dpp<0x121>(l);        // row_ror:1
  out[64 + l] = dpp<0x12F>(l);   // row_ror:15
  out[128 + l] = dpp<0x111>(l);  // row_shr:1
  out[192 + l] = dpp<0x101>(l);  // row_shl:1
}
int main() {
  int* d;
  hipMalloc(&d, 256 * sizeof(int));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"row_ror:1 (0x121)", "row_ror:15 (0x12F)", "row_shr:1 (0x111)", "row_shl:1 (0x101)"};
  for (int t = 0; t < 4; ++t) {
    printf("%-20s lane<-src:", names[t]);
    for (int l = 0; l < 18; ++l) printf(" %d<-%d", l, h[64 * t + l]);
    printf("\n");
  }
  return 0;
}
