// Does the DPP form of the 64-lane FP64 sum (quad_perm / row mirrors inside rows of 16, row_bcast:15 and :31 across
// rows, total in lane 63) assemble for gfx950 and give the sum?  (settles wh::wave_sum in wh_device.h)
// hipcc --offload-arch=gfx950 -O2 wave_sum_check.hip -o wave_sum_check.bin && ./wave_sum_check.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../python-world_amd/csrc/wh_device.h"
__global__ void k(const double* in, double* out) {
  const double v = in[threadIdx.x];
  out[threadIdx.x] = wh::wave_sum(v);
}
int main() {
  double h[64], *d_in, *d_out, o[64];
  double want = 0.0;
  for (int i = 0; i < 64; ++i) {
    h[i] = 1.0 / (i + 1) + (i % 7) * 1e3;
    want += h[i];
  }
  hipMalloc(&d_in, sizeof(h));
  hipMalloc(&d_out, sizeof(h));
  hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out);
  hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) bad += (o[i] - want > 1e-9 || want - o[i] > 1e-9);
  printf("want %.12f got lane0 %.12f lane63 %.12f  lanes off: %d\n", want, o[0], o[63], bad);
  return bad != 0;
}
