// Operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950 (settles the lane mapping used by wh_features.hip).
// A[i][k] = i+1 for k == kk else 0 ; B[k][j] = 100*(j+1) for k == kk else 0  ->  D[i][j] = 100*(i+1)*(j+1) when the assumed
// operand mapping (A: lane 16k+i, B: lane 16k+j) is right; the print shows which (i, j) each lane/register holds.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(double* out, int kk) {
  const int l = threadIdx.x;
  const double a = (l / 16 == kk) ? (double)(l % 16 + 1) : 0.0;
  const double b = (l / 16 == kk) ? 100.0 * (l % 16 + 1) : 0.0;
  double4_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  double* d;
  (void)hipMalloc(&d, 256 * sizeof(double));
  for (int kk = 0; kk < 4; kk += 3) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, kk);
    double h[256];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("kk=%d\n", kk);
    for (int l : {0, 1, 15, 16, 17, 32, 48, 63})
      for (int r = 0; r < 4; ++r) {
        const int v = (int)(h[l * 4 + r] / 100.0 + 0.5);
        int fi = -1, fj = -1;
        for (int i = 1; i <= 16; ++i) for (int j = 1; j <= 16; ++j) if (i * j == v && fi < 0) { /* ambiguous products: print raw */ }
        printf("  lane %2d reg %d : %6.0f\n", l, r, h[l * 4 + r]);
      }
  }
  return 0;
}
