// Micro-benchmark: cost per element of a single-lane dependent FP64 add chain with different LDS traffic
// patterns around it (sizing the exact phase scan of wh_synthesis.hip).  hipcc --offload-arch=gfx950 -O3 chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int T = 2048;

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, int reps, int active) {
  __shared__ __attribute__((aligned(16))) double tile[T];
  for (int i = threadIdx.x; i < T; i += 64) tile[i] = 1e-3 * i;
  __syncthreads();
  double run = 0.0;
  if ((int)threadIdx.x < active) {
    double2* t2 = reinterpret_cast<double2*>(tile);
    for (int r = 0; r < reps; ++r) {
      if (MODE == 0) {  // pure chain, operands in registers
        double2 a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = t2[q];
        for (int k = 0; k < T / 16; ++k) {
#pragma unroll
          for (int q = 0; q < 8; ++q) { run += a[q].x; run += a[q].y; }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if (MODE == 1) {  // chain + reads (b128), no writes
        for (int k = 0; k < T / 2; k += 8) {
          double2 a[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) a[q] = t2[k + q];
#pragma unroll
          for (int q = 0; q < 8; ++q) { run += a[q].x; run += a[q].y; }
        }
      } else if (MODE == 2) {  // chain + writes b128 interleaved, operands in registers
        double2 a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = t2[q];
        for (int k = 0; k < T / 2; k += 8) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            double2 o;
            run += a[q].x; o.x = run; run += a[q].y; o.y = run;
            t2[k + q] = o;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else if (MODE == 3) {  // chain + writes b64 per element
        double2 a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = t2[q];
        for (int k = 0; k < T; k += 16) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            run += a[q].x; tile[k + 2 * q] = run; __builtin_amdgcn_sched_barrier(0);
            run += a[q].y; tile[k + 2 * q + 1] = run; __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else if (MODE == 4) {  // two independent chains interleaved (ILP=2) in registers
        double2 a[8];
        double run2 = 1.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = t2[q];
        for (int k = 0; k < T / 32; ++k) {
#pragma unroll
          for (int q = 0; q < 8; ++q) { run += a[q].x; run2 += a[q].y; run += a[q].y; run2 += a[q].x; }
          __builtin_amdgcn_sched_barrier(0);
        }
        run += run2;
      }
    }
  }
  if (threadIdx.x == 0) out[blockIdx.x] = run;
}

template <int MODE>
void run(const char* name, int active, int blocks) {
  double* d;
  hipMalloc(&d, 8 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 200;
  k<MODE><<<blocks, 64>>>(d, 10, active);
  hipEventRecord(e0);
  k<MODE><<<blocks, 64>>>(d, reps, active);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s active=%2d blocks=%3d : %.2f ns/element\n", name, active, blocks, ms * 1e6 / ((double)reps * T));
  hipFree(d);
}

int main() {
  for (int blocks : {1, 64}) {
    for (int act : {1, 64}) {
      run<0>("chain only", act, blocks);
      run<1>("chain + ds_read_b128", act, blocks);
      run<2>("chain + ds_write_b128 per 2", act, blocks);
      run<3>("chain + ds_write_b64 per 1", act, blocks);
      run<4>("2 independent chains (per element)", act, blocks);
    }
  }
  return 0;
}
