// Device-side building blocks shared by every WORLD kernel (gfx950 / CDNA4, wave64).
//
//  * block_sum / block_scan : 256-thread workgroup reductions and prefix sums built from
//    64-lane wave shuffles plus one LDS hop across the 4 waves.
//  * fft_lds<N>             : in-place complex FP64 Stockham FFT on an LDS-resident buffer.
//    Every pass pulls its radix-4 (or final radix-2) operands into registers, barriers, and
//    writes the auto-sorted outputs back into the same buffer, so no ping-pong copy is needed
//    and a 4096-point transform fits 64 KiB of the CU's 160 KiB LDS.
//
// All arithmetic is FP64: the reference is float64 end to end and the F0 stages take discrete
// decisions on it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WH_BLOCK 256
#define WH_WAVE 64

namespace wh {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = WH_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WH_WAVE);
  return v;
}

// Sum over the whole 256-thread block; result broadcast to every thread.
// `scratch` must hold >= 8 doubles of LDS.  Contains two barriers.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[w] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < WH_BLOCK / WH_WAVE; ++i) t += scratch[i];
  return t;
}

// Two sums at once (saves barriers).
__device__ __forceinline__ void block_sum2(double& a, double& b, double* scratch) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    scratch[w] = a;
    scratch[4 + w] = b;
  }
  __syncthreads();
  double ta = 0.0, tb = 0.0;
#pragma unroll
  for (int i = 0; i < WH_BLOCK / WH_WAVE; ++i) {
    ta += scratch[i];
    tb += scratch[4 + i];
  }
  a = ta;
  b = tb;
}

__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, double* scratch) {
  a = wave_sum(a);
  b = wave_sum(b);
  c = wave_sum(c);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    scratch[w] = a;
    scratch[4 + w] = b;
    scratch[8 + w] = c;
  }
  __syncthreads();
  double ta = 0.0, tb = 0.0, tc = 0.0;
#pragma unroll
  for (int i = 0; i < WH_BLOCK / WH_WAVE; ++i) {
    ta += scratch[i];
    tb += scratch[4 + i];
    tc += scratch[8 + i];
  }
  a = ta;
  b = tb;
  c = tc;
}

__device__ __forceinline__ double wave_scan_incl(double v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < WH_WAVE; o <<= 1) {
    double u = __shfl_up(v, o, WH_WAVE);
    if (lane >= o) v += u;
  }
  return v;
}

// In-place inclusive prefix sum of n doubles in LDS (n a multiple of WH_BLOCK).  Each thread
// owns a contiguous run of n/256 elements.  `scratch` >= 8 doubles.  Ends with a barrier.
__device__ __forceinline__ void block_scan_lds(double* a, int n, double* scratch) {
  const int per = n / WH_BLOCK;
  const int base = threadIdx.x * per;
  double run = 0.0;
  for (int i = 0; i < per; ++i) {
    run += a[base + i];
    a[base + i] = run;
  }
  double incl = wave_scan_incl(run);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 63) scratch[w] = incl;
  __syncthreads();
  double off = incl - run;
  for (int i = 0; i < w; ++i) off += scratch[i];
  for (int i = 0; i < per; ++i) a[base + i] += off;
  __syncthreads();
}

// ------------------------------------------------------------------------------------------
// FFT
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// One Stockham pass of radix R over N points with NT threads; NS = product of earlier radices.
// tw[i] = exp(-2*pi*i*sqrt(-1)/N), i in [0,N).  INV conjugates twiddles and butterflies.
template <int N, int NT, int R, int NS, bool INV>
__device__ __forceinline__ void fft_pass(double2* __restrict__ s, const double2* __restrict__ tw) {
  constexpr int J = N / R;
  constexpr int PER = (J + NT - 1) / NT;
  double2 v[PER][R];
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    const int j = tid + p * NT;
    if (J % NT == 0 || j < J) {
#pragma unroll
      for (int r = 0; r < R; ++r) v[p][r] = s[j + r * J];
    }
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    const int j = tid + p * NT;
    if (J % NT == 0 || j < J) {
      const int k = j & (NS - 1);
      if (NS > 1) {
        constexpr int STEP = N / (NS * R);
#pragma unroll
        for (int r = 1; r < R; ++r) {
          double2 w = tw[(k * r * STEP) & (N - 1)];
          if (INV) w.y = -w.y;
          v[p][r] = cmul(v[p][r], w);
        }
      }
      const int base = (j - k) * R + k;
      if (R == 4) {
        const double2 a0 = make_double2(v[p][0].x + v[p][2].x, v[p][0].y + v[p][2].y);
        const double2 a1 = make_double2(v[p][0].x - v[p][2].x, v[p][0].y - v[p][2].y);
        const double2 a2 = make_double2(v[p][1].x + v[p][3].x, v[p][1].y + v[p][3].y);
        const double2 a3 = make_double2(v[p][1].x - v[p][3].x, v[p][1].y - v[p][3].y);
        // forward: y1 = a1 - i*a3, y3 = a1 + i*a3 ; inverse swaps them
        const double2 ia3 = INV ? make_double2(-a3.y, a3.x) : make_double2(a3.y, -a3.x);
        s[base] = make_double2(a0.x + a2.x, a0.y + a2.y);
        s[base + NS] = make_double2(a1.x + ia3.x, a1.y + ia3.y);
        s[base + 2 * NS] = make_double2(a0.x - a2.x, a0.y - a2.y);
        s[base + 3 * NS] = make_double2(a1.x - ia3.x, a1.y - ia3.y);
      } else {
        s[base] = make_double2(v[p][0].x + v[p][1].x, v[p][0].y + v[p][1].y);
        s[base + NS] = make_double2(v[p][0].x - v[p][1].x, v[p][0].y - v[p][1].y);
      }
    }
  }
  __syncthreads();
}

template <int N, int NT, int NS, bool INV>
__device__ __forceinline__ void fft_passes(double2* s, const double2* tw) {
  if constexpr (NS < N) {
    if constexpr (NS * 4 <= N) {
      fft_pass<N, NT, 4, NS, INV>(s, tw);
      fft_passes<N, NT, NS * 4, INV>(s, tw);
    } else {
      fft_pass<N, NT, 2, NS, INV>(s, tw);
      fft_passes<N, NT, NS * 2, INV>(s, tw);
    }
  }
}

// In-place unnormalised DFT of N complex doubles resident in LDS.  Caller guarantees that the
// buffer is fully written and visible (barrier) on entry; visible on exit.  The inverse does
// NOT divide by N.
template <int N, bool INV>
__device__ __forceinline__ void fft_lds(double2* s, const double2* tw) {
  fft_passes<N, WH_BLOCK, 1, INV>(s, tw);
}

}  // namespace wh
