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

// Workgroup-wide synchronisation for NT cooperating threads.  A single-wave group (NT == 64) needs no
// hardware barrier: its lanes run in lockstep, so a compiler-level wavefront fence is enough to order the
// LDS traffic.  This is what lets the per-frame kernels run as one wave per frame with zero s_barrier.
template <int NT>
__device__ __forceinline__ void sync() {
  if constexpr (NT <= WH_WAVE) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = WH_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WH_WAVE);
  return v;
}

// Sum over the whole 256-thread block; result broadcast to every thread.
// `scratch` must hold >= 3*NT/64 doubles of LDS (24 for the largest block used, 512).  Contains two barriers.
template <int NT = WH_BLOCK>
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = wave_sum(v);
  if constexpr (NT <= WH_WAVE) {
    sync<NT>();
    return v;
  }
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[w] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < NT / WH_WAVE; ++i) t += scratch[i];
  return t;
}

// Two sums at once (saves barriers).
template <int NT = WH_BLOCK>
__device__ __forceinline__ void block_sum2(double& a, double& b, double* scratch) {
  a = wave_sum(a);
  b = wave_sum(b);
  if constexpr (NT <= WH_WAVE) {
    sync<NT>();
    return;
  }
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    scratch[w] = a;
    scratch[NT / WH_WAVE + w] = b;
  }
  __syncthreads();
  double ta = 0.0, tb = 0.0;
#pragma unroll
  for (int i = 0; i < NT / WH_WAVE; ++i) {
    ta += scratch[i];
    tb += scratch[NT / WH_WAVE + i];
  }
  a = ta;
  b = tb;
}

template <int NT = WH_BLOCK>
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, double* scratch) {
  a = wave_sum(a);
  b = wave_sum(b);
  c = wave_sum(c);
  if constexpr (NT <= WH_WAVE) {
    sync<NT>();
    return;
  }
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    scratch[w] = a;
    scratch[NT / WH_WAVE + w] = b;
    scratch[2 * (NT / WH_WAVE) + w] = c;
  }
  __syncthreads();
  double ta = 0.0, tb = 0.0, tc = 0.0;
#pragma unroll
  for (int i = 0; i < NT / WH_WAVE; ++i) {
    ta += scratch[i];
    tb += scratch[NT / WH_WAVE + i];
    tc += scratch[2 * (NT / WH_WAVE) + i];
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

// In-place inclusive prefix sum of n doubles in LDS (n a multiple of NT).  Each thread
// owns a contiguous run of n/256 elements.  `scratch` >= 8 doubles.  Ends with a barrier.
template <int NT = WH_BLOCK>
__device__ __forceinline__ void block_scan_lds(double* a, int n, double* scratch) {
  const int per = n / NT;
  const int base = threadIdx.x * per;
  double run = 0.0;
  for (int i = 0; i < per; ++i) {
    run += a[base + i];
    a[base + i] = run;
  }
  double incl = wave_scan_incl(run);
  double off = incl - run;
  if constexpr (NT > WH_WAVE) {
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 63) scratch[w] = incl;
    __syncthreads();
    for (int i = 0; i < w; ++i) off += scratch[i];
  }
  for (int i = 0; i < per; ++i) a[base + i] += off;
  sync<NT>();
}

// ------------------------------------------------------------------------------------------
// FFT
// ------------------------------------------------------------------------------------------
// complex multiply with fused multiply-adds (2 mul + 2 fma instead of 4 mul + 2 add)
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}

// One Stockham pass of radix R over N points with NT threads; NS = product of earlier radices.
// tw[i] = exp(-2*pi*i*sqrt(-1)/N), i in [0,N).  INV conjugates twiddles and butterflies.
// SNT >= NT: the barrier spans SNT threads while NT of them (thread index modulo NT) cooperate on this buffer,
// so that SNT/NT independent transforms on different buffers advance in lockstep through the same barriers.
template <int N, int NT, int R, int NS, bool INV, int SNT = NT>
__device__ __forceinline__ void fft_pass(double2* __restrict__ s, const double2* __restrict__ tw) {
  constexpr int J = N / R;
  constexpr int PER = (J + NT - 1) / NT;
  double2 v[PER][R];
  const int tid = threadIdx.x & (NT - 1);
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    const int j = tid + p * NT;
    if (J % NT == 0 || j < J) {
#pragma unroll
      for (int r = 0; r < R; ++r) v[p][r] = s[j + r * J];
    }
  }
  sync<SNT>();
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
  sync<SNT>();
}

template <int N, int NT, int NS, bool INV, int SNT = NT>
__device__ __forceinline__ void fft_passes(double2* s, const double2* tw) {
  if constexpr (NS < N) {
    if constexpr (NS * 4 <= N) {
      fft_pass<N, NT, 4, NS, INV, SNT>(s, tw);
      fft_passes<N, NT, NS * 4, INV, SNT>(s, tw);
    } else {
      fft_pass<N, NT, 2, NS, INV, SNT>(s, tw);
      fft_passes<N, NT, NS * 2, INV, SNT>(s, tw);
    }
  }
}

// In-place unnormalised DFT of N complex doubles resident in LDS.  Caller guarantees that the
// buffer is fully written and visible (barrier) on entry; visible on exit.  The inverse does
// NOT divide by N.
template <int N, bool INV, int NT = WH_BLOCK, int SNT = NT>
__device__ __forceinline__ void fft_lds(double2* s, const double2* tw) {
  fft_passes<N, NT, 1, INV, SNT>(s, tw);
}

// ------------------------------------------------------------------------------------------
// Real-input / real-output transforms through a half-size complex FFT
// ------------------------------------------------------------------------------------------
// Every WORLD transform is of real data or produces real data, so an N-point transform is done as an
// N/2-point complex FFT on the sample pairs (x[2j], x[2j+1]) plus one O(N) butterfly pass: half the
// butterflies and half the LDS of a complex N-point FFT.  tw_base is the context's table base: the table of
// size M (exp(-2*pi*i*k/M), k < M) lives at tw_base + M.

// Forward.  in: z[j] = (x[2j], x[2j+1]), j < N/2 (i.e. the real array itself).  out: z[k] = X[k], k = 0..N/2
// (N/2 + 1 entries).  Buffer must be visible on entry; visible on exit.
template <int N, int NT = WH_BLOCK, int SNT = NT>
__device__ __forceinline__ void rfft_lds(double2* z, const double2* __restrict__ tw_base) {
  fft_lds<N / 2, false, NT, SNT>(z, tw_base + N / 2);
  const double2* __restrict__ w = tw_base + N;
  for (int k = threadIdx.x & (NT - 1); k <= N / 4; k += NT) {
    if (k == 0) {
      const double2 a = z[0];
      z[0] = make_double2(a.x + a.y, 0.0);
      z[N / 2] = make_double2(a.x - a.y, 0.0);
    } else {
      const double2 a = z[k], b = z[N / 2 - k];
      const double er = 0.5 * (a.x + b.x), ei = 0.5 * (a.y - b.y);  // E = (A + conj(B))/2   (even samples)
      const double dr = 0.5 * (a.x - b.x), di = 0.5 * (a.y + b.y);  // D = (A - conj(B))/2 ; O = -i*D (odd samples)
      const double2 wk = w[k];
      const double tr = fma(wk.x, di, wk.y * dr);   // T = W^k * O,  O = (di, -dr)
      const double ti = fma(wk.y, di, -(wk.x * dr));
      z[k] = make_double2(er + tr, ei + ti);
      z[N / 2 - k] = make_double2(er - tr, ti - ei);  // conj(E - T)
    }
  }
  sync<SNT>();
}

// Inverse.  in: z[k] = X[k], k = 0..N/2: the half spectrum; the result is Re(IDFT) of its Hermitian extension
// (imaginary parts of the DC / Nyquist bins are ignored, as taking .real of a full complex IFFT would).
// out: z[j] = N * (x[2j], x[2j+1]), j < N/2 (unnormalised like fft_lds<.., true>: divide by N).
template <int N, int NT = WH_BLOCK, int SNT = NT>
__device__ __forceinline__ void irfft_lds(double2* z, const double2* __restrict__ tw_base) {
  const double2* __restrict__ w = tw_base + N;
  for (int k = threadIdx.x & (NT - 1); k <= N / 4; k += NT) {
    double2 a = z[k], b = z[N / 2 - k];
    if (k == 0) {  // DC and Nyquist bins: only their real parts reach a real output (Re of the inverse DFT)
      a.y = 0.0;
      b.y = 0.0;
    }
    const double er = a.x + b.x, ei = a.y - b.y;  // 2E = A + conj(B)
    const double dr = a.x - b.x, di = a.y + b.y;  // 2D = A - conj(B)
    const double2 wk = w[k];
    const double orr = fma(dr, wk.x, di * wk.y);     // 2O = 2D * conj(W^k),  conj(wk) = (wk.x, -wk.y)
    const double oi = fma(di, wk.x, -(dr * wk.y));
    z[k] = make_double2(er - oi, ei + orr);                      // Z[k]     = 2E + i*2O
    if (k != 0) z[N / 2 - k] = make_double2(er + oi, orr - ei);  // Z[N/2-k] = conj(2E) + i*conj(2O)
  }
  sync<SNT>();
  fft_lds<N / 2, true, NT, SNT>(z, tw_base + N / 2);
}

}  // namespace wh
