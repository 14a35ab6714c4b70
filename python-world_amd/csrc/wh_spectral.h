// Device routines shared by the CheapTrick and D4C kernels: pitch-synchronous sample gather,
// the "mirror the bins below f0" correction and the cumsum-based rectangular smoothing.
// Formulas follow world/cheaptrick.py:64-131 and world/d4c.py:92-110,178-233 (reference);
// the data movement (LDS staging, block scans) is this build's own.
#pragma once
#include "wh_device.h"

namespace wh {

// 1-based centre sample int(pos*fs + 0.501) + 1 (Python int() truncates toward zero).
__device__ __forceinline__ long long frame_centre(double pos, double fs) { return (long long)(pos * fs + 0.501) + 1; }

// Sample of utterance (x, n) at 1-based index clamped to [1, n]  (cheaptrick.py:89, d4c.py:98).
__device__ __forceinline__ double sample_clamped(const double* __restrict__ x, long long n, long long idx1) {
  idx1 = idx1 < 1 ? 1 : (idx1 > n ? n : idx1);
  return x[idx1 - 1];
}

// Mirror-add the bins below f0 around f0 (cheaptrick.py:67-73 with reach=f0+fs/N; d4c.py:213-220
// with reach=1.2*f0).  p[] holds bins 0..N/2 (at least every bin below `reach`) in LDS; tmp[] is
// LDS scratch of at least as many doubles as there are bins below `reach`.
// Nodes f0 - f_j (j < nlow, f_j < reach) are sorted ascending like interp1d does, queried at f_k
// with SciPy's linear kernel slope*(x-x_lo)+y_lo and end-segment extrapolation; the result is
// added to bins with f_k < f0.  Two barriers; p must be visible on entry, is visible on exit.
template <int NT = WH_BLOCK>
__device__ __forceinline__ void low_band_replica(double* p, double* tmp, int N, double fs, double f0, double reach) {
  int nlow = (int)(reach / fs * N) + 2;  // count of bins with k/N*fs < reach (monotone in k)
  if (nlow > N) nlow = N;
  while (nlow > 0 && !(((double)(nlow - 1) / N * fs) < reach)) --nlow;
  if (nlow >= 2) {
    for (int kk = threadIdx.x; kk < nlow; kk += NT) {
      const double fk = (double)kk / N * fs;
      if (fk < f0) {
        // ascending nodes a_m = f0 - f_{nlow-1-m}; hi = clamp(#nodes < fk, 1, nlow-1).  The node predicate
        // a_m < fk is monotone in m, so the count is its boundary: estimated in closed form, then settled with
        // the exact floating-point predicate (the estimate is within one of the truth).
        auto below = [&](int mm) { return (f0 - ((double)(nlow - 1 - mm) / N * fs)) < fk; };
        int cnt = (int)ceil((double)(nlow - 1) - (f0 - fk) / fs * N);
        cnt = cnt < 0 ? 0 : (cnt > nlow ? nlow : cnt);
        while (cnt > 0 && !below(cnt - 1)) --cnt;
        while (cnt < nlow && below(cnt)) ++cnt;
        const int hi = cnt < 1 ? 1 : (cnt > nlow - 1 ? nlow - 1 : cnt);
        const int lo = hi - 1;
        const double a_lo = f0 - ((double)(nlow - 1 - lo) / N * fs);
        const double a_hi = f0 - ((double)(nlow - 1 - hi) / N * fs);
        const double y_lo = p[nlow - 1 - lo];
        const double y_hi = p[nlow - 1 - hi];
        const double slope = (y_hi - y_lo) / (a_hi - a_lo);
        tmp[kk] = slope * (fk - a_lo) + y_lo;
      }
    }
  }
  sync<NT>();
  if (nlow >= 2) {
    for (int kk = threadIdx.x; kk < nlow; kk += NT) {
      const double fk = (double)kk / N * fs;
      if (fk < f0) p[kk] = tmp[kk] + p[kk];
    }
  }
  sync<NT>();
}

// The prefix-sum array is stored with one double of padding after every 8 (element i at i + i/8, N + N/8 doubles):
// the scan gives each thread a contiguous run, and with an unpadded layout the lanes of a wave would walk LDS at a
// stride of 8 doubles, 8 of them on every bank.
__device__ __forceinline__ int scan_pad(int i) { return i + (i >> 3); }

// Doubled-spectrum cumulative lookup (cheaptrick.py:103-131 / d4c.py:178-233).
// cum[scan_pad(i)], i<N: inclusive prefix sum of the Hermitian-symmetric spectrum times fs/N.
struct BandLookup {
  const double* cum;
  int N;
  double x0, dx, inv_dx, xlast, total;
  __device__ __forceinline__ void init(const double* c, int n, double fs) {
    cum = c;
    N = n;
    const double half = fs / n / 2;
    x0 = (0.0 / n * fs - fs) + half;
    const double x1 = (1.0 / n * fs - fs) + half;
    dx = x1 - x0;
    inv_dx = 1.0 / dx;  // the interpolant is continuous across bins, so a last-bit change of q is harmless
    xlast = ((double)(2 * n - 1) / n * fs - fs) + half;
    total = c[scan_pad(n - 1)];
  }
  __device__ __forceinline__ double seg(int i) const { return i < N ? cum[scan_pad(i)] : total + cum[scan_pad(i - N)]; }
  // Band mean around every bin centre: the look-up positions centre_k +- half sit at a CONSTANT fractional
  // offset from bin k (q_k = k + const), so base index and fraction are found once per frame instead of by a
  // divide/floor per bin.  (Differs from evaluating q_k per bin only by rounding of q; the interpolant is
  // continuous, so the value is unaffected.)  Requires centre +- half inside the doubled axis (always true for
  // k <= N/2 and half < fs/2 - fs/N).
  int b_lo, b_hi;
  double f_lo, f_hi;
  __device__ __forceinline__ void set_half_width(double half) {
    const double q_lo = ((0.0 - half) - x0) * inv_dx, q_hi = ((0.0 + half) - x0) * inv_dx;
    const double fl = floor(q_lo), fh = floor(q_hi);
    b_lo = (int)fl;
    b_hi = (int)fh;
    f_lo = q_lo - fl;
    f_hi = q_hi - fh;
  }
  __device__ __forceinline__ double band(int k) const {  // at(c_k + half) - at(c_k - half)
    const double l0 = seg(k + b_lo), h0 = seg(k + b_hi);
    const double lo = l0 + (seg(k + b_lo + 1) - l0) * f_lo;
    const double hi = h0 + (seg(k + b_hi + 1) - h0) * f_hi;
    return hi - lo;
  }
  __device__ __forceinline__ double at(double xi) const {
    xi = fmax(x0, fmin(xlast, xi));
    const double q = (xi - x0) * inv_dx;
    const double b = floor(q);
    const double fr = q - b;
    const int bi = (int)b;
    const double y0 = seg(bi);
    const double dy = (bi < 2 * N - 1) ? seg(bi + 1) - y0 : 0.0;
    return y0 + dy * fr;
  }
};

// p_half[0..N/2] (LDS) → cum (LDS, scan_pad layout, N + N/8 doubles) = inclusive scan of the mirrored full
// spectrum × fs/N.  Fill with thread-strided ownership (neighbouring lanes on neighbouring addresses), scan with
// thread-contiguous runs held in registers (read once, written once).  Contains barriers; p_half must be visible
// on entry; cum visible on exit.  `scratch` >= NT/64 doubles.
template <int NT, int N>
__device__ __forceinline__ void scan_mirrored(const double* p_half, double* cum, double fs, double* scratch) {
  static_assert(N % NT == 0, "scan_mirrored: N must be a multiple of the thread count");
  constexpr int PER = N / NT;
  const double df = fs / N;
  for (int i = threadIdx.x; i < N; i += NT) {
    const int k = i <= N / 2 ? i : N - i;
    cum[scan_pad(i)] = p_half[k] * df;
  }
  sync<NT>();
  const int base = threadIdx.x * PER;
  double r[PER];
  double run = 0.0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    run += cum[scan_pad(base + i)];
    r[i] = run;
  }
  const double incl = wave_scan_incl(run);
  double off = incl - run;
  if constexpr (NT > WH_WAVE) {
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 63) scratch[w] = incl;
    __syncthreads();
    for (int i = 0; i < w; ++i) off += scratch[i];
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) cum[scan_pad(base + i)] = r[i] + off;
  sync<NT>();
}

}  // namespace wh
