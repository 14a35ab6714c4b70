// Device routines shared by the CheapTrick and D4C kernels: pitch-synchronous sample gather,
// the "mirror the bins below f0" correction and the cumsum-based rectangular smoothing.
// Formulas follow world/cheaptrick.py:64-131 and world/d4c.py:92-110,178-233 (reference);
// the data movement (LDS staging, sliding windows) is this build's own.
#pragma once
#include "wh_device.h"

namespace wh {

// 1-based centre sample int(pos*fs + 0.501) + 1 (Python int() truncates toward zero).
__device__ __forceinline__ long long frame_centre(double pos, double fs) { return (long long)(pos * fs + 0.501) + 1; }

// Sample of utterance (x, n) at 1-based index clamped to [1, n]  (cheaptrick.py:89, d4c.py:98).
__device__ __forceinline__ double sample_clamped(ckp<const double> WH_RESTRICT x, long long n, long long idx1) {
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
__device__ __forceinline__ void low_band_replica(ckp<double> p, ckp<double> tmp, int N, double fs, double f0, double reach) {
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
        // (bins above N/2 — reached only by an f0 within fs/N of fs/2 or beyond, which no estimator returns but a
        // caller-supplied contour can hold — are the mirror images of the stored half: never an index past it)
        const int i_lo = nlow - 1 - lo, i_hi = nlow - 1 - hi;
        const double y_lo = p[i_lo <= N / 2 ? i_lo : N - i_lo];
        const double y_hi = p[i_hi <= N / 2 ? i_hi : N - i_hi];
        const double slope = (y_hi - y_lo) / (a_hi - a_lo);
        tmp[kk] = slope * (fk - a_lo) + y_lo;
      }
    }
  }
  sync<NT>();
  if (nlow >= 2) {
    for (int kk = threadIdx.x; kk < nlow; kk += NT) {
      const double fk = (double)kk / N * fs;
      if (fk < f0 && kk <= N / 2) p[kk] = tmp[kk] + p[kk];
    }
  }
  sync<NT>();
}

// Rectangular smoothing on the doubled frequency axis (cheaptrick.py:103-131 / d4c.py:178-233).
// The reference takes the cumulative sum c of the Hermitian-symmetric spectrum s (times fs/N), interpolates it
// linearly and differences it at centre +- half.  With q = the real-valued bin index of a look-up position,
// interp(c)(q) = c[floor q] + frac(q) * s[floor q + 1], so
//     band(k) = sum_{i = k+b_lo+1}^{k+b_hi} s[i]  +  f_hi * s[k+b_hi+1]  -  f_lo * s[k+b_lo+1]
// (indices modulo N on the mirrored spectrum; b_*, f_* are the integer and fractional parts of the two look-up
// offsets, constant per frame): a plain windowed sum.  Each thread owns a run of consecutive bins and slides the
// window along it (two LDS reads per further bin), so the smoothing needs the mirrored spectrum in LDS and ONE
// barrier — not a 2048-element block prefix sum (four barriers and a shuffle scan) plus four interpolated
// look-ups per bin; and it is free of the cancellation of differencing two large cumulative values.
template <int NT, int N>
__device__ __forceinline__ void fill_mirrored(ckp<const double> p_half, ckp<double> v, double fs) {
  const double df = fs / N;
  for (int i = threadIdx.x; i < N; i += NT) v[i] = p_half[i <= N / 2 ? i : N - i] * df;
  sync<NT>();
}

struct BandWindow {
  ckp<const double> v;
  int mask, b_lo, b_hi;
  double f_lo, f_hi;
  __device__ __forceinline__ void init(ckp<const double> vv, int n, double fs, double half) {
    v = vv;
    mask = n - 1;
    const double half_bin = fs / n / 2;
    const double x0 = (0.0 / n * fs - fs) + half_bin;
    const double x1 = (1.0 / n * fs - fs) + half_bin;
    const double inv_dx = 1.0 / (x1 - x0);
    const double q_lo = ((0.0 - half) - x0) * inv_dx, q_hi = ((0.0 + half) - x0) * inv_dx;
    const double fl = floor(q_lo), fh = floor(q_hi);
    b_lo = (int)fl;
    b_hi = (int)fh;
    f_lo = q_lo - fl;
    f_hi = q_hi - fh;
  }
  // out[r] = band(k0 + r), r < R (bins at or beyond k_end are skipped)
  template <int R>
  __device__ __forceinline__ void run(int k0, int k_end, double (&out)[R]) const {
    int lo = k0 + b_lo, hi = k0 + b_hi;
    double s = 0.0;
    if (k0 < k_end) {
      // four independent partial sums: the LDS reads of the first window pipeline instead of each waiting for the
      // previous add
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int i = lo + 1;
      for (; i + 3 <= hi; i += 4) {
        s0 += v[i & mask];
        s1 += v[(i + 1) & mask];
        s2 += v[(i + 2) & mask];
        s3 += v[(i + 3) & mask];
      }
      for (; i <= hi; ++i) s0 += v[i & mask];
      s = (s0 + s1) + (s2 + s3);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (k0 + r < k_end) {
        out[r] = (s + f_hi * v[(hi + 1) & mask]) - f_lo * v[(lo + 1) & mask];
        ++lo;
        ++hi;
        s += v[hi & mask] - v[lo & mask];
      }
    }
  }
};

}  // namespace wh
