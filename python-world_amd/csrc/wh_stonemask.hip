// StoneMask F0 refinement — one 64-lane wave per voiced frame.
// The reference takes two zero-padded FFTs per frame and then reads at most 8 bins of them
// (world/stonemask.py:30-76).  Here the Blackman-windowed frame and its derivative-windowed twin
// are staged once in LDS and only those bins are evaluated as direct DFT sums with table twiddles
// (same maths, no FFT), so a frame costs ~L*8 complex MACs instead of 2*N*log2(N).
#include "wh_host.h"
#include "wh_device.h"

namespace {

// X[b] and D[b] for NB bins of the two LDS-resident windowed sequences (length L), FFT length nfft.
template <int NB>
__device__ __forceinline__ void dft_bins(const double* __restrict__ sm, const double* __restrict__ sd, int L,
                                         int nfft, const double2* __restrict__ tw, const int* bins, double2* X,
                                         double2* D) {
#pragma unroll
  for (int h = 0; h < NB; ++h) {
    X[h] = make_double2(0.0, 0.0);
    D[h] = make_double2(0.0, 0.0);
  }
  const int lane = threadIdx.x & 63;
  for (int j = lane; j < L; j += 64) {
    const double a = sm[j], d = sd[j];
#pragma unroll
    for (int h = 0; h < NB; ++h) {
      const double2 w = tw[(int)(((long long)bins[h] * j) & (nfft - 1))];
      X[h].x += a * w.x;
      X[h].y += a * w.y;
      D[h].x += d * w.x;
      D[h].y += d * w.y;
    }
  }
#pragma unroll
  for (int h = 0; h < NB; ++h) {
    X[h].x = wh::wave_sum(X[h].x);
    X[h].y = wh::wave_sum(X[h].y);
    D[h].x = wh::wave_sum(D[h].x);
    D[h].y = wh::wave_sum(D[h].y);
  }
}

template <int NB>
__device__ __forceinline__ double weighted_if(const double2* X, const double2* D, const int* bins, int nfft,
                                              double fs) {
  double num = 0.0, den = 0.0;
#pragma unroll
  for (int h = 0; h < NB; ++h) {
    double p = X[h].x * X[h].x + X[h].y * X[h].y;
    if (p == 0.0) p = 2.220446049250313e-16;  // stonemask.py:54
    const double nm = X[h].x * D[h].y - X[h].y * D[h].x;
    const double inst = ((double)bins[h] / nfft * fs) + nm / p * fs / 2 / M_PI;
    const double amp = sqrt(p);
    num += amp * inst;
    den += amp * (double)(h + 1);
  }
  return num / den;
}

__global__ __launch_bounds__(64) void stonemask_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int32_t* __restrict__ frame_utt,
    const double* __restrict__ tp, const double* __restrict__ f0_in, double* __restrict__ f0_out, double fs,
    const double* __restrict__ qtime, int kmax, const double2* __restrict__ tw_base, int32_t* __restrict__ err,
    long long n_frames) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* sm = reinterpret_cast<double*>(smem);  // x*main window
  double* sd = sm + (2 * kmax + 1);              // x*derivative window
  const int64_t f = wh::xcd_unit(blockIdx.x, n_frames);
  if (f >= n_frames) return;
  const double f0i = f0_in[f];
  const int lane = threadIdx.x;
  if (f0i == 0.0) {
    if (lane == 0) f0_out[f] = f0i;
    return;
  }
  const double hwl_d = ceil(3 * fs / f0i / 2);
  if (!(hwl_d <= (double)kmax)) {  // table / LDS too small for this f0 (host sized it from the f0 floor)
    if (lane == 0) {
      f0_out[f] = f0i;
      atomicOr(err, 1);
    }
    return;
  }
  const int hwl = (int)hwl_d;
  const int L = 2 * hwl + 1;
  const double wlit = (2 * hwl_d + 1) / fs;
  const double inv_fs = 1.0 / fs, two_over_wlit = 2.0 / wlit;  // per-frame reciprocals instead of per-sample divides
  int nfft = 1;
  {
    int e = 0;
    while ((1 << e) < L) ++e;  // ceil(log2(L)); L is odd so never an exact power of two (except 1)
    nfft = 1 << (e + 1);
  }
  const int u = frame_utt[f];
  const double* xu = x + x_off[u];
  const long long xn = x_off[u + 1] - x_off[u];
  const double t0 = tp[f];

  // main window at the (quantised, half-sample shifted) sample times — stonemask.py:38-45 (Q1, Q2)
  auto main_at = [&](int j) -> double {
    if (j < 0 || j >= L) return 0.0;
    const double bt = qtime[(j - hwl) + kmax];
    const double v = (t0 + bt) * fs;
    const double idx_raw = v > 0 ? v + 0.5 : v - 0.5;
    const double wt = (idx_raw - 1) * inv_fs - t0;
    const double c = cospi(wt * two_over_wlit);           // cos(2*pi*wt/wlit) without the generic range reduction
    return 0.42 + 0.5 * c + 0.08 * (2 * c * c - 1);  // cos(4a) = 2cos^2(2a) - 1
  };
  double prev_last = 0.0;
  double cur = main_at(lane);
  for (int base = 0; base < L; base += 64) {
    const int j = base + lane;
    const double nxt = main_at(j + 64);
    double left = __shfl_up(cur, 1, 64);
    if (lane == 0) left = prev_last;
    double right = __shfl_down(cur, 1, 64);
    const double nxt0 = __shfl(nxt, 0, 64);
    if (lane == 63) right = nxt0;
    if (j < L) {
      const double dw = -((cur - left) + (right - cur)) / 2;  // -(diff([0,w]) + diff([w,0]))/2, stonemask.py:46
      const double bt = qtime[(j - hwl) + kmax];
      const double v = (t0 + bt) * fs;
      double idx_raw = v > 0 ? v + 0.5 : v - 0.5;
      idx_raw = fmax(1.0, fmin((double)xn, idx_raw));
      const double s = xu[(long long)idx_raw - 1];
      sm[j] = s * cur;
      sd[j] = s * dw;
    }
    prev_last = __shfl(cur, 63, 64);
    cur = nxt;
  }
  __syncthreads();

  const double2* tw = tw_base + nfft;
  int bins[6];
  double2 X[6], D[6];
  // harmonics 1-2 around the initial f0 (stonemask.py:57-62)
  for (int h = 0; h < 2; ++h) bins[h] = (int)(f0i * nfft / fs * (h + 1) + 0.5);
  dft_bins<2>(sm, sd, L, nfft, tw, bins, X, D);
  const double f_first = weighted_if<2>(X, D, bins, nfft, fs);
  double refined;
  if (f_first < 0) {
    refined = 0.0;
  } else {
    bool ok = true;
    for (int h = 0; h < 6; ++h) {
      const double b = f_first * nfft / fs * (h + 1);
      bins[h] = (int)(b > 0 ? b + 0.5 : b - 0.5);
      if (bins[h] >= nfft || bins[h] < 0) ok = false;
    }
    if (ok) {
      dft_bins<6>(sm, sd, L, nfft, tw, bins, X, D);
      refined = weighted_if<6>(X, D, bins, nfft, fs);
    } else {
      refined = 0.0;  // the reference would raise IndexError here; treated as "keep the input f0"
    }
  }
  if (fabs(refined - f0i) / f0i > 0.2) refined = f0i;  // stonemask.py:25
  if (lane == 0) f0_out[f] = refined;
}

}  // namespace

extern "C" int wh_stonemask(wh_ctx* ctx, void* stream, const wh_batch* b, const double* x, const double* tp,
                            const double* f0, double fs, const double* h_qtime, int kmax, double* refined_f0) {
  if (!ctx || !b || !x || !tp || !f0 || !h_qtime || !refined_f0) return wh::fail_msg("wh_stonemask", "null argument");
  WH_ENTER(ctx);
  if (b->total_frames == 0) return 0;
  if (kmax < 1) return wh::fail_msg("wh_stonemask", "kmax must be >= 1");
  const size_t lds = sizeof(double) * 2 * (2 * (size_t)kmax + 1);
  if (lds > 160 * 1024) return wh::fail_msg("wh_stonemask", "window too long for LDS (f0 floor too low for this fs)");
  if (2 * kmax + 1 > WH_MAX_FFT / 2) return wh::fail_msg("wh_stonemask", "window longer than the largest twiddle table");
  hipStream_t st = (hipStream_t)stream;
  // quantised time table (host-built, Python string-formatting semantics — SURVEY Q2)
  std::vector<double> qt(h_qtime, h_qtime + 2 * kmax + 1);
  const double* d_qt = nullptr;
  char key[64];
  snprintf(key, sizeof key, "qtime:%.3f:%d", fs, kmax);
  if (int rc = wh::const_table(ctx, key, qt, &d_qt)) return rc;
  int32_t* err = ctx->d_flags + WH_FLAG_STONEMASK_WINDOW;
  if (int rc = wh::allow_lds(&stonemask_kernel, lds)) return rc;
  { wh::KernelTimer _kt(ctx, st, "stonemask_kernel"); hipLaunchKernelGGL(stonemask_kernel, dim3((unsigned)wh::xcd_grid(b->total_frames)), dim3(64), lds, st, x, b->d_x_off,
                     b->d_frame_utt, tp, f0, refined_f0, fs, d_qt, kmax, ctx->d_twiddle, err, (long long)b->total_frames); }
  WH_LAUNCH_CHECK("stonemask_kernel");
  return 0;
}
