// StoneMask F0 refinement.
// The reference takes two zero-padded FFTs per frame and then reads at most 8 bins of them
// (world/stonemask.py:30-76).  Here only those bins are evaluated, as direct DFT sums with table twiddles (same
// maths, no FFT), so a frame costs ~L*8 complex MACs instead of 2*N*log2(N).  Two kernels:
//   stonemask_tab_kernel : eight lanes per frame, window pairs from a per-length table, nothing staged (the default
//                          for every frame whose window lies at positive times);
//   stonemask_kernel     : one wave per frame, the Blackman-windowed frame and its derivative-windowed twin staged
//                          in LDS, windows evaluated per sample — the general form, run on the frames the first
//                          kernel leaves (and on all frames when its tables would not fit LDS).
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
    const uint8_t* __restrict__ only, long long n_frames) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* sm = reinterpret_cast<double*>(smem);  // x*main window
  double* sd = sm + (2 * kmax + 1);              // x*derivative window
  const int64_t f = wh::xcd_unit(blockIdx.x, n_frames);
  if (f >= n_frames) return;
  if (only && !only[f]) return;  // second launch behind stonemask_tab_kernel: the frames it left
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

// ---- tabulated form ------------------------------------------------------------------------------------------------
// index_raw_j = (t0 + bt_k)*fs + 0.5 is never truncated before it enters the window argument
//   wt_j = (index_raw_j - 1)/fs - t0 = bt_k - 0.5/fs,   k = j - hwl
// (bt_k the reference's 4-decimal quantised times, stonemask.py:38): the frame time cancels, so the window pair of a
// frame depends on its half length alone and comes from a host-built table (row hwl at offset hwl^2); only the sample
// PICK floor(index_raw_j) depends on the frame and is evaluated per tap exactly as the reference does (the
// quantisation moves picks by up to a sample at 16 kHz).  Nothing is staged: eight lanes per frame accumulate the 2,
// then the 6, harmonic bins straight from global memory with LDS twiddles — the shape of hv_refine_row's tabulated
// path (wh_harvest.hip), where the per-wave-pass set-up and cross-lane sums are shared by 8 frames instead of being
// paid per frame by a whole wave (32 wave-wide reductions per frame were half of the staged kernel's instructions).
// Frames the table does not cover (windows reaching before the signal start, where the reference's rounding changes
// sign, or f0 below the table's floor) are flagged in `todo` and taken by stonemask_kernel in a second launch.
#ifndef WH_SM_LANES
#define WH_SM_LANES 8
#endif
constexpr int kSmLanes = WH_SM_LANES;  // lanes per frame (2 / 4 / 8 / 16: 0.48 / 0.38 / 0.31 / 0.31 ms at config 2)

template <int CTRL>
__device__ __forceinline__ double sm_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double group_sum(double v) {  // over the kSmLanes lanes of a frame
  v += sm_dpp<0xB1>(v);                        // quad_perm [1,0,3,2]
  if (kSmLanes >= 4) v += sm_dpp<0x4E>(v);     // quad_perm [2,3,0,1]
  if (kSmLanes >= 8) v += sm_dpp<0x141>(v);    // row_half_mirror
  if (kSmLanes >= 16) v += sm_dpp<0x140>(v);   // row_mirror
  return v;
}

// X[b], D[b] for NB bins, this lane's share (j = lane, lane + 4, ...), then summed over the group
template <int NB>
__device__ __forceinline__ void tab_bins(wh::ckp<const double> WH_RESTRICT xu, long long xn, double t0, double fs, int hwl, int L,
                                         wh::ckp<const double2> WH_RESTRICT wt, wh::ckp<const double> WH_RESTRICT qt, int nfft,
                                         int tw_sh, const int* bins, double2* X, double2* D, int tw_n) {
  const int lg = threadIdx.x & (kSmLanes - 1);
  int tix[NB], tstep[NB];
  const int tmask = (nfft - 1) << tw_sh;
#pragma unroll
  for (int h = 0; h < NB; ++h) {
    X[h] = make_double2(0.0, 0.0);
    D[h] = make_double2(0.0, 0.0);
    tix[h] = ((bins[h] * lg) & (nfft - 1)) << tw_sh;
    tstep[h] = ((bins[h] * kSmLanes) & (nfft - 1)) << tw_sh;
  }
  const int n_it = (L + kSmLanes - 1) / kSmLanes;
  int j = lg;
  const double xn_d = (double)xn;
  double2 cur = j < L ? wt[j] : make_double2(0.0, 0.0);
  // 1-based pick floor((t0 + bt)*fs + 0.5), clamped like the reference (stonemask.py:39-41; every tap time is
  // positive here: the kernel checked it)
  auto pick = [&](double bt) -> double {
    const double ir = fmax(1.0, fmin(xn_d, (t0 + bt) * fs + 0.5));
    return xu[(long long)ir - 1];
  };
  // The sample of a tap is addressed through its tabulated time: a chain table -> index -> waveform.  The time is
  // fetched two taps ahead and the sample one tap ahead, so neither round trip sits in front of the tap that uses it
  // (fetched in the iteration that consumed it, every tap waited for a global load).
  const double bc = j < L ? qt[j - hwl] : 0.0;
  double bn = j + kSmLanes < L ? qt[j + kSmLanes - hwl] : 0.0;
  double smp = j < L ? pick(bc) : 0.0;
  for (int it = 0; it < n_it; ++it) {
    const int jn = j + kSmLanes, jnn = j + 2 * kSmLanes;
    const double2 nxt = jn < L ? wt[jn] : make_double2(0.0, 0.0);
    const double bnn = jnn < L ? qt[jnn - hwl] : 0.0;
    const double smp_n = jn < L ? pick(bn) : 0.0;
    const double a = smp * cur.x, d = smp * cur.y;
#pragma unroll
    for (int h = 0; h < NB; ++h) {
      typedef double v2d __attribute__((ext_vector_type(2)));
#if WH_BOUNDS
      if ((unsigned)tix[h] + 16u > (unsigned)tw_n * 16u || (tix[h] & 15)) wh::oob_report(wh::WH_CK_TWIDDLE, tix[h] >> 4, tw_n);
#endif
      const v2d w = *(const v2d __attribute__((address_space(3)))*)(size_t)(uint32_t)tix[h];  // table at LDS address 0
      X[h].x = fma(a, w.x, X[h].x);
      X[h].y = fma(a, w.y, X[h].y);
      D[h].x = fma(d, w.x, D[h].x);
      D[h].y = fma(d, w.y, D[h].y);
      tix[h] = (tix[h] + tstep[h]) & tmask;
    }
    cur = nxt;
    bn = bnn;
    smp = smp_n;
    j = jn;
  }
#pragma unroll
  for (int h = 0; h < NB; ++h) {
    X[h].x = group_sum(X[h].x);
    X[h].y = group_sum(X[h].y);
    D[h].x = group_sum(D[h].x);
    D[h].y = group_sum(D[h].y);
  }
}

#ifndef WH_SM_MINW
#define WH_SM_MINW 1  // (5: 96 VGPRs but 24 spilled, 0.30 -> 0.29 ms — not taken; 6: 0.76 ms)
#endif
__global__ __launch_bounds__(256, WH_SM_MINW) void stonemask_tab_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int32_t* __restrict__ frame_utt,
    const double* __restrict__ tp, const double* __restrict__ f0_in, double* __restrict__ f0_out, double fs, int kmax,
    const double2* __restrict__ win_tab, const double* __restrict__ qtime, const double2* __restrict__ tw_base,
    int tw_n, uint8_t* __restrict__ todo, long long n_frames) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // the twiddle table of tw_n points, at LDS address 0
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
  const wh::ckp<double2> twl = wh::ck_make(reinterpret_cast<double2*>(smem), tw_n, wh::WH_CK_TWIDDLE);  // (T* unless WH_BOUNDS)
  for (int i = threadIdx.x; i < tw_n; i += 256) twl[i] = tw_base[tw_n + i];
  __syncthreads();
  constexpr int kPerBlock = 256 / kSmLanes;
  const long long n_blocks = (n_frames + kPerBlock - 1) / kPerBlock;
  const long long blk = wh::xcd_unit(blockIdx.x, n_blocks);
  if (blk >= n_blocks) return;
  const long long f = blk * kPerBlock + threadIdx.x / kSmLanes;
  const int lg = threadIdx.x & (kSmLanes - 1);
  const bool live = f < n_frames;  // the group stays together through the DPP sums
  const double f0i = live ? f0_in[f] : 0.0;
  bool work = live && f0i != 0.0;
  if (live && lg == 0) todo[f] = 0;
  if (live && f0i == 0.0 && lg == 0) f0_out[f] = f0i;
  double hwl_d = 1.0, t0 = 0.0;
  long long xn = 1;
  wh::ckp<const double> xu = wh::ck_make(x, 1, wh::WH_CK_WAVEFORM);
  if (work) {
    hwl_d = ceil(3 * fs / f0i / 2);
    t0 = tp[f];
    const int u = frame_utt[f];
    xn = x_off[u + 1] - x_off[u];
    xu = wh::ck_make(x + x_off[u], xn, wh::WH_CK_WAVEFORM);
    // the table holds windows up to kmax; every tap must sit at a positive time (the quantisation moves a tap by less
    // than a sample)
    if (!(hwl_d <= (double)kmax) || !(t0 * fs - hwl_d - 2.0 > 0.0)) {
      if (lg == 0) todo[f] = 1;
      work = false;
    }
  }
  const int hwl = work ? (int)hwl_d : 1;
  const int L = work ? 2 * hwl + 1 : 0;
  int nfft = 4;
  {
    int e = 0;
    while ((1 << e) < L) ++e;
    nfft = 1 << (e + 1);
  }
  const int tw_sh = (__ffs(tw_n) - __ffs(nfft)) + 4;  // table subsampling, and elements -> bytes
  // row hwl of the window table: 2 hwl + 1 pairs at offset hwl^2 ((kmax + 1)^2 pairs in all); the tap times: 2 kmax + 1
  const wh::ckp<const double2> wt = wh::ck_make(win_tab, (long long)(kmax + 1) * (kmax + 1), wh::WH_CK_TABLE) + (long long)hwl * hwl;
  const wh::ckp<const double> qt = wh::ck_make(qtime, 2 * kmax + 1, wh::WH_CK_TABLE) + kmax;
  int bins[6];
  double2 X[6], D[6];
  auto weighted = [&](int nbins) -> double {  // lane l evaluates bins l and l + 4, the group adds up
    double num = 0.0, den = 0.0;
#pragma unroll
    for (int q = 0; q < (6 + kSmLanes - 1) / kSmLanes; ++q) {
      const int h = lg + q * kSmLanes;
      double2 Xh = make_double2(0.0, 0.0), Dh = make_double2(0.0, 0.0);
      int bh = 0;
#pragma unroll
      for (int hh = 0; hh < 6; ++hh)
        if (hh == h) {
          Xh = X[hh];
          Dh = D[hh];
          bh = bins[hh];
        }
      if (h < nbins) {
        double p = Xh.x * Xh.x + Xh.y * Xh.y;
        if (p == 0.0) p = 2.220446049250313e-16;  // stonemask.py:54
        const double nm = Xh.x * Dh.y - Xh.y * Dh.x;
        const double inst = ((double)bh / nfft * fs) + nm / p * fs / 2 / M_PI;
        const double amp = sqrt(p);
        num += amp * inst;
        den += amp * (double)(h + 1);
      }
    }
    return group_sum(num) / group_sum(den);
  };
  // harmonics 1-2 around the initial f0 (stonemask.py:57-62)
#pragma unroll
  for (int h = 0; h < 6; ++h) bins[h] = 0;
  for (int h = 0; h < 2; ++h) bins[h] = (int)(f0i * nfft / fs * (h + 1) + 0.5);
  tab_bins<2>(xu, xn, t0, fs, hwl, L, wt, qt, nfft, tw_sh, bins, X, D, tw_n);
  const double f_first = weighted(2);
  double refined = 0.0;
  bool second = work && !(f_first < 0);
  if (second) {
#pragma unroll
    for (int h = 0; h < 6; ++h) {
      const double b = f_first * nfft / fs * (h + 1);
      bins[h] = (int)(b > 0 ? b + 0.5 : b - 0.5);
      if (bins[h] >= nfft || bins[h] < 0) second = false;  // the reference would raise IndexError: keep the input f0
    }
  }
  // (the group is uniform in `second`: every lane derived it from the same sums)
  tab_bins<6>(xu, xn, t0, fs, hwl, second ? L : 0, wt, qt, nfft, tw_sh, bins, X, D, tw_n);
  if (second) refined = weighted(6);
  else (void)weighted(6);
  if (work) {
    if (fabs(refined - f0i) / f0i > 0.2) refined = f0i;  // stonemask.py:25
    if (lg == 0) f0_out[f] = refined;
  }
}

}  // namespace

extern "C" int wh_stonemask(wh_ctx* ctx, void* stream, const wh_batch* b, const double* x, const double* tp,
                            const double* f0, double fs, const double* h_qtime, int kmax, double* refined_f0) {
  if (!ctx || !b || !x || !tp || !f0 || !h_qtime || !refined_f0) return wh::fail_msg("wh_stonemask", "null argument");
  WH_ENTER(ctx);
  if (b->total_frames == 0) return 0;
  if (kmax < 1) return wh::fail_msg("wh_stonemask", "kmax must be >= 1");
  if (int rc = wh::tables_make_room(ctx)) return rc;
  const size_t lds = sizeof(double) * 2 * (2 * (size_t)kmax + 1);
  if (lds > 160 * 1024) return wh::fail_msg("wh_stonemask", "window too long for LDS (f0 floor too low for this fs)");
  if (2 * kmax + 1 > WH_MAX_TWIDDLE / 2) return wh::fail_msg("wh_stonemask", "window longer than the largest twiddle table");
  hipStream_t st = (hipStream_t)stream;
  // quantised time table (host-built, Python string-formatting semantics — SURVEY Q2)
  std::vector<double> qt(h_qtime, h_qtime + 2 * kmax + 1);
  const double* d_qt = nullptr;
  char key[64];
  snprintf(key, sizeof key, "qtime:%.3f:%d", fs, kmax);
  if (int rc = wh::const_table(ctx, key, qt, &d_qt)) return rc;
  int32_t* err = ctx->d_flags + WH_FLAG_STONEMASK_WINDOW;
  if (int rc = wh::allow_lds(&stonemask_kernel, lds)) return rc;
  // Tabulated form first (see stonemask_tab_kernel); it needs the largest transform's twiddles in LDS.
  const uint8_t* d_only = nullptr;
#ifndef WH_STONEMASK_TABLE
#define WH_STONEMASK_TABLE 1
#endif
  int tw_n = 1;
  while (tw_n < 2 * kmax + 1) tw_n <<= 1;
  tw_n <<= 1;
  const bool use_tab = WH_STONEMASK_TABLE && tw_n <= 2048;
  if (use_tab) {
    snprintf(key, sizeof key, "smtab:%.3f:%d", fs, kmax);
    const double2* d_wtab = nullptr;
    auto it = ctx->tables.find(key);
    if (it == ctx->tables.end()) {
      std::vector<double> tab((size_t)2 * (kmax + 1) * (kmax + 1), 0.0), mw;
      const double inv_fs = 1.0 / fs;
      for (int h = 1; h <= kmax; ++h) {
        const int Lh = 2 * h + 1;
        const double wlit = (2 * (double)h + 1) / fs, two_over_wlit = 2.0 / wlit;
        mw.assign(Lh, 0.0);
        for (int j = 0; j < Lh; ++j) {
          const double wt = qt[(j - h) + kmax] - 0.5 * inv_fs;  // (idx_raw - 1)/fs - t0 with t0*fs integral
          const double c = cos(M_PI * (wt * two_over_wlit));
          mw[j] = 0.42 + 0.5 * c + 0.08 * (2 * c * c - 1);
        }
        double* row = tab.data() + 2 * (size_t)h * h;
        for (int j = 0; j < Lh; ++j) {
          const double left = j > 0 ? mw[j - 1] : 0.0, right = j + 1 < Lh ? mw[j + 1] : 0.0;
          row[2 * j] = mw[j];
          row[2 * j + 1] = -((mw[j] - left) + (right - mw[j])) / 2;  // stonemask.py:46
        }
      }
      const double* d = nullptr;
      if (int rc = wh::const_table(ctx, key, tab, &d)) return rc;
      d_wtab = reinterpret_cast<const double2*>(d);
    } else {
      d_wtab = reinterpret_cast<const double2*>(it->second);
    }
    if (int rc = wh::ws_reserve(ctx, (size_t)b->total_frames + 256)) return rc;
    uint8_t* d_todo = reinterpret_cast<uint8_t*>(ctx->ws);
    const size_t lds_tab = sizeof(double2) * (size_t)tw_n;
    const long long n_blocks = (b->total_frames + 256 / kSmLanes - 1) / (256 / kSmLanes);
    {
      wh::KernelTimer _kt(ctx, st, "stonemask_tab_kernel");
      hipLaunchKernelGGL(stonemask_tab_kernel, dim3((unsigned)wh::xcd_grid(n_blocks)), dim3(256), lds_tab, st, x, b->d_x_off,
                         b->d_frame_utt, tp, f0, refined_f0, fs, kmax, d_wtab, d_qt, ctx->d_twiddle, tw_n, d_todo,
                         (long long)b->total_frames);
    }
    WH_LAUNCH_CHECK("stonemask_tab_kernel");
    d_only = d_todo;
  }
  { wh::KernelTimer _kt(ctx, st, "stonemask_kernel"); hipLaunchKernelGGL(stonemask_kernel, dim3((unsigned)wh::xcd_grid(b->total_frames)), dim3(64), lds, st, x, b->d_x_off,
                     b->d_frame_utt, tp, f0, refined_f0, fs, d_qt, kmax, ctx->d_twiddle, err, d_only, (long long)b->total_frames); }
  WH_LAUNCH_CHECK("stonemask_kernel");
  return 0;
}
