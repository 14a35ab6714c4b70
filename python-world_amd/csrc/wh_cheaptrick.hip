// CheapTrick spectral envelope — one 128-thread workgroup per frame (256 from N = 2048), everything between the
// waveform gather and the final envelope stays in LDS (12*N bytes): window → real FFT → power →
// low-band replica → sliding-window smoothing → log → real FFT → lifter → inverse real FFT → exp.
// All three transforms run as N/2-point complex FFTs on sample pairs (wh_device.h: rfft_lds / irfft_lds).
// Replaces cheaptrick()/estimate_one_slice() of the reference (world/cheaptrick.py:9-157).
#include "wh_host.h"
#include "wh_math.h"
#ifndef WH_FAST_MATH64
#define WH_FAST_MATH64 1  // wh_math.h's log / exp for the 2 x 513 transcendentals of a frame (0: the device library's)
#endif
#if WH_FAST_MATH64
#define WH_CT_LOG wh::flog
#define WH_CT_EXP exp
#else
#define WH_CT_LOG log
#define WH_CT_EXP exp
#endif
#include "wh_spectral.h"

namespace {

#ifndef WH_FT_CHEAPTRICK
#define WH_FT_CHEAPTRICK 128
#endif
// Threads cooperating on one frame: 128 up to N = 1024, 256 from N = 2048 (measured on configs 2 and 5).
constexpr int ft_ct(int n) { return n >= 2048 ? 2 * WH_FT_CHEAPTRICK : WH_FT_CHEAPTRICK; }

// waves per SIMD the register allocation leaves room for: at N <= 1024 the kernel wants 84 VGPRs (5 waves); capped at
// 80 it runs 6 (1.31 -> 1.25 ms at config 2; 7 -> 72 VGPRs, 2 spilled: 1.28; 8: 1.46).  Longer transforms keep what
// they ask for.
#ifndef WH_CT_MINW
#define WH_CT_MINW 6
#endif
#ifndef WH_CT_MINW_MAXN
#define WH_CT_MINW_MAXN 1024
#endif
constexpr int ct_minw(int n) { return n <= WH_CT_MINW_MAXN ? WH_CT_MINW : 1; }
template <int N>
__global__ __launch_bounds__(ft_ct(N), ct_minw(N)) void cheaptrick_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int32_t* __restrict__ frame_utt,
    const double* __restrict__ tp, double* __restrict__ f0_io, const double* __restrict__ vuv, double fs, double q1,
    double f0_low_limit, const double2* __restrict__ tw_base, double* __restrict__ spec_out,
    double2* __restrict__ ps_out, long long n_frames) {
  constexpr int FT = ft_ct(N);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int K = N / 2 + 1;
  // (wh::ckp<T> is T* in every shipped build; the bounds build checks each access against the range named here)
  const wh::ckp<double> lds_all = wh::ck_make(reinterpret_cast<double*>(smem), (N + 2) + (K + 1) + 32, wh::WH_CK_LDS_OTHER);
  const wh::ckp<double> zr = wh::ck_sub(lds_all, 0, N + 2, wh::WH_CK_LDS_MAIN);  // the N-sample real buffer (+1 bin); also the mirrored power spectrum
  const wh::ckp<double2> zb = wh::ck_as<double2>(zr);                        // the same memory as N/2+1 complex
  const wh::ckp<double> aux = wh::ck_sub(lds_all, N + 2, K + 1, wh::WH_CK_LDS_AUX);                // K+1 reals
  const wh::ckp<double> scratch = wh::ck_sub(lds_all, (N + 2) + (K + 1), 32, wh::WH_CK_LDS_SCRATCH);  // 32 doubles
  const wh::ckp<const double2> tw = wh::ck_make(tw_base, 2 * WH_MAX_TWIDDLE, wh::WH_CK_TWIDDLE);  // the table of size n at offset n

  const int64_t f = wh::xcd_unit(blockIdx.x, n_frames);
  if (f >= n_frames) return;
  const int u = frame_utt[f];
  const long long xn = x_off[u + 1] - x_off[u];
  const wh::ckp<const double> xu = wh::ck_make(x + x_off[u], xn, wh::WH_CK_WAVEFORM);
  const double pos = tp[f];
  double f0 = f0_io[f];
  // cheaptrick.py:26-27,32-33 — default 500 Hz on unvoiced / too-low frames, written back (Q6)
  if (vuv[f] == 0.0) f0 = 500.0;
  if (f0 < f0_low_limit) f0 = 500.0;
  if (threadIdx.x == 0) f0_io[f] = f0;

  // ---- step 1: 3*T0 Hann window, L2-normalised, DC removed (cheaptrick.py:79-99) ----------
  const int hwl = (int)(1.5 * fs / f0 + 0.5);
  const int L = 2 * hwl + 1;
  const long long centre = wh::frame_centre(pos, fs);
  // per-frame constants are inverted once and multiplied in (an FP64 divide is ~12 dependent instructions; the
  // results move by an ulp)
  const double inv_span = 1.0 / fs / 1.5;
  // One walk over the window's samples and ONE block reduction: the L2 norm of the window, the mean of x*w and the
  // mean of w.  The reference normalises the window first and takes the means of the normalised products
  // (cheaptrick.py:84-95); the DC ratio mean(x w') / mean(w') is that of the unnormalised sums — the norm and the 1/L
  // cancel — so the second reduction and the pass that rescaled the stored window are gone.
  double s_w2 = 0.0, s_sw = 0.0, s_w = 0.0;
  // The thread's samples j = tid + q FT, q < N / FT, are fetched in ONE round of loads (clamped indices: always valid
  // addresses) and stay in registers for both walks; taken inside the walks — a loop of load, use, load — every row
  // cost a global round trip, twice.  Rows at or past the window's end are not used.
  constexpr int Q = N / FT;
  double xs[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) xs[q] = wh::sample_clamped(xu, xn, centre + ((int)threadIdx.x + q * FT - hwl));
  {
    // cos(pi*f0*t_j) for this thread's samples j = tid, tid + FT, ...: one sincospi for the first, a fixed rotation by
    // FT samples after that (at most N/FT + 1 steps: error growth ~1e-15) instead of a cospi per sample
    double sn, cs, rs = 0.0, rc = 1.0;
    sincospi(((double)((int)threadIdx.x - hwl) * inv_span) * f0, &sn, &cs);
    if (L > FT) sincospi(((double)FT * inv_span) * f0, &rs, &rc);  // frame-uniform
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int j = threadIdx.x + q * FT;
      if (j < L) {
        const double w = 0.5 * cs + 0.5;
        s_w2 += w * w;
        s_sw += xs[q] * w;
        s_w += w;
        zr[j] = w;
        const double cn = cs * rc - sn * rs;
        sn = sn * rc + cs * rs;
        cs = cn;
      }
    }
    for (int j = N + threadIdx.x; j < L; j += FT) {  // np.fft crops rows longer than N, the means still see them (Q7)
      const double w = 0.5 * cs + 0.5;
      const double seg = wh::sample_clamped(xu, xn, centre + (j - hwl));
      s_w2 += w * w;
      s_sw += seg * w;
      s_w += w;
      const double cn = cs * rc - sn * rs;
      sn = sn * rc + cs * rs;
      cs = cn;
    }
  }
  wh::block_sum3<FT>(s_w2, s_sw, s_w, scratch);
  const double inv_norm = 1.0 / sqrt(s_w2);
  const double dc = s_sw / s_w;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int j = threadIdx.x + q * FT;
    double v = 0.0;
    if (j < L) {
      const double w = zr[j] * inv_norm;
      v = xs[q] * w - w * dc;
    }
    zr[j] = v;
  }
  wh::sync<FT>();

  // ---- power spectrum (cheaptrick.py:64-75): real FFT through an N/2-point complex transform -------
  if (ps_out) {  // the caller wants the complex pitch-synchronous spectrum ('ps spectrogram'): materialise it
    wh::rfft_lds<N, FT>(zb, tw);
    const wh::ckp<double2> o = wh::ck_make(ps_out + f * (int64_t)N, N, wh::WH_CK_OUT);
    for (int k = threadIdx.x; k < N; k += FT) {
      const double2 z = zb[k <= N / 2 ? k : N - k];
      o[k] = k <= N / 2 ? z : make_double2(z.x, -z.y);
    }
    for (int k = threadIdx.x; k < K; k += FT) {
      const double2 z = zb[k];
      aux[k] = z.x * z.x + z.y * z.y;
    }
  } else {  // only its power is needed: the post-pass of the real transform goes straight to |X|^2 (no store of X)
    constexpr int M = N / 2;
    wh::fft_lds<M, false, FT>(zb, tw + M);
    const wh::ckp<const double2> WH_RESTRICT wtw = tw + N;
    for (int k = threadIdx.x; k <= M / 2; k += FT) {
      const double2 a = zb[k], b = zb[M - k];
      if (k == 0) {
        aux[0] = (a.x + a.y) * (a.x + a.y);
        aux[M] = (a.x - a.y) * (a.x - a.y);
      } else {
        const double er = 0.5 * (a.x + b.x), ei = 0.5 * (a.y - b.y);
        const double dr = 0.5 * (a.x - b.x), di = 0.5 * (a.y + b.y);
        const double2 wk = wh::ldg2(wtw + k);
        const double tr = fma(wk.x, di, wk.y * dr);
        const double ti = fma(wk.y, di, -(wk.x * dr));
        const double xr = er + tr, xi = ei + ti, yr = er - tr, yi = ti - ei;
        aux[k] = xr * xr + xi * xi;
        aux[M - k] = yr * yr + yi * yi;
      }
    }
  }
  wh::sync<FT>();
  wh::low_band_replica<FT>(aux, zr, N, fs, f0, f0 + fs / N);

  // ---- step 2: rectangular smoothing, width 2*f0/3 (cheaptrick.py:103-131) -------------------
  // (a sliding windowed sum per thread-owned run of bins, wh_spectral.h: BandWindow)
  wh::fill_mirrored<FT, N>(aux, zr, fs);
  {
    constexpr int KR = (K + FT - 1) / FT;
    const int k0 = threadIdx.x * KR;
    double bandv[KR];
    wh::BandWindow bw;
    bw.init(zr, N, fs, f0 / 3);
    bw.run<KR>(k0, K, bandv);
    const double scale_f0 = 1.5 / f0;
    // the reference adds rand()*eps here "to avoid log(0)" (cheaptrick.py:117, unseeded, Q10); its mean eps/2 keeps
    // that guarantee (digital silence) deterministically
#pragma unroll
    for (int r = 0; r < KR; ++r)
      if (k0 + r < K) aux[k0 + r] = WH_CT_LOG(bandv[r] * scale_f0 + 0.5 * 2.220446049250313e-16);
  }
  wh::sync<FT>();

  // ---- step 3: liftering in the quefrency domain (cheaptrick.py:136-157) ---------------------
  // The log-spectrum and both lifters are even about N/2, so the transcendental functions run on the K
  // distinct bins only and both transforms are real (forward of a real sequence, inverse to a real one).
  for (int n = threadIdx.x; n < N; n += FT) zr[n] = aux[n <= N / 2 ? n : N - n];
  const double inv_fs = 1.0 / fs;
  // the rotation by FT bins is the same for every thread: one wave evaluates it and the others read it behind the barrier
  // (a sincospi is 82 instructions whatever the number of active lanes)
  if (threadIdx.x < 64) {
    double rs0, rc0;
    sincospi(f0 * ((double)FT * inv_fs), &rs0, &rc0);
    if (threadIdx.x == 0) {
      scratch[0] = rs0;
      scratch[1] = rc0;
    }
  }
  wh::sync<FT>();
  {
    // sin(pi*f0*q_m) for m = tid, tid + FT, ...: start value + rotation by FT bins, as for the window
    double sn, cs;
    sincospi(f0 * ((double)threadIdx.x * inv_fs), &sn, &cs);
    const double rs = scratch[0], rc = scratch[1];
    for (int m = threadIdx.x; m < K; m += FT) {
      const double q = (double)m * inv_fs;
      double sl = 1.0;
      if (m > 0) sl = wh::fdiv(sn, M_PI * f0 * q);
      const double cl = (1 - 2 * q1) + 2 * q1 * (1 - 2 * sn * sn);  // cos(2*pi*q*f0) = 1 - 2 sin^2(pi*q*f0)
      aux[m] = sl * cl;
      const double cn = cs * rc - sn * rs;
      sn = sn * rc + cs * rs;
      cs = cn;
    }
  }
  // forward transform of the mirrored log spectrum (real and even, so its spectrum is real), lifter, inverse
  // transform: between the two half-size complex FFTs a thread keeps its pair of bins (k, N/2 - k) in registers through
  // the forward post-pass (real parts only), the product with the lifters and the pre-pass of the inverse real
  // transform — one pass over the half spectrum instead of three (wh::rfft_lds / multiply / wh::irfft_lds).
  {
    constexpr int M = N / 2;
#if defined(WH_CT_ABLATE_T) && WH_CT_ABLATE_T
    wh::sync<FT>();  // TIMING EXPERIMENT ONLY (wrong results): one of the two lifter transforms costs nothing — the upper bound
#else                // of halving both (real-even data: DCT-I through quarter-size transforms)
    wh::fft_lds<M, false, FT>(zb, tw + M);  // (its barriers also complete the lifter table for the loop below)
#endif
    const wh::ckp<const double2> WH_RESTRICT wtw = tw + N;
    for (int k = threadIdx.x; k <= M / 2; k += FT) {
      const double2 a = zb[k], b = zb[M - k];
      const double2 wk = wh::ldg2(wtw + k);
      double x0, x1;  // Re X[k], Re X[M-k]
      if (k == 0) {
        x0 = a.x + a.y;
        x1 = a.x - a.y;
      } else {
        const double er = 0.5 * (a.x + b.x), dr = 0.5 * (a.x - b.x), di = 0.5 * (a.y + b.y);
        const double tr = fma(wk.x, di, wk.y * dr);
        x0 = er + tr;
        x1 = er - tr;
      }
      const double A = x0 * aux[k], B = x1 * aux[M - k];
      const double er = A + B, dr = A - B;
      const double orr = dr * wk.x, oi = -(dr * wk.y);
      zb[k] = make_double2(er - oi, orr);
      if (k != 0) zb[M - k] = make_double2(er + oi, orr);
    }
    wh::sync<FT>();
    wh::fft_lds<M, true, FT>(zb, tw + M);
  }
  const wh::ckp<double> o = wh::ck_make(spec_out + f * (int64_t)K, K, wh::WH_CK_OUT);
  for (int k = threadIdx.x; k < K; k += FT) o[k] = WH_CT_EXP(zr[k] * (1.0 / N));  // N is a power of two: exact
}

template <int N>
int launch(wh_ctx* ctx, hipStream_t st, const wh_batch* b, const double* x, const double* tp, double* f0,
           const double* vuv, double fs, double q1, double* spec, double* ps) {
  const size_t lds = sizeof(double) * ((N + 2) + (N / 2 + 2) + 32);
  const double low = fs * 3.0 / (N - 3.0);
  if (int rc = wh::allow_lds(&cheaptrick_kernel<N>, lds)) return rc;
  { wh::KernelTimer _kt(ctx, st, "cheaptrick_kernel"); hipLaunchKernelGGL(cheaptrick_kernel<N>, dim3((unsigned)wh::xcd_grid(b->total_frames)), dim3(ft_ct(N)), lds, st, x, b->d_x_off,
                     b->d_frame_utt, tp, f0, vuv, fs, q1, low, ctx->d_twiddle, spec,
                     reinterpret_cast<double2*>(ps), (long long)b->total_frames); }
  WH_LAUNCH_CHECK("cheaptrick_kernel");
  return 0;
}

}  // namespace

extern "C" int wh_cheaptrick(wh_ctx* ctx, void* stream, const wh_batch* b, const double* x, const double* tp,
                             double* f0, const double* vuv, double fs, int fft_size, double q1, double* spectrogram,
                             double* ps_spectrogram) {
  if (!ctx || !b || !x || !tp || !f0 || !vuv || !spectrogram) return wh::fail_msg("wh_cheaptrick", "null argument");
  WH_ENTER(ctx);
  if (b->total_frames == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  switch (fft_size) {
    case 256: return launch<256>(ctx, st, b, x, tp, f0, vuv, fs, q1, spectrogram, ps_spectrogram);
    case 512: return launch<512>(ctx, st, b, x, tp, f0, vuv, fs, q1, spectrogram, ps_spectrogram);
    case 1024: return launch<1024>(ctx, st, b, x, tp, f0, vuv, fs, q1, spectrogram, ps_spectrogram);
    case 2048: return launch<2048>(ctx, st, b, x, tp, f0, vuv, fs, q1, spectrogram, ps_spectrogram);
    case 4096: return launch<4096>(ctx, st, b, x, tp, f0, vuv, fs, q1, spectrogram, ps_spectrogram);
    default: return wh::fail_msg("wh_cheaptrick", "fft_size must be a power of two in [256, 4096]");
  }
}
