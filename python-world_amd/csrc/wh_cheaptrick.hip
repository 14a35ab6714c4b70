// CheapTrick spectral envelope — one 256-thread workgroup per frame, everything between the
// waveform gather and the final envelope stays in LDS (24*N bytes): window → FFT → power →
// low-band replica → block-scan smoothing → log → FFT → lifter → IFFT → exp.
// Replaces cheaptrick()/estimate_one_slice() of the reference (world/cheaptrick.py:9-157).
#include "wh_host.h"
#include "wh_spectral.h"

namespace {

template <int N>
__global__ __launch_bounds__(WH_BLOCK) void cheaptrick_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int32_t* __restrict__ frame_utt,
    const double* __restrict__ tp, double* __restrict__ f0_io, const double* __restrict__ vuv, double fs, double q1,
    double f0_low_limit, const double2* __restrict__ tw, double* __restrict__ spec_out, double2* __restrict__ ps_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double2* buf = reinterpret_cast<double2*>(smem);                    // N complex
  double* aux = reinterpret_cast<double*>(smem + sizeof(double2) * N);  // N real
  double* scratch = aux + N;                                            // 16 doubles
  constexpr int K = N / 2 + 1;

  const int64_t f = blockIdx.x;
  const int u = frame_utt[f];
  const double* xu = x + x_off[u];
  const long long xn = x_off[u + 1] - x_off[u];
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
  double s_w2 = 0.0;
  for (int j = threadIdx.x; j < L; j += WH_BLOCK) {
    const int rel = j - hwl;
    const double seg = wh::sample_clamped(xu, xn, centre + rel);
    const double t = (double)rel / fs / 1.5;
    const double w = 0.5 * cos(M_PI * t * f0) + 0.5;
    s_w2 += w * w;
    if (j < N) {
      buf[j].x = seg;
      aux[j] = w;
    }
  }
  const double norm = sqrt(wh::block_sum(s_w2, scratch));
  double s_sw = 0.0, s_w = 0.0;
  for (int j = threadIdx.x; j < L; j += WH_BLOCK) {
    double seg, w;
    if (j < N) {
      seg = buf[j].x;
      w = aux[j];
    } else {  // np.fft crops rows longer than N, the means still see them (Q7)
      const int rel = j - hwl;
      seg = wh::sample_clamped(xu, xn, centre + rel);
      w = 0.5 * cos(M_PI * ((double)rel / fs / 1.5) * f0) + 0.5;
    }
    w = w / norm;
    s_sw += seg * w;
    s_w += w;
    if (j < N) aux[j] = w;
  }
  wh::block_sum2(s_sw, s_w, scratch);
  const double mean_sw = s_sw / (double)L;
  const double mean_w = s_w / (double)L;
  for (int j = threadIdx.x; j < N; j += WH_BLOCK) {
    double v = 0.0;
    if (j < L) {
      const double w = aux[j];
      v = buf[j].x * w - w * mean_sw / mean_w;
    }
    buf[j] = make_double2(v, 0.0);
  }
  __syncthreads();

  // ---- power spectrum (cheaptrick.py:64-75) -------------------------------------------------
  wh::fft_lds<N, false>(buf, tw);
  if (ps_out) {
    double2* o = ps_out + f * (int64_t)N;
    for (int k = threadIdx.x; k < N; k += WH_BLOCK) o[k] = buf[k];
  }
  for (int k = threadIdx.x; k < K; k += WH_BLOCK) {
    const double2 z = buf[k];
    aux[k] = z.x * z.x + z.y * z.y;
  }
  __syncthreads();
  double* cum = reinterpret_cast<double*>(buf);  // FFT buffer is free now: N doubles of prefix sums
  wh::low_band_replica(aux, cum, N, fs, f0, f0 + fs / N);

  // ---- step 2: rectangular smoothing, width 2*f0/3 (cheaptrick.py:103-131) -------------------
  wh::scan_mirrored(aux, cum, N, fs, scratch);
  wh::BandLookup lk;
  lk.init(cum, N, fs);
  for (int k = threadIdx.x; k < K; k += WH_BLOCK) {
    const double c = (double)k / N * fs;
    const double lo = lk.at(c - f0 / 3);
    const double hi = lk.at(c + f0 / 3);
    aux[k] = (hi - lo) * 1.5 / f0;  // the reference's rand*eps dither is omitted (Q10)
  }
  __syncthreads();

  // ---- step 3: liftering in the quefrency domain (cheaptrick.py:136-157) ---------------------
  for (int n = threadIdx.x; n < N; n += WH_BLOCK) {
    const int k = n <= N / 2 ? n : N - n;
    buf[n] = make_double2(log(aux[k]), 0.0);
  }
  __syncthreads();
  wh::fft_lds<N, false>(buf, tw);
  for (int n = threadIdx.x; n < N; n += WH_BLOCK) {
    const int m = n <= N / 2 ? n : N - n;  // both lifters are mirrored about N/2
    const double q = (double)m / fs;
    double sl = 1.0;
    if (m > 0) {
      const double a = M_PI * f0 * q;
      sl = sin(a) / a;
    }
    const double cl = (1 - 2 * q1) + 2 * q1 * cos(2 * M_PI * q * f0);
    double2 z = buf[n];
    z.x = z.x * sl * cl;
    z.y = z.y * sl * cl;
    buf[n] = z;
  }
  __syncthreads();
  wh::fft_lds<N, true>(buf, tw);
  double* o = spec_out + f * (int64_t)K;
  for (int k = threadIdx.x; k < K; k += WH_BLOCK) o[k] = exp(buf[k].x / N);
}

template <int N>
int launch(wh_ctx* ctx, hipStream_t st, const wh_batch* b, const double* x, const double* tp, double* f0,
           const double* vuv, double fs, double q1, double* spec, double* ps) {
  const size_t lds = sizeof(double2) * N + sizeof(double) * (N + 16);
  const double low = fs * 3.0 / (N - 3.0);
  if (int rc = wh::allow_lds(&cheaptrick_kernel<N>, lds)) return rc;
  { wh::KernelTimer _kt(ctx, st, "cheaptrick_kernel"); hipLaunchKernelGGL(cheaptrick_kernel<N>, dim3((unsigned)b->total_frames), dim3(WH_BLOCK), lds, st, x, b->d_x_off,
                     b->d_frame_utt, tp, f0, vuv, fs, q1, low, wh::twiddle(ctx, N), spec,
                     reinterpret_cast<double2*>(ps)); }
  WH_LAUNCH_CHECK("cheaptrick_kernel");
  return 0;
}

}  // namespace

extern "C" int wh_cheaptrick(wh_ctx* ctx, void* stream, const wh_batch* b, const double* x, const double* tp,
                             double* f0, const double* vuv, double fs, int fft_size, double q1, double* spectrogram,
                             double* ps_spectrogram) {
  if (!ctx || !b || !x || !tp || !f0 || !vuv || !spectrogram) return wh::fail_msg("wh_cheaptrick", "null argument");
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
