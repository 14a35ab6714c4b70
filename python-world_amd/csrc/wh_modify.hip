// The callers either side of encode()/decode() on a device-resident batch: the parameter
// modifiers World.warp_spectrum / modify_duration (world/main.py:180-196) and the 16-bit PCM conversions of the
// reference's WAV usage (example/prosody.py:12-13,57), so that a batch goes  int16 -> encode -> modify -> decode -> int16
// without its dense tensors ever crossing PCIe.
#include <math.h>

#include "wh_device.h"
#include "wh_host.h"

namespace {

// warp_spectrum (main.py:191-196): every frame s[0..K) becomes np.interp((k/K)^factor, k/K, s).  The query points do
// not depend on the frame, so the host evaluates NumPy's search once per bin (interval index j, x - xp[j],
// xp[j+1] - xp[j], and whether NumPy returns fp[j] itself: exact knot hit or last knot) and the kernel applies
// NumPy's arithmetic  (fp[j+1]-fp[j]) / (xp[j+1]-xp[j]) * (x - xp[j]) + fp[j]  to the frame held in LDS (the update is
// in place, like `dat['spectrogram'][:] = ...`).
__global__ __launch_bounds__(256) void warp_spectrum_kernel(double* __restrict__ spec, long long n_frames, int k_bins,
                                                            const int32_t* __restrict__ src, const double* __restrict__ dx,
                                                            const double* __restrict__ den) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* row = reinterpret_cast<double*>(smem);
  const long long f = blockIdx.x;
  double* s = spec + f * k_bins;
  for (int k = threadIdx.x; k < k_bins; k += 256) row[k] = s[k];
  __syncthreads();
  for (int k = threadIdx.x; k < k_bins; k += 256) {
    const int j = src[k];
    double v = row[j];
    if (den[k] != 0.0) {  // den == 0 marks "NumPy returns fp[j]"
      const double slope = (row[j + 1] - row[j]) / den[k];
      v = slope * dx[k] + row[j];
    }
    s[k] = v;
  }
}

// modify_duration (main.py:180-189): tp <- np.interp(tp, xp_u, fp_u) with per-utterance anchor tables
// xp_u = [0, from_time..., end_u], fp_u = to_time (a trailing -1 replaced by end_u) — n_anchor points each.
__global__ __launch_bounds__(256) void modify_duration_kernel(const double* __restrict__ tp_in, double* __restrict__ tp_out,
                                                              const int64_t* __restrict__ frame_off,
                                                              const int32_t* __restrict__ frame_utt, long long n_frames,
                                                              const double* __restrict__ xp_all,
                                                              const double* __restrict__ fp_all, int n_anchor) {
  const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
  if (f >= n_frames) return;
  const int u = frame_utt[f];
  const double* xp = xp_all + (long long)u * n_anchor;
  const double* fp = fp_all + (long long)u * n_anchor;
  const double x = tp_in[f];
  double r;
  if (x > xp[n_anchor - 1]) r = fp[n_anchor - 1];
  else if (x < xp[0]) r = fp[0];
  else {
    int j = 0;  // last j with xp[j] <= x (the anchors are few: linear scan)
    for (int i = 1; i < n_anchor; ++i) j = xp[i] <= x ? i : j;
    if (j == n_anchor - 1 || xp[j] == x) r = fp[j];
    else {
      const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
      r = slope * (x - xp[j]) + fp[j];
    }
  }
  (void)frame_off;
  tp_out[f] = r;
}

// x = x_int16 / (2**15 - 1)  (example/prosody.py:13, test/speed.py:14)
__global__ __launch_bounds__(256) void pcm16_to_f64_kernel(const int16_t* __restrict__ in, long long n, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (double)in[i] / 32767.0;
}
// (out * 2**15).astype(np.int16)  (example/prosody.py:57): truncation toward zero, then the low 16 bits (what the
// conversion does on the reference's platform when |out| reaches 1.0)
__global__ __launch_bounds__(256) void f64_to_pcm16_kernel(const double* __restrict__ in, long long n, int16_t* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const double v = in[i] * 32768.0;
    const int t = (v != v) ? (int)0x80000000 : (int)fmax(-2147483648.0, fmin(2147483647.0, trunc(v)));
    out[i] = (int16_t)(t & 0xffff);
  }
}

}  // namespace

extern "C" int wh_warp_spectrum(wh_ctx* ctx, void* stream, double* spectrogram, int64_t n_frames, int k_bins,
                                const int32_t* h_src, const double* h_dx, const double* h_den) {
  if (!ctx || !spectrogram || !h_src || !h_dx || !h_den) return wh::fail_msg("wh_warp_spectrum", "null argument");
  WH_ENTER(ctx);
  if (n_frames <= 0) return 0;
  if (k_bins < 2 || k_bins > 16385) return wh::fail_msg("wh_warp_spectrum", "k_bins out of range");
  hipStream_t st = (hipStream_t)stream;
  for (int k = 0; k < k_bins; ++k)
    if (h_src[k] < 0 || h_src[k] > k_bins - 1 || (h_den[k] != 0.0 && h_src[k] > k_bins - 2))
      return wh::fail_msg("wh_warp_spectrum", "interval index outside the frame");
  std::vector<int32_t> src(h_src, h_src + k_bins);
  std::vector<double> dx(h_dx, h_dx + k_bins), den(h_den, h_den + k_bins);
  int32_t* d_src = nullptr;
  double *d_dx = nullptr, *d_den = nullptr;
  if (int rc = wh::persistent_upload(ctx, st, "warp.src", src, &d_src)) return rc;
  if (int rc = wh::persistent_upload(ctx, st, "warp.dx", dx, &d_dx)) return rc;
  if (int rc = wh::persistent_upload(ctx, st, "warp.den", den, &d_den)) return rc;
  const size_t lds = sizeof(double) * (size_t)k_bins;
  if (int rc = wh::allow_lds(&warp_spectrum_kernel, lds)) return rc;
  { wh::KernelTimer _kt(ctx, st, "warp_spectrum_kernel"); hipLaunchKernelGGL(warp_spectrum_kernel, dim3((unsigned)n_frames), dim3(256), lds, st, spectrogram, (long long)n_frames, k_bins, d_src, d_dx, d_den); }
  WH_LAUNCH_CHECK("warp_spectrum_kernel");
  return 0;
}

extern "C" int wh_modify_duration(wh_ctx* ctx, void* stream, const wh_batch* b, const double* tp_in, double* tp_out,
                                  const double* h_xp, const double* h_fp, int n_anchor) {
  if (!ctx || !b || !tp_in || !tp_out || !h_xp || !h_fp) return wh::fail_msg("wh_modify_duration", "null argument");
  WH_ENTER(ctx);
  if (n_anchor < 2) return wh::fail_msg("wh_modify_duration", "need at least two anchors per utterance");
  if (b->total_frames == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  std::vector<double> xp(h_xp, h_xp + (size_t)b->n_utt * n_anchor), fp(h_fp, h_fp + (size_t)b->n_utt * n_anchor);
  double *d_xp = nullptr, *d_fp = nullptr;
  if (int rc = wh::persistent_upload(ctx, st, "moddur.xp", xp, &d_xp)) return rc;
  if (int rc = wh::persistent_upload(ctx, st, "moddur.fp", fp, &d_fp)) return rc;
  { wh::KernelTimer _kt(ctx, st, "modify_duration_kernel"); hipLaunchKernelGGL(modify_duration_kernel, dim3((unsigned)((b->total_frames + 255) / 256)), dim3(256), 0, st, tp_in, tp_out, b->d_frame_off, b->d_frame_utt, (long long)b->total_frames, d_xp, d_fp, n_anchor); }
  WH_LAUNCH_CHECK("modify_duration_kernel");
  return 0;
}

extern "C" int wh_pcm16_to_f64(wh_ctx* ctx, void* stream, const int16_t* pcm, int64_t n, double* x) {
  if (!ctx || !pcm || !x) return wh::fail_msg("wh_pcm16_to_f64", "null argument");
  WH_ENTER(ctx);
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  { wh::KernelTimer _kt(ctx, st, "pcm16_to_f64_kernel"); hipLaunchKernelGGL(pcm16_to_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pcm, (long long)n, x); }
  WH_LAUNCH_CHECK("pcm16_to_f64_kernel");
  return 0;
}

extern "C" int wh_f64_to_pcm16(wh_ctx* ctx, void* stream, const double* y, int64_t n, int16_t* pcm) {
  if (!ctx || !y || !pcm) return wh::fail_msg("wh_f64_to_pcm16", "null argument");
  WH_ENTER(ctx);
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  { wh::KernelTimer _kt(ctx, st, "f64_to_pcm16_kernel"); hipLaunchKernelGGL(f64_to_pcm16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, (long long)n, pcm); }
  WH_LAUNCH_CHECK("f64_to_pcm16_kernel");
  return 0;
}
