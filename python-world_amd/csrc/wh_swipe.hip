// SWIPE' pitch estimator, batched.  Replaces swipe() (world/swipe.py:9-105), the third f0_method of World.encode
// (world/main.py:45-46,134-135).
//
// Per power-of-two window size (5 at 16 kHz: 2048 ... 128 samples, hop = half a window):
//   swipe_stft_kernel<WS>  : one workgroup per STFT frame: zero-padded gather, Hann window, real FFT in LDS,
//                            magnitude row [frame][WS/2+1]                    (mlab.specgram(mode='complex'), :38)
//   wh_feature_matmul      : magnitude -> ERB-spaced loudness: the cubic-spline resampling onto the ERB grid is a
//                            fixed linear map of the magnitude row (matrix from the host, SciPy's own spline), then
//                            sqrt(max(0, .)) in the epilogue                                                  (:42-44)
//   swipe_normalise_kernel : unit-norm loudness rows                                                        (:118-120)
//   wh_feature_matmul      : loudness x candidate kernels -> pitch strength [frame][candidate]              (:122-149)
//   swipe_accumulate_kernel: linear interpolation from the STFT frame times onto the output grid, weighted by the
//                            window-size membership mu of each candidate                                   (:60-67)
// then swipe_pick_kernel: per output frame, the strongest candidate and its parabolic refinement on a 1/768-octave
// grid (:69-100).  Every table that defines the estimator (windows, spline and kernel matrices, candidate set,
// refinement grids) is data supplied by the host, built with the reference's NumPy / SciPy expressions.
#include <math.h>

#include "wh_device.h"
#include "wh_host.h"


namespace {

struct SwUtt {
  int64_t x_off, n;       // samples
  int64_t seg_off, nseg;  // STFT frames of the current window size
  int64_t f_off, nf;      // output frames
};

template <int WS>
__global__ __launch_bounds__(WS >= 512 ? 256 : WS / 2) void swipe_stft_kernel(const double* __restrict__ x,
                                                                               const SwUtt* __restrict__ meta, int n_utt,
                                                                               const double* __restrict__ window, int hop,
                                                                               const double2* __restrict__ tw_base,
                                                                               double* __restrict__ mag) {
  constexpr int NT = WS >= 512 ? 256 : WS / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // WS + 2 doubles (dynamic: 64 KB + 16 B at WS = 8192, 128 KB at 16384)
  double* buf = reinterpret_cast<double*>(smem);
  const int64_t g = blockIdx.x;  // flat STFT frame
  int u = 0;
  {
    int lo = 0, hi = n_utt;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (meta[mid].seg_off <= g) lo = mid; else hi = mid;
    }
    u = lo;
  }
  const SwUtt m = meta[u];
  const int64_t s = g - m.seg_off;
  if (s >= m.nseg) return;
  const double* xu = x + m.x_off;
  const int64_t first = s * hop - WS / 2;  // xzp = [zeros(WS/2), x, zeros(...)]
  for (int i = threadIdx.x; i < WS; i += NT) {
    const int64_t k = first + i;
    buf[i] = (k >= 0 && k < m.n) ? xu[k] * window[i] : 0.0;
  }
  wh::sync<NT>();
  wh::rfft_lds<WS, NT>(reinterpret_cast<double2*>(buf), tw_base);
  const double2* z = reinterpret_cast<const double2*>(buf);
  double* out = mag + g * (WS / 2 + 1);
  for (int k = threadIdx.x; k <= WS / 2; k += NT) out[k] = sqrt(z[k].x * z[k].x + z[k].y * z[k].y);
}

__global__ __launch_bounds__(256) void swipe_normalise_kernel(double* __restrict__ L, int64_t n_rows, int n_erb) {
  __shared__ double scratch[16];
  const int64_t r = blockIdx.x;
  double* row = L + r * n_erb;
  double s = 0.0;
  for (int e = threadIdx.x; e < n_erb; e += 256) s += row[e] * row[e];
  s = wh::block_sum<256>(s, scratch);
  double den = sqrt(s);
  if (den == 0.0) den = 2.220446049250313e-16;
  for (int e = threadIdx.x; e < n_erb; e += 256) row[e] = row[e] / den;
}

// S[t][j0 + c] += mu[c] * interp1d(ti, Si[:, c])(t): linear, ti = [0, (k*hop + WS/2)/fs for k = 0 .. nseg-2]
__global__ __launch_bounds__(256) void swipe_accumulate_kernel(const SwUtt* __restrict__ meta, const double* __restrict__ si,
                                                               int n_c, int j0, const double* __restrict__ mu, int ws,
                                                               int hop, double fs, double dt, int n_cand,
                                                               double* __restrict__ S) {
  // four output frames per workgroup, one wave each (n_c is 100-200 candidates: a 256-thread workgroup per frame left
  // half of its lanes idle and made the launch 640 k tiny workgroups per window size)
  const SwUtt m = meta[blockIdx.y];
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= m.nf) return;
  const double tt = (double)t * dt;
  auto ti_at = [&](int64_t k) -> double { return k == 0 ? 0.0 : ((double)((k - 1) * hop) + ws / 2.0) / fs; };
  // SciPy's linear interp1d: hi = clip(searchsorted(ti, tt, 'left'), 1, nseg-1), lo = hi - 1 — a query that sits
  // exactly on a knot is evaluated on the interval to its LEFT (nseg >= 2 is guaranteed by the host)
  int64_t lo = 0, hi = m.nseg;  // count of knots strictly below tt
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (ti_at(mid) < tt) lo = mid + 1; else hi = mid;
  }
  double* srow = S + (m.f_off + t) * n_cand + j0;
  const bool beyond = tt > ti_at(m.nseg - 1);
  int64_t k1 = lo < 1 ? 1 : (lo > m.nseg - 1 ? m.nseg - 1 : lo);
  const int64_t k0 = k1 - 1;
  const double x0 = ti_at(k0), x1 = ti_at(k0 + 1);
  const double* r0 = si + (m.seg_off + k0) * n_c;
  const double* r1 = r0 + n_c;
  for (int c = threadIdx.x & 63; c < n_c; c += 64) {
    double v;
    if (beyond) v = NAN;  // interp1d(bounds_error=False, fill_value=nan)
    else {
      const double y0 = r0[c], y1 = r1[c];
      const double slope = (y1 - y0) / (x1 - x0);
      v = slope * (tt - x0) + y0;
    }
    srow[c] += mu[c] * v;
  }
}

// Per output frame: strongest candidate, threshold, parabolic refinement (swipe.py:69-100).
// nt[j][3] = ntc of the triple centred on candidate j; fine[j][kFine] = nftc grid of that triple, n_fine[j] its length.
constexpr int kFine = 20;
__global__ __launch_bounds__(64) void swipe_pick_kernel(const double* __restrict__ S, int64_t n_frames, int n_cand,
                                                        const double* __restrict__ pc, const double* __restrict__ nt,
                                                        const double* __restrict__ fine, const int32_t* __restrict__ n_fine,
                                                        double s_thr, double* __restrict__ f0, double* __restrict__ vuv) {
  const int64_t t = blockIdx.x;
  const double* col = S + t * n_cand;
  const int lane = threadIdx.x;
  // np.max / np.argmax semantics: NaN wins and the first NaN is the argmax, else the first maximum
  double best = -INFINITY;
  int bi = n_cand;
  bool nan_seen = false;
  for (int j = lane; j < n_cand; j += 64) {
    const double v = col[j];
    if (v != v) {
      if (!nan_seen) { nan_seen = true; bi = j; }
    } else if (!nan_seen && v > best) { best = v; bi = j; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    const int on = __shfl_xor((int)nan_seen, o, 64);
    if (on && (!nan_seen || oi < bi)) { nan_seen = true; bi = oi; best = ob; }
    else if (!on && !nan_seen && (ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; }
  }
  if (lane != 0) return;
  double p = NAN;
  const double s = nan_seen ? NAN : best;
  if (!(s < s_thr)) {
    if (bi == 0 || bi == n_cand - 1) p = pc[0];
    else {
      const double y0 = col[bi - 1], y1 = col[bi], y2 = col[bi + 1];
      const double x0 = nt[bi * 3], x1 = nt[bi * 3 + 1], x2 = nt[bi * 3 + 2];
      if (y0 == y0 && y1 == y1 && y2 == y2) {
        // the parabola through three points (np.polyfit(.., 2) on three points), Newton form
        const double d01 = (y1 - y0) / (x1 - x0), d12 = (y2 - y1) / (x2 - x1);
        const double a = (d12 - d01) / (x2 - x0);
        int kb = 0;
        double vb = -INFINITY;
        const int nf = n_fine[bi];
        for (int k = 0; k < nf; ++k) {
          const double q = fine[bi * kFine + k];
          const double v = y0 + (q - x0) * (d01 + a * (q - x1));
          if (v > vb) { vb = v; kb = k; }
        }
        p = exp2(log2(pc[bi - 1]) + (double)kb / 12 / 64);
      }
    }
  }
  const bool voiced = p == p && p > 0;
  f0[t] = voiced ? p : 0.0;
  vuv[t] = voiced ? 1.0 : 0.0;
}

template <int WS>
int launch_stft(hipStream_t st, int64_t total_seg, const double* x, const SwUtt* d_meta, int B, const double* d_win, int hop,
                const double2* tw, double* mag) {
  constexpr int NT = WS >= 512 ? 256 : WS / 2;
  const size_t lds = sizeof(double) * (WS + 2);
  if (int rc = wh::allow_lds(&swipe_stft_kernel<WS>, lds)) return rc;
  hipLaunchKernelGGL(swipe_stft_kernel<WS>, dim3((unsigned)total_seg), dim3(NT), lds, st, x, d_meta, B, d_win, hop, tw, mag);
  return 0;
}

}  // namespace

extern "C" int wh_swipe(wh_ctx* ctx, void* stream, const wh_batch* b, const double* x, double fs, double dt, double s_thr,
                        int n_cand, const double* h_pc, int n_erb, int n_win, const wh_swipe_window* h_win,
                        const double* h_ntc, const double* h_fine, const int32_t* h_n_fine, int fine_stride,
                        double* f0_out, double* vuv_out) {
  if (!ctx || !b || !x || !h_pc || !h_win || !h_ntc || !h_fine || !h_n_fine || !f0_out || !vuv_out)
    return wh::fail_msg("wh_swipe", "null argument");
  WH_ENTER(ctx);
  if (fine_stride != kFine) return wh::fail_msg("wh_swipe", "fine_stride must be 20");
  if (n_cand < 3 || n_erb < 1 || n_win < 1) return wh::fail_msg("wh_swipe", "bad table sizes");
  hipStream_t st = (hipStream_t)stream;
  const int B = b->n_utt;
  if (b->total_frames == 0) return 0;
  // workspace: S [F][n_cand], and per window size: magnitude rows, loudness rows, strength rows
  std::vector<std::vector<SwUtt>> metas(n_win, std::vector<SwUtt>(B));
  size_t max_mag = 0, max_l = 0, max_si = 0;
  for (int i = 0; i < n_win; ++i) {
    const wh_swipe_window& w = h_win[i];
    if (w.ws < 64 || w.ws > 16384 || (w.ws & (w.ws - 1)) || w.hop < 1) return wh::fail_msg("wh_swipe", "window size outside [64, 16384]");
    if (w.j0 < 0 || w.n_c < 1 || w.j0 + w.n_c > n_cand) return wh::fail_msg("wh_swipe", "candidate range outside the set");
    int64_t seg = 0;
    for (int u = 0; u < B; ++u) {
      SwUtt& m = metas[i][u];
      m.x_off = b->h_x_off[u];
      m.n = b->h_x_off[u + 1] - b->h_x_off[u];
      m.f_off = b->h_frame_off[u];
      m.nf = b->h_frame_off[u + 1] - b->h_frame_off[u];
      // len(xzp) = ws/2 + n + hop + ws/2 ; segments = (len - noverlap) // hop with noverlap = ws - hop
      m.nseg = (m.n + w.ws + w.hop - (w.ws - w.hop)) / w.hop;
      if (m.nseg < 2) return wh::fail_msg("wh_swipe", "utterance too short for the largest window");
      m.seg_off = seg;
      seg += m.nseg;
    }
    max_mag = std::max(max_mag, (size_t)seg * (w.ws / 2 + 1));
    max_l = std::max(max_l, (size_t)seg * n_erb);
    max_si = std::max(max_si, (size_t)seg * w.n_c);
  }
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t off = 0;
  const size_t o_s = off; off += al(sizeof(double) * (size_t)b->total_frames * n_cand);
  const size_t o_mag = off; off += al(sizeof(double) * max_mag);
  const size_t o_l = off; off += al(sizeof(double) * max_l);
  const size_t o_si = off; off += al(sizeof(double) * max_si);
  if (int rc = wh::ws_reserve(ctx, off)) return rc;
  char* ws = reinterpret_cast<char*>(ctx->ws);
  double* d_S = reinterpret_cast<double*>(ws + o_s);
  double* d_mag = reinterpret_cast<double*>(ws + o_mag);
  double* d_L = reinterpret_cast<double*>(ws + o_l);
  double* d_si = reinterpret_cast<double*>(ws + o_si);
  WH_CHECK(hipMemsetAsync(d_S, 0, sizeof(double) * (size_t)b->total_frames * n_cand, st));
  int64_t max_nf = 0;
  for (int u = 0; u < B; ++u) max_nf = std::max(max_nf, b->h_frame_off[u + 1] - b->h_frame_off[u]);
  for (int i = 0; i < n_win; ++i) {
    const wh_swipe_window& w = h_win[i];
    const std::string tag = std::to_string(i);
    SwUtt* d_meta = nullptr;
    double *d_win = nullptr, *d_mu = nullptr;
    if (int rc = wh::persistent_upload(ctx, st, "swipe.meta" + tag, metas[i], &d_meta)) return rc;
    std::vector<double> win(w.h_window, w.h_window + w.ws), mu(w.h_mu, w.h_mu + w.n_c);
    if (int rc = wh::persistent_upload(ctx, st, "swipe.win" + tag, win, &d_win)) return rc;
    if (int rc = wh::persistent_upload(ctx, st, "swipe.mu" + tag, mu, &d_mu)) return rc;
    const int64_t total_seg = metas[i][B - 1].seg_off + metas[i][B - 1].nseg;
    {
      wh::KernelTimer _kt(ctx, st, "swipe_stft_kernel");
      int rc_stft = 0;
      switch (w.ws) {
        case 64: rc_stft = launch_stft<64>(st, total_seg, x, d_meta, B, d_win, w.hop, ctx->d_twiddle, d_mag); break;
        case 128: rc_stft = launch_stft<128>(st, total_seg, x, d_meta, B, d_win, w.hop, ctx->d_twiddle, d_mag); break;
        case 256: rc_stft = launch_stft<256>(st, total_seg, x, d_meta, B, d_win, w.hop, ctx->d_twiddle, d_mag); break;
        case 512: rc_stft = launch_stft<512>(st, total_seg, x, d_meta, B, d_win, w.hop, ctx->d_twiddle, d_mag); break;
        case 1024: rc_stft = launch_stft<1024>(st, total_seg, x, d_meta, B, d_win, w.hop, ctx->d_twiddle, d_mag); break;
        case 2048: rc_stft = launch_stft<2048>(st, total_seg, x, d_meta, B, d_win, w.hop, ctx->d_twiddle, d_mag); break;
        case 4096: rc_stft = launch_stft<4096>(st, total_seg, x, d_meta, B, d_win, w.hop, ctx->d_twiddle, d_mag); break;
        // (8 fs / f0_floor rounds to 2^13 from 88.2 kHz up at the default floor, and for floors below ~60 Hz at 44.1 / 48 kHz)
        case 8192: rc_stft = launch_stft<8192>(st, total_seg, x, d_meta, B, d_win, w.hop, ctx->d_twiddle, d_mag); break;
        // (and to 2^14 below ~66 Hz at 88.2 / 96 kHz: a real transform of 16384 samples is the 8192-point complex one D4C runs at 96 kHz)
        default: rc_stft = launch_stft<16384>(st, total_seg, x, d_meta, B, d_win, w.hop, ctx->d_twiddle, d_mag); break;
      }
      if (rc_stft) return rc_stft;
    }
    WH_LAUNCH_CHECK("swipe_stft_kernel");
    const int nbins = w.ws / 2 + 1;
    // loudness = sqrt(max(0, spline resampling of the magnitude row))            (epilogue 3)
    const uint64_t tag_i = w.table_tag ? w.table_tag * 2 + 1 : 0, tag_k = w.table_tag ? w.table_tag * 2 + 2 : 0;
    if (int rc = wh_feature_matmul_tagged(ctx, stream, d_mag, total_seg, nbins, nbins, 0, nullptr, 1.0, w.h_interp, n_erb, 3, d_L, n_erb, tag_i)) return rc;
    { wh::KernelTimer _kt(ctx, st, "swipe_normalise_kernel"); hipLaunchKernelGGL(swipe_normalise_kernel, dim3((unsigned)total_seg), dim3(256), 0, st, d_L, total_seg, n_erb); }
    WH_LAUNCH_CHECK("swipe_normalise_kernel");
    if (int rc = wh_feature_matmul_tagged(ctx, stream, d_L, total_seg, n_erb, n_erb, 0, nullptr, 1.0, w.h_kernels, w.n_c, 0, d_si, w.n_c, tag_k)) return rc;
    { wh::KernelTimer _kt(ctx, st, "swipe_accumulate_kernel"); hipLaunchKernelGGL(swipe_accumulate_kernel, dim3((unsigned)((max_nf + 3) / 4), B), dim3(256), 0, st, d_meta, d_si, w.n_c, w.j0, d_mu, w.ws, w.hop, fs, dt, n_cand, d_S); }
    WH_LAUNCH_CHECK("swipe_accumulate_kernel");
  }
  std::vector<double> pc(h_pc, h_pc + n_cand), ntc(h_ntc, h_ntc + (size_t)n_cand * 3), fine(h_fine, h_fine + (size_t)n_cand * kFine);
  std::vector<int32_t> nfine(h_n_fine, h_n_fine + n_cand);
  double *d_pc = nullptr, *d_ntc = nullptr, *d_fine = nullptr;
  int32_t* d_nf = nullptr;
  if (int rc = wh::persistent_upload(ctx, st, "swipe.pc", pc, &d_pc)) return rc;
  if (int rc = wh::persistent_upload(ctx, st, "swipe.ntc", ntc, &d_ntc)) return rc;
  if (int rc = wh::persistent_upload(ctx, st, "swipe.fine", fine, &d_fine)) return rc;
  if (int rc = wh::persistent_upload(ctx, st, "swipe.nfine", nfine, &d_nf)) return rc;
  { wh::KernelTimer _kt(ctx, st, "swipe_pick_kernel"); hipLaunchKernelGGL(swipe_pick_kernel, dim3((unsigned)b->total_frames), dim3(64), 0, st, d_S, (int64_t)b->total_frames, n_cand, d_pc, d_ntc, d_fine, d_nf, s_thr, f0_out, vuv_out); }
  WH_LAUNCH_CHECK("swipe_pick_kernel");
  return 0;
}
