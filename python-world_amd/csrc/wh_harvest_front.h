// Harvest front end, fused: band filter -> four crossing trains -> raw F0 candidate per (1 ms frame, channel), one
// workgroup per (utterance, tile of frames) looping over the 152 channels.  Replaces the chain
//   band_tile_fft_kernel -> band_events_ols_kernel (edge lists in HBM) -> hv_raw_kernel
// whose dense edge lists and tile spectra cost 4.9 GB of HBM traffic per 64 x 10 s for 85 MB of compulsory I/O
// (profiles/hbm_traffic_cfg4_latest.txt): here the tile's spectrum lives in registers for all channels, a channel's
// filtered tile, its crossings and its interval trains never leave LDS, and what reaches HBM is the candidate row.
// Reference: CalculateRawEvent / ZeroCrossingEngine / GetF0Candidates (world/harvest.py:252-297,499-529).
//
// Tiles.  A tile OWNS the frames [k*tf, (k+1)*tf) of its utterance and filters one block of kOlsN samples centred on
// them (overlap-save: channel b's outputs are valid for samples [s0 + h_b - 1, s0 + N - h_b - 2]).  A frame's candidate
// needs, per train, the two crossing intervals around its time — up to ~1.5 periods of the band on either side — so a
// tile's block reaches a margin (2.5 periods of the LOWEST band, more for every other: their filters are shorter)
// beyond its frames.  Whether that was enough is checked, not assumed: a frame whose interval pair is not inside the
// tile's own crossings (irregular crossings next to a tile boundary; fewer than four crossings of a train in the whole
// block; an edge list that overflows) marks its utterance in `fb`, and the unfused chain — kept for that purpose, gated
// per utterance — recomputes that utterance's rows.  0 of 43 synthetic 10 s utterances and none of the speech fixtures
// take that path; the result is exact either way.
//
// Per channel and tile: spectrum product + inverse real transform as in band_events_ols_kernel (registers -> LDS, four
// radix-8/4 passes); crossings compacted into four LDS edge lists (emit_crossings_lds); then SEGMENT-major interpolation
// — one thread per pair of neighbouring intervals writes the interpolated F0 of the frames that fall between the two
// interval locations (a handful), instead of every frame searching every train: no search, two divides per interval
// instead of three per (frame, train) — and a frame pass that averages the four trains and applies the band's range
// test.  Same arithmetic as hv_raw_kernel / wh::interp_four_trains, value for value, on the same edges.
#pragma once

namespace {

struct HvTile {
  int32_t tf, ntiles;  // frames per tile, tiles of the utterance
};

constexpr int kFrN = wh::kOlsN;
constexpr int kFrNH = kFrN / 2;
#ifndef WH_HV_FRONT_ECAP
#define WH_HV_FRONT_ECAP 512
#endif
#ifndef WH_HV_FRONT_ABLATE
#define WH_HV_FRONT_ABLATE 0  // timing experiments only: 3 = stop after the transform, 1 = after the crossings, 2 = after the segments
#endif
constexpr int kFrECap = WH_HV_FRONT_ECAP;  // crossings per train and block (the 880 Hz channel: ~450 in 0.51 s; a channel
                                           // that overflows goes to the unfused chain, alone)
constexpr int kFrTFMax = 512;              // frames per tile at most: two per thread
#ifndef WH_HV_FRONT_PER
#define WH_HV_FRONT_PER 16
#endif
constexpr int kFrPer = WH_HV_FRONT_PER;    // positions per thread of the crossing pass
// filtered tile (the four trains' values alias it) | scan scratch | edge lists | frame times of the tile: 53 328 B, three
// workgroups per CU
constexpr size_t kFrLds = sizeof(double2) * (kFrNH + 1) + 64 + sizeof(double) * 4 * kFrECap + sizeof(double) * kFrTFMax;

// margin (samples) a tile keeps between its frames and the end of the lowest channel's valid outputs
// (WH_HV_FRONT_MARGIN overrides the 2.5 periods: a tuning knob — the result does not depend on it, only how many
// channels go through the unfused chain)
__host__ inline int hv_front_margin(double fs_d, double lowest_band) {
  static const double periods = getenv("WH_HV_FRONT_MARGIN") ? atof(getenv("WH_HV_FRONT_MARGIN")) : 2.5;
  return (int)ceil(periods * fs_d / lowest_band);
}

template <bool WRITE_ALL>
__global__ __launch_bounds__(256, WH_OLS_MINW) void hv_front_kernel(
    const HvUtt* __restrict__ meta, const HvTile* __restrict__ geo, const double* __restrict__ z_all, int pad, int nb,
    const int32_t* __restrict__ half, const double* __restrict__ band_f0, const double2* __restrict__ tspec,
    const double2* __restrict__ tw_base, double fs_d, double f0_floor, double f0_ceil, double* __restrict__ raw,
    uint8_t* __restrict__ live, int32_t* __restrict__ fb, int32_t* __restrict__ fbc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int N = kFrN, NH = kFrNH, KS = NH + 1;
  double2* ybuf = reinterpret_cast<double2*>(smem);
  double* sig_all = reinterpret_cast<double*>(smem);
  double* vals = reinterpret_cast<double*>(smem);  // [4][kFrTFMax], aliases the filtered tile once its crossings are out
  unsigned long long* scan_scratch = reinterpret_cast<unsigned long long*>(ybuf + KS);  // 8
  double* edges = reinterpret_cast<double*>(scan_scratch + 8);                         // [4][kFrECap]
  double* t_tab = edges + 4 * kFrECap;  // [kFrTFMax] time of frame fa + i: (double)f * 1 / 1000 is a DIVIDE, and the segment
                                        // pass compares a dozen frame times per segment
  __shared__ int s_cov[8];  // per train: first / last frame its segments cover
  const int u = blockIdx.y;
  const HvUtt m = meta[u];
  const HvTile g = geo[u];
  if ((int)blockIdx.x >= g.ntiles) return;
  const int64_t M = m.ylen;
  const int fa = blockIdx.x * g.tf;
  const int fb_end = (int64_t)fa + g.tf < m.nf1 ? fa + g.tf : (int)m.nf1;
  const int ntf = fb_end - fa;
  // the block is centred on the tile's frames
  const int64_t s0 = (int64_t)floor(0.5 * (double)(fa + fb_end - 1) * fs_d / 1000.0) - N / 2 + 1;
#pragma unroll
  for (int q = 0; q < kFrTFMax / 256; ++q) {
    const int fl = threadIdx.x + q * 256;
    t_tab[fl] = (double)(fa + fl) * 1 / 1000;  // basic_temporal_positions (harvest.py:21)
  }
  {  // Z = rfft(z[s0 .. s0 + N))
    const double* z = z_all + m.z_off;
    for (int i = threadIdx.x; i < N; i += 256) {
      const int64_t j = s0 + i + pad;
      sig_all[i] = (j >= 0 && j < M + 2 * pad) ? z[j] : 0.0;
    }
    __syncthreads();
    wh::rfft_lds<N, 256>(ybuf, tw_base);
  }
  // a thread holds the bins k = tid + 256 q (q < 4) of the two spectra together with their mirrors N/2 - k and the
  // self-paired bin N/4 (band_events_ols_kernel: product and the inverse real transform's pre-pass in registers)
  constexpr int PQ = NH / 2 / 256;
  constexpr int SLOTS = 2 * PQ + 1;
  double2 zr[SLOTS], tr[SLOTS];
#pragma unroll
  for (int q = 0; q < PQ; ++q) {
    const int k = threadIdx.x + q * 256;
    zr[2 * q] = ybuf[k];
    zr[2 * q + 1] = ybuf[NH - k];
  }
  zr[2 * PQ] = ybuf[NH / 2];
  auto load_taps = [&](const double2* src) {
    const int tid = WH_TID;  // (opaque: the eight offsets must not become loop invariants, see band_events_ols_kernel)
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
      const int k = tid + q * 256;
      tr[2 * q] = src[k];
      tr[2 * q + 1] = src[NH - k];
    }
    tr[2 * PQ] = src[NH / 2];
  };
  load_taps(tspec);
  const double half_inv_fs = 0.5 / fs_d;
  double* out_u = raw + m.f1_off * nb;
  uint8_t* live_u = live + m.f1_off * nb;
#pragma unroll 1
  for (int b = 0; b < nb; ++b) {
    bool dirty = false;
    const int h = half[b];
    // (barriers of the loop fence LDS only: what the threads hand each other is LDS; a __syncthreads would also wait
    // for the tap-spectrum prefetch and for the candidate stores of the previous channel to be acknowledged)
    wh::sync_lds<256>();  // the previous channel's values have been read out of the buffer (first round: the spectrum)
    {
      const double2* __restrict__ w = tw_base + N;
      auto fold = [&](double2 a, double2 bb, double2 wk, double2* lo, double2* hi) {
        const double er = a.x + bb.x, ei = a.y - bb.y;  // 2E = A + conj(B)
        const double dr = a.x - bb.x, di = a.y + bb.y;  // 2D = A - conj(B)
        const double orr = fma(dr, wk.x, di * wk.y);    // 2O = 2D * conj(W^k)
        const double oi = fma(di, wk.x, -(dr * wk.y));
        *lo = make_double2(er - oi, ei + orr);  // Z[k]       = 2E + i*2O
        *hi = make_double2(er + oi, orr - ei);  // Z[N/2 - k] = conj(2E) + i*conj(2O)
      };
#pragma unroll
      for (int q = 0; q < PQ; ++q) {
        const int k = threadIdx.x + q * 256;
        double2 a = wh::cmul(zr[2 * q], tr[2 * q]), bb = wh::cmul(zr[2 * q + 1], tr[2 * q + 1]);
        if (k == 0) {  // DC and Nyquist bins: only their real parts reach a real output
          a.y = 0.0;
          bb.y = 0.0;
        }
        double2 lo, hi;
        fold(a, bb, wh::ldg2(w + k), &lo, &hi);
        ybuf[k] = lo;
        if (k != 0) ybuf[NH - k] = hi;
      }
      if (threadIdx.x == 0) {
        const double2 a = wh::cmul(zr[2 * PQ], tr[2 * PQ]);
        double2 lo, hi;
        fold(a, a, wh::ldg2(w + NH / 2), &lo, &hi);
        ybuf[NH / 2] = hi;
      }
    }
    wh::sync_lds<256>();
    wh::fft_lds<NH, true, 256>(ybuf, tw_base + NH);
    if (b + 1 < nb) load_taps(tspec + (int64_t)(b + 1) * KS);  // in flight under the passes below
#if WH_HV_FRONT_ABLATE == 3
    if (threadIdx.x == 0 && sig_all[5] == 123.456) fb[u] = 1;
    continue;
#endif
    // ---- crossings of the channel's valid outputs: block index i is sample s0 + i - h - 1 ----
    // kFrPer x 256 positions centred on the tile (16: every valid output of every channel; 15 — a lane stride that is
    // conflict-free in LDS, leaving every channel but the lowest ones a margin of > 5 periods — measured no faster)
    const int64_t m_lo = s0 + h - 1, m_hi = s0 + N - h - 2;
    const int64_t c_lo = s0 + (N - kFrPer * 256) / 2;
    int64_t g0 = m_lo > c_lo ? m_lo : c_lo;
    g0 = g0 > 0 ? g0 : 0;
    int64_t g_last = m_hi - 2 < M - 2 ? m_hi - 2 : M - 2;
    g_last = g_last < g0 + kFrPer * 256 - 1 ? g_last : g0 + kFrPer * 256 - 1;
    const int n_pos = g_last >= g0 ? (int)(g_last - g0 + 1) : 0;
    const bool at_start = g0 == 0, at_end = g_last == M - 2;  // the tile's lists begin / end with the train's
    if (threadIdx.x < 8) s_cov[threadIdx.x] = threadIdx.x < 4 ? 0x7fffffff : -0x7fffffff;
    int cnt[4];
    wh::emit_crossings_lds<kFrPer>(sig_all + (g0 - s0 + h + 1), g0, n_pos, M, edges, kFrECap, cnt, scan_scratch);
    wh::sync_lds<256>();  // edge lists complete; nobody reads the filtered tile any more
#if WH_HV_FRONT_ABLATE == 1
    if (threadIdx.x == 0 && cnt[0] == 123456) fb[u] = 1;
    continue;
#endif
    // ---- segments: train k's segment j (1 <= j <= n_k - 2) lies between the locations of intervals j-1 and j ----
    bool ok = true;
    int pre[5];
    pre[0] = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ok = ok && cnt[k] >= 4 && cnt[k] <= kFrECap;  // >= 4 crossings here: the train has its 3 intervals (dio.py:159-162)
      pre[k + 1] = pre[k] + (cnt[k] > 2 ? cnt[k] - 2 : 0);
    }
    if (ok) {
      for (int sgm = threadIdx.x; sgm < pre[4]; sgm += 256) {
        const int k = (sgm >= pre[1]) + (sgm >= pre[2]) + (sgm >= pre[3]);
        const int j = sgm - pre[k] + 1;
        const double* e = edges + k * kFrECap;
        const double e0 = e[j - 1], e1 = e[j], e2 = e[j + 1];
        const double x_lo = (e0 + e1) * half_inv_fs, x_hi = (e1 + e2) * half_inv_fs;
        const double y_lo = fs_d / (e1 - e0), y_hi = fs_d / (e2 - e1);
        const double slope = (y_hi - y_lo) / (x_hi - x_lo);
        const bool first = j == 1, last = j == cnt[k] - 2;
        // frames with x_lo < t <= x_hi (lower_bound over the locations picks this segment for exactly those), clamped to
        // the tile; the first and the last segment of the whole train extrapolate (interp1d fill_value='extrapolate')
        int fl = 0, fh = ntf - 1;  // tile-local
        if (!(first && at_start)) {
          int f = (int)floor(x_lo * 1000.0) - 1 - fa;
          f = f < 0 ? 0 : (f > ntf ? ntf : f);
          while (f < ntf && t_tab[f] <= x_lo) ++f;
          fl = f;
        }
        if (!(last && at_end)) {
          int f = (int)floor(x_hi * 1000.0) + 1 - fa;
          f = f < -1 ? -1 : (f > ntf - 1 ? ntf - 1 : f);
          while (f >= 0 && t_tab[f] > x_hi) --f;
          fh = f;
        }
        if (first) s_cov[k] = fl;
        if (last) s_cov[4 + k] = fh;
        for (int f = fl; f <= fh; ++f) vals[k * kFrTFMax + f] = slope * (t_tab[f] - x_lo) + y_lo;
      }
    }
    wh::sync_lds<256>();
#if WH_HV_FRONT_ABLATE == 2
    continue;
#endif
    // ---- frames: mean of the four trains, the band's range test (harvest.py:271-276) ----
    const double bf = band_f0[b];
    double* out = out_u + (int64_t)b * m.nf1;
    uint8_t* lv = live_u + (int64_t)b * m.nf1;
#pragma unroll
    for (int q = 0; q < kFrTFMax / 256; ++q) {
      const int fl = threadIdx.x + q * 256;
      if (fl < ntf) {
        const int f = fa + fl;
        bool res = ok;
#pragma unroll
        for (int k = 0; k < 4; ++k) res = res && fl >= s_cov[k] && fl <= s_cov[4 + k];
        double cand = 0.0;
        if (res) {
          const double v0 = vals[fl], v1 = vals[kFrTFMax + fl], v2 = vals[2 * kFrTFMax + fl], v3 = vals[3 * kFrTFMax + fl];
          cand = (((v0 + v1) + v2) + v3) / 4;
          if (cand > bf * 1.1 || cand < bf * 0.9 || cand > f0_ceil || cand < f0_floor) cand = 0.0;
        } else {
          dirty = true;
        }
        if (WRITE_ALL || cand > 0) out[f] = cand;
        lv[f] = cand > 0 ? 1 : 0;
      }
    }
    if (dirty) {  // this channel of this utterance goes to the unfused chain (every writer stores the same values)
      fbc[(int64_t)u * nb + b] = 1;
      fb[u] = 1;
    }
  }
}

}  // namespace
