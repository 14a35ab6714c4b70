// Band FIR + zero-crossing event extraction kernel shared by DIO (7 Nuttall low-pass bands on the
// low-cut filtered 4 kHz signal) and Harvest (152 cosine-modulated Nuttall band-pass channels on the
// 8 kHz signal).  One workgroup per (band, utterance[, segment]): the signal is walked in 1024-sample tiles, each
// tile's FIR input window and the taps live in LDS, outputs go straight into the crossing detector
// (wh_events.h) — the filtered signal itself never touches HBM.
// Reference: get_raw_event (world/dio.py:128-140) / CalculateRawEvent (world/harvest.py:252-269).
#pragma once
#include "wh_events.h"
#include "wh_host.h"

namespace wh {

struct BandJob {
  const double* z;  // padded input signal: z[pad + m] is sample m, zeros (or filter tails) outside [0, M)
  int64_t M;        // signal length
  double* edges;    // [4][cap] fine edge positions (1-based sample units)
  int64_t cap;
  int32_t* counts;  // [4]
  // segmented mode (launch_band_events nseg > 1): the signal is cut into nseg runs of whole tiles, each with its
  // own workgroup and private lists, concatenated afterwards (band_concat_kernel)
  double* seg_edges;    // [nseg][4][seg_cap]
  int32_t* seg_counts;  // [nseg][4]
  int64_t seg_cap;
  // cursor hints of the overlap-save walker (emit_crossings_block; Harvest): hints[T * hint_stride + train], tile T = 64
  // frames = hint_spt samples, hint_tiles of them; nullptr: none
  int32_t* hints;
  int64_t hint_tiles;
  double hint_spt, hint_inv_spt;
  int32_t hint_stride;
};

constexpr int kBandTile = 1024;   // outputs per tile of the plain path (and the unit of the segment split)
constexpr int kBandTileR = 2048;  // outputs per tile of the register-tiled path: 8 consecutive outputs per thread
constexpr int kBandR = 8;
constexpr int kBandPlanes = kBandR / 2;

// Register-tiled path: the staged input window is kept as 16-byte pairs dealt round-robin to kBandPlanes planes:
// pair q = (z[2q], z[2q+1]) sits in plane q % 4 at slot q / 4.  Lane t (outputs 8t .. 8t+7) reads pair 4t + e with e
// uniform, i.e. slot t + e/4 of plane e % 4: neighbouring lanes on neighbouring 16-byte slots (no bank conflicts)
// and an address linear in t (no per-read index arithmetic, no bounds guard).
// LDS arithmetic (MI355X_MICROARCH.md): a CU has ONE LDS pipe for its four SIMDs, so a wave may spend at most a
// quarter as many LDS cycles as FP64-FMA cycles before LDS becomes the bound.  4 outputs per thread with the tap
// pair read from LDS is exactly at that limit (2 x 4 LDS cycles per 8 FMAs x 4 cycles, measured 17 TFLOP/s);
// 8 outputs per thread and the taps through the scalar unit (uniform address -> s_load -> SGPR operand of
// v_fmac_f64) is 4 LDS cycles per 64 FMA cycles.
__host__ __device__ __forceinline__ int band_plane_len(int zlen) {
  const int slots = ((zlen + 1) / 2 + kBandPlanes - 1) / kBandPlanes + 1;
  return 2 * slots + 2;  // +2 doubles: successive planes start 16 bytes further round the bank row
}
__device__ __forceinline__ int zmap(int i, int pl) {
  const int q = i >> 1;
  return (q & (kBandPlanes - 1)) * pl + ((q / kBandPlanes) << 1) + (i & 1);
}

// s[g] = sum_k taps[k] * z[(bias + 1 + g) - k], g in [0, M)
template <bool FMA>
__global__ __launch_bounds__(256) void band_events_kernel(const BandJob* __restrict__ jobs, int pad,
                                                          const double* __restrict__ taps_all,
                                                          const int32_t* __restrict__ tap_off,
                                                          const int32_t* __restrict__ tap_len,
                                                          const int32_t* __restrict__ bias, int nb,
                                                          int32_t* __restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x;
  const BandJob job = jobs[(int64_t)blockIdx.y * nb + b];
  const int lb = tap_len[b];
  // register-tiled FIR on the plane layout (odd tap counts — every filter of the reference — keep the input pairs
  // 16-byte aligned), else the plain sum on a linear layout
  const bool tiled = FMA && (lb & 1);
  const int tile = tiled ? kBandTileR : kBandTile;
  double* taps = reinterpret_cast<double*>(smem);      // lb
  double* zt = taps + ((lb + 1) & ~1);                 // tile + 2 + lb - 1 staged inputs
  const int pl = band_plane_len(kBandTileR + 2 + lb);
  double* sig = zt + (FMA ? kBandPlanes * pl : ((kBandTile + 2 + lb + 1) & ~1));  // tile + 2 filtered samples
  unsigned long long* scan_scratch = reinterpret_cast<unsigned long long*>(sig + (FMA ? kBandTileR : kBandTile) + 2);  // 8
  for (int k = threadIdx.x; k < ((lb + 1) & ~1); k += 256) taps[k] = k < lb ? taps_all[tap_off[b] + k] : 0.0;
  const double* __restrict__ tg = taps_all + tap_off[b];  // the same taps through the scalar unit (uniform address)
  int base_cnt[4] = {0, 0, 0, 0};
  const int64_t M = job.M;
  // segment blockIdx.z of gridDim.z: a run of whole kBandTile tiles with its own lists (gridDim.z == 1: the whole
  // signal, straight into the final lists)
  const int nseg = gridDim.z;
  const int64_t tiles = (M + kBandTile - 1) / kBandTile;
  const int64_t tps = (tiles + nseg - 1) / nseg;
  const int64_t t_begin = (int64_t)blockIdx.z * tps * kBandTile;
  const int64_t t_end = (t_begin + tps * kBandTile) < M ? t_begin + tps * kBandTile : M;
  double* edges_out = nseg > 1 ? job.seg_edges + (int64_t)blockIdx.z * 4 * job.seg_cap : job.edges;
  const int64_t cap_out = nseg > 1 ? job.seg_cap : job.cap;
  int32_t* counts_out = nseg > 1 ? job.seg_counts + blockIdx.z * 4 : job.counts;
  for (int64_t t0 = t_begin; t0 < t_end; t0 += tile) {
    __syncthreads();
    const int64_t zlo = t0 + bias[b] + 1 - (lb - 1);
    for (int i = threadIdx.x; i < tile + 2 + lb - 1; i += 256) {
      const int64_t j = zlo + i + pad;
      zt[tiled ? zmap(i, pl) : i] = (j >= 0 && j < M + 2 * pad) ? job.z[j] : 0.0;
    }
    __syncthreads();
    if (tiled) {
      constexpr int R = kBandR;
      const int i0 = threadIdx.x * R;
      double acc[R], r[R];
      // input index of output i0 for tap 0: top = i0 + (lb - 1), even
      const double2* zt2 = reinterpret_cast<const double2*>(zt);
      const int pl2 = pl >> 1;
      const int full = (lb - 1) >> 1;  // steps with two real taps; one last tap follows (lb is odd)
#pragma unroll
      for (int q = 0; q < R; q += 2) {  // the window of tap 0: pairs top/2 + q/2 = 4*tid + full + q/2
        const int eq = full + (q >> 1);
        const double2 v = zt2[(eq & (kBandPlanes - 1)) * pl2 + (eq >> 2) + threadIdx.x];
        acc[q] = 0.0;
        acc[q + 1] = 0.0;
        r[q] = v.x;
        r[q + 1] = v.y;
      }
      // step s consumes taps 2s, 2s+1 and the input pair top/2 - 1 - s = 4*tid + e, e = (lb-1)/2 - 1 - s >= 0
      int e = full - 1;
#pragma unroll 4
      for (int sidx = 0; sidx < full; ++sidx, --e) {
        const double tx = tg[2 * sidx], ty = tg[2 * sidx + 1];
        const double2 fresh = zt2[(e & (kBandPlanes - 1)) * pl2 + (e >> 2) + threadIdx.x];
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] = fma(tx, r[q], acc[q]);
#pragma unroll
        for (int q = R - 1; q > 0; --q) r[q] = r[q - 1];
        r[0] = fresh.y;
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] = fma(ty, r[q], acc[q]);
#pragma unroll
        for (int q = R - 1; q > 0; --q) r[q] = r[q - 1];
        r[0] = fresh.x;
      }
      {
        const double tx = tg[lb - 1];
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] = fma(tx, r[q], acc[q]);
      }
#pragma unroll
      for (int q = 0; q < R; ++q) sig[i0 + q] = t0 + i0 + q < M ? acc[q] : 0.0;
      {  // the two look-ahead samples the crossing detector needs: 128 threads each, taps strided, then reduced
        const int i = tile + (threadIdx.x >> 7);
        double a = 0.0;
        if (t0 + i < M)
          for (int k = threadIdx.x & 127; k < lb; k += 128) a = fma(taps[k], zt[zmap(i + (lb - 1) - k, pl)], a);
        a = wave_sum(a);
        if ((threadIdx.x & 63) == 0) reinterpret_cast<double*>(scan_scratch)[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x < 2) {
          const double* la = reinterpret_cast<const double*>(scan_scratch);
          sig[tile + threadIdx.x] = la[2 * threadIdx.x] + la[2 * threadIdx.x + 1];
        }
      }
    } else {
      for (int i = threadIdx.x; i < kBandTile + 2; i += 256) {
        double a = 0.0;
        if (t0 + i < M) {
          for (int k = 0; k < lb; ++k) a += taps[k] * zt[i + (lb - 1) - k];
        }
        sig[i] = a;
      }
    }
    __syncthreads();
    emit_crossings(sig, t0, M, kBandTile, edges_out, cap_out, base_cnt, scan_scratch, flags);
    if (tiled && t0 + kBandTile < t_end) {  // second half of the 2048-sample tile (its look-ahead is sig[2048..2049])
      __syncthreads();
      emit_crossings(sig + kBandTile, t0 + kBandTile, M, kBandTile, edges_out, cap_out, base_cnt, scan_scratch, flags);
    }
  }
  if (threadIdx.x < 4) counts_out[threadIdx.x] = base_cnt[threadIdx.x];
}

// Segment lists -> the final ordered lists of each (band, utterance) job.
static __global__ __launch_bounds__(256) void band_concat_kernel(const BandJob* __restrict__ jobs, int nseg,
                                                          int32_t* __restrict__ flags) {
  const BandJob job = jobs[blockIdx.x];
  for (int t = 0; t < 4; ++t) {
    int64_t off = 0;
    for (int sgm = 0; sgm < nseg; ++sgm) {
      const int c = job.seg_counts[sgm * 4 + t];
      const double* src = job.seg_edges + ((int64_t)sgm * 4 + t) * job.seg_cap;
      for (int i = threadIdx.x; i < c; i += 256)
        if (off + i < job.cap) job.edges[(int64_t)t * job.cap + off + i] = src[i];
      off += c;
    }
    if (threadIdx.x == 0) {
      if (off > job.cap) atomicOr(flags, 1);
      job.counts[t] = (int32_t)(off > job.cap ? job.cap : off);
    }
  }
}

inline int launch_band_events(wh_ctx* ctx, hipStream_t st, const BandJob* d_jobs, int nb, int n_utt, int pad,
                              const double* d_taps, const int32_t* d_tap_off, const int32_t* d_tap_len,
                              const int32_t* d_bias, int max_lb, bool use_fma, int32_t* d_flag, int nseg = 1) {
  const size_t lds_tiled = sizeof(double) * (((max_lb + 1) & ~1) + kBandPlanes * band_plane_len(kBandTileR + 2 + max_lb) +
                                             kBandTileR + 2) + 64;
  const size_t lds_plain = sizeof(double) * (((max_lb + 1) & ~1) + ((kBandTile + 2 + max_lb + 1) & ~1) + kBandTile + 2) + 64;
  const size_t lds = use_fma ? lds_tiled : lds_plain;
  if (use_fma) {
    if (int rc = allow_lds(&band_events_kernel<true>, lds)) return rc;
    { KernelTimer _kt(ctx, st, "band_events_kernel"); hipLaunchKernelGGL(band_events_kernel<true>, dim3(nb, n_utt, nseg), dim3(256), lds, st, d_jobs, pad, d_taps, d_tap_off, d_tap_len, d_bias, nb, d_flag); }
  } else {
    if (int rc = allow_lds(&band_events_kernel<false>, lds)) return rc;
    { KernelTimer _kt(ctx, st, "band_events_kernel"); hipLaunchKernelGGL(band_events_kernel<false>, dim3(nb, n_utt, nseg), dim3(256), lds, st, d_jobs, pad, d_taps, d_tap_off, d_tap_len, d_bias, nb, d_flag); }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("band_events_kernel", e);
  if (nseg > 1) {
    { KernelTimer _kt(ctx, st, "band_concat_kernel"); hipLaunchKernelGGL(band_concat_kernel, dim3(nb * n_utt), dim3(256), 0, st, d_jobs, nseg, d_flag); }
    e = hipGetLastError();
    if (e != hipSuccess) return fail("band_concat_kernel", e);
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
// Overlap-save form of the same filter bank (Harvest: 152 channels of 41-493 taps on every sample).
//
// The reference filters by FFT products over the whole utterance (world/harvest.py:262-266); the direct sums above
// cost 26 750 FMAs per sample at 8 kHz.  Here the signal is cut into tiles of kOlsValid outputs; one forward real
// FFT of kOlsN = 4096 samples per (utterance, tile) is shared by all channels (band_tile_fft_kernel, the block starts
// H = max half-length before the tile so that one block serves every channel's delay), and per channel a tile costs a
// spectrum product and one inverse real FFT (2048-point complex, four radix-8/4 passes in LDS): ~34 flops per output
// and channel instead of 82-986.  A workgroup walks the tiles of kOlsBands channels of one utterance; the tile's
// spectrum is fetched once into registers and reused for its channels, the spectra the next channel / tile needs are
// fetched under the current channel's crossing pass, and the crossings of a tile's kOlsValid outputs are extracted in
// one pass (emit_crossings_block).  The inverse transform is left unnormalised: the crossing detector only looks at
// signs and ratios.
// ------------------------------------------------------------------------------------------------------------
#ifndef WH_OLS_MINW
#define WH_OLS_MINW 3  // waves per SIMD the channel walker's register allocation leaves room for (168 VGPRs; the walker
                       // needs 169 with the thread index read opaquely — wh_harvest.hip — and ~310 without: 2 -> 4.45 ms,
                       // 3 -> 3.7 ms at config 3)
#endif
#ifndef WH_OLS_FUSED_PRE
#define WH_OLS_FUSED_PRE 1
#endif
#ifndef WH_OLS_N
#define WH_OLS_N 4096
#endif
constexpr int kOlsN = WH_OLS_N;
#ifndef WH_OLS_VALID
#define WH_OLS_VALID 3584
#endif
constexpr int kOlsValid = WH_OLS_VALID;  // outputs kept per block: 256 x 14 positions (+2 look-ahead samples); the longest
                                         // filter (493 taps) leaves 4096 - 495 = 3601
constexpr int kOlsPer = kOlsValid / 256;
#ifndef WH_OLS_BANDS
#define WH_OLS_BANDS 1  // 1 (round 6): one channel per workgroup at every batch size; 4: four; 0: by batch size (rounds 4-5)
#endif
// Channels per workgroup of band_events_ols_kernel (they share the tile spectrum of each tile): a template parameter.
// ONE (round 6, every batch size): the channel's tap spectrum stays in registers for all tiles, and with the XCD-aware
// workgroup order the tile spectra come out of the L2 the utterance's other channels share.  Four channels per workgroup
// were 1 ms faster at 1024 utterances (40.7 against 41.7 ms) but re-read their tap spectra per tile — and the 16 GB
// stream of edge stores kept pushing those (and the tile spectra) out of L2: 36.0 GB of HBM traffic per launch against
// 20.0 GB (profiles/r06_*; non-temporal edge stores stop the evictions but, scattered 8-byte writes, double the bytes
// written: 38.6 GB; collected in LDS and written as whole non-temporal runs: 20.0 GB but 44.5 ms).
constexpr int kOlsBands = 4;  // band_events_ols2_kernel (the pair variant); band_events_ols_kernel is templated on it

// T_b = rfft(taps_b zero-padded to kOlsN): [nb][kOlsN/2+1] complex (the fused front end's product), and
// R_b = the same transform of the taps rotated so that their centre tap sits at index 0: [nb][kOlsN/2+1] REAL.  A band
// filter is symmetric about its centre tap (a Nuttall window times a cosine, harvest.py:253-256), so the rotated
// sequence is even and its spectrum real — the imaginary parts are the taps' rounding asymmetry, ~1e-17 of the
// passband, and are dropped.  The channel walker multiplies the tile spectrum by R_b: a real factor per bin instead of a
// complex one (half the operands to fetch and keep in registers, a third of the product's arithmetic), and reads its
// outputs half a filter length earlier (the rotation advances the output by `half` samples).
static __global__ __launch_bounds__(256) void band_taps_fft_kernel(const double* __restrict__ taps_all,
                                                                   const int32_t* __restrict__ tap_off,
                                                                   const int32_t* __restrict__ tap_len,
                                                                   const double2* __restrict__ tw_base,
                                                                   double2* __restrict__ tspec, double* __restrict__ tre) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* buf = reinterpret_cast<double*>(smem);
  const int b = blockIdx.x;
  const int lb = tap_len[b];
  for (int i = threadIdx.x; i < kOlsN; i += 256) buf[i] = i < lb ? taps_all[tap_off[b] + i] : 0.0;
  __syncthreads();
  rfft_lds<kOlsN, 256>(reinterpret_cast<double2*>(buf), tw_base);
  const double2* z = reinterpret_cast<const double2*>(buf);
  for (int k = threadIdx.x; k <= kOlsN / 2; k += 256) tspec[(int64_t)b * (kOlsN / 2 + 1) + k] = z[k];
  if (!tre) return;
  __syncthreads();
  const int hb = (lb - 1) / 2;  // the centre tap
  for (int i = threadIdx.x; i < kOlsN; i += 256) {
    const int j = (i + hb) & (kOlsN - 1);  // rotated[i] = taps[i + hb] for i <= hb, taps[i + hb - N] for i >= N - hb
    buf[i] = j < lb && (i <= hb || i >= kOlsN - hb) ? taps_all[tap_off[b] + j] : 0.0;
  }
  __syncthreads();
  rfft_lds<kOlsN, 256>(reinterpret_cast<double2*>(buf), tw_base);
  for (int k = threadIdx.x; k <= kOlsN / 2; k += 256) tre[(int64_t)b * (kOlsN / 2 + 1) + k] = z[k].x;
}

// Z_{u,tile} = rfft(z_u[t0 - H .. t0 - H + kOlsN)), t0 = tile * kOlsValid (samples outside the padded signal are 0)
static __global__ __launch_bounds__(256) void band_tile_fft_kernel(const BandJob* __restrict__ jobs, int nb, int pad, int H,
                                                                   const int64_t* __restrict__ tile_off,
                                                                   const double2* __restrict__ tw_base,
                                                                   double2* __restrict__ zspec) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* buf = reinterpret_cast<double*>(smem);
  const int u = blockIdx.y;
  const BandJob job = jobs[(int64_t)u * nb];  // z and M are the same for every band of the utterance
  const int64_t tiles = (job.M + kOlsValid - 1) / kOlsValid;
  if ((int64_t)blockIdx.x >= tiles) return;
  const int64_t first = (int64_t)blockIdx.x * kOlsValid - H;
  for (int i = threadIdx.x; i < kOlsN; i += 256) {
    const int64_t j = first + i + pad;
    buf[i] = (j >= 0 && j < job.M + 2 * pad) ? job.z[j] : 0.0;
  }
  __syncthreads();
  rfft_lds<kOlsN, 256>(reinterpret_cast<double2*>(buf), tw_base);
  const double2* z = reinterpret_cast<const double2*>(buf);
  double2* out = zspec + (tile_off[u] + blockIdx.x) * (kOlsN / 2 + 1);
  for (int k = threadIdx.x; k <= kOlsN / 2; k += 256) out[k] = z[k];
}

#ifndef WH_OLS_XCD
#define WH_OLS_XCD 1
#endif
template <int kOlsBands>
static __global__ __launch_bounds__(256, WH_OLS_MINW) void band_events_ols_kernel(const BandJob* __restrict__ jobs, int nb, int n_utt, int n_xcd, int H,
                                                                     const int32_t* __restrict__ half,
                                                                     const double* __restrict__ tspec,
                                                                     const double2* __restrict__ zspec,
                                                                     const int64_t* __restrict__ tile_off,
                                                                     const double2* __restrict__ tw_base,
                                                                     int32_t* __restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = kOlsN / 2 + 1;           // 2049 spectrum bins (tspec: the REAL tap spectra, band_taps_fft_kernel)
  double2* ybuf = reinterpret_cast<double2*>(smem);
  double* sig_all = reinterpret_cast<double*>(smem);
  unsigned long long* scan_scratch = reinterpret_cast<unsigned long long*>(ybuf + KS + 1);  // 8
#if WH_OLS_XCD
  // Workgroup -> (utterance, channel group), XCD-aware (round 6).  Every channel group of an utterance walks the SAME tile
  // spectra (754 KB per 10 s utterance); with the groups of an utterance dealt over the whole launch (utterance-fastest
  // order, rounds 2-5) each of them fetched its own copy from HBM: 38 x 0.77 GB = 29 GB of the 46.5 GB this kernel moved
  // per 1024 utterances.  Workgroup ids go to the 8 XCDs round-robin (wh::xcd_unit), so id -> XCD id % 8, and XCD x takes
  // the utterances u = x (mod 8), all groups of one utterance on consecutive local ids: they start together, walk the
  // tiles in step, and a tile spectrum comes from HBM once and from that XCD's L2 for the other groups.  The real tap
  // spectra (2.5 MB for 152 channels, re-read per tile) fit the 4 MB L2 next to them.
  const int groups = (nb + kOlsBands - 1) / kOlsBands;
  // (n_xcd: 8, or 1 for a batch of fewer than eight utterances — one utterance per XCD would leave the other XCDs idle,
  // and what a handful of utterances re-reads fits any L2)
  const int xcd = blockIdx.x % n_xcd, local = blockIdx.x / n_xcd;
  const int u = (local / groups) * n_xcd + xcd;
  if (u >= n_utt) return;
  const int b0 = (local % groups) * kOlsBands;
#else
  // utterance-fastest workgroup order: the workgroups in flight at any time share a few channel groups, so the
  // 33 KB tap spectra they stream stay in every XCD's L2 (channel-fastest, each XCD cycled through all 5 MB of them
  // and half of the 8.6 GB requested per launch came from HBM)
  const int u = blockIdx.x;
  const int b0 = blockIdx.y * kOlsBands;
#endif
  const BandJob job0 = jobs[(int64_t)u * nb + b0];
  const int64_t M = job0.M;
  const int64_t tiles = (M + kOlsValid - 1) / kOlsValid;
  // running counts of the four crossing trains of each channel: kept in LDS between tiles so that the channel loop
  // stays a real loop (unrolled, its four inlined inverse transforms and twelve crossing passes share one register
  // allocation: 256 VGPRs + AGPRs)
  __shared__ int s_cnt[kOlsBands][4];
  if (threadIdx.x < kOlsBands * 4) s_cnt[threadIdx.x >> 2][threadIdx.x & 3] = 0;
  __syncthreads();
  // Software pipeline over (tile, channel): the tap spectrum of the next channel and, behind a tile's last channel, the
  // next tile's spectrum are fetched while the current channel's crossings are extracted — the one stretch of the
  // loop that needs few registers — so neither load's L2 latency (33 KB per channel-tile) sits in front of a product.
#if WH_OLS_FUSED_PRE
  // A thread holds the bins k = tid + 256 q (q < 4) of the two spectra TOGETHER WITH their mirrors N/2 - k (and the
  // self-paired bin N/4): the product and the real transform's pre-pass (irfft_lds: Z[k] = 2E + i 2O from Y[k] and
  // Y[N/2 - k]) then happen in registers and the buffer is written once, already as the half-size complex sequence —
  // against storing the 2049 products, a barrier, and a pre-pass that reads and rewrites them (19 of a channel-tile's 51
  // LDS stores per thread, the expensive direction on this LDS, and one of its barriers).  Same arithmetic as
  // irfft_lds, value for value.
  constexpr int NH = kOlsN / 2;
  constexpr int PQ = NH / 2 / 256;  // 4 bin pairs per thread
  constexpr int SLOTS = 2 * PQ + 1;
  double2 zr[SLOTS];
  double tr[SLOTS];
  auto load_taps = [&](double (&dst)[SLOTS], const double* src) {
    const int tid = WH_TID;
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
      const int k = tid + q * 256;
      dst[2 * q] = src[k];
      dst[2 * q + 1] = src[NH - k];
    }
    dst[2 * PQ] = src[NH / 2];
  };
  auto load_spec = [&](double2 (&dst)[SLOTS], const double2* src) {
    // (the thread index read opaquely: as loop invariants the eight per-thread offsets were kept in registers across
    // the whole walk, spilled, and every prefetch load then waited for the scratch load of its own address)
    const int tid = WH_TID;
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
      const int k = tid + q * 256;
      dst[2 * q] = src[k];
      dst[2 * q + 1] = src[NH - k];
    }
    dst[2 * PQ] = src[NH / 2];
  };
#else
  constexpr int PER = (KS + 255) / 256;  // 9 per thread
  double2 zr[PER];
  double tr[PER];
  auto load_taps = [&](double (&dst)[PER], const double* src) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int k = threadIdx.x + q * 256;
      dst[q] = k < KS ? src[k] : 0.0;
    }
  };
  auto load_spec = [&](double2 (&dst)[PER], const double2* src) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int k = threadIdx.x + q * 256;
      dst[q] = k < KS ? src[k] : make_double2(0.0, 0.0);
    }
  };
#endif
  auto scale = [](double2 z, double t) { return make_double2(z.x * t, z.y * t); };
  const int n_ch = nb - b0 < kOlsBands ? nb - b0 : kOlsBands;
  load_spec(zr, zspec + tile_off[u] * KS);
  load_taps(tr, tspec + (int64_t)b0 * KS);
#pragma unroll 1
  for (int64_t tile = 0; tile < tiles; ++tile) {
    const int64_t t0 = tile * kOlsValid;
#pragma unroll 1
    for (int g = 0; g < n_ch; ++g) {
      const int b = b0 + g;
      const BandJob job = jobs[(int64_t)u * nb + b];
      __syncthreads();  // the previous channel's crossings have been read out of the buffer
      int base_cnt[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) base_cnt[t] = s_cnt[g][t];
#if WH_OLS_FUSED_PRE
      {
        const double2* __restrict__ w = tw_base + kOlsN;
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
          double2 a = scale(zr[2 * q], tr[2 * q]), bb = scale(zr[2 * q + 1], tr[2 * q + 1]);
          if (k == 0) {  // DC and Nyquist bins: only their real parts reach a real output
            a.y = 0.0;
            bb.y = 0.0;
          }
          double2 lo, hi;
          fold(a, bb, ldg2(w + k), &lo, &hi);
          ybuf[k] = lo;
          if (k != 0) ybuf[NH - k] = hi;
        }
        if (threadIdx.x == 0) {  // k = N/4 pairs with itself; irfft_lds leaves the second of its two stores there
          const double2 a = scale(zr[2 * PQ], tr[2 * PQ]);
          double2 lo, hi;
          fold(a, a, ldg2(w + NH / 2), &lo, &hi);
          ybuf[NH / 2] = hi;
        }
      }
      sync_lds<256>();
#if WH_OLS_ABLATE != 1
      fft_lds<NH, true, 256>(ybuf, tw_base + NH);
#endif
#else
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int k = threadIdx.x + q * 256;
        if (k < KS) ybuf[k] = scale(zr[q], tr[q]);
      }
      sync_lds<256>();
      irfft_lds<kOlsN, 256>(ybuf, tw_base);
#endif
      if (g + 1 < n_ch) {
        load_taps(tr, tspec + (int64_t)(b + 1) * KS);
      } else {
        if (kOlsBands > 1) load_taps(tr, tspec + (int64_t)b0 * KS);  // (a single channel's tap spectrum never leaves the registers)
        if (tile + 1 < tiles) load_spec(zr, zspec + (tile_off[u] + tile + 1) * KS);
      }
      // output i of the block is s[t0 + i - (H + 1)]: the tile's outputs start at index H + 1 (H + h + 1 with the taps as
      // they lie; the zero-phase rotation of band_taps_fft_kernel advances the output by the half length h)
      const double* sig = sig_all + (H + 1);
#if WH_OLS_ABLATE == 1 || WH_OLS_ABLATE == 2
      __syncthreads();
      if (sig[threadIdx.x * kOlsPer] == 1.2345e300) stg(job.edges, sig[threadIdx.x]);  // (keeps what was computed alive)
#else
      emit_crossings_block<1, kOlsPer>(sig, t0, M, job.edges, job.cap, base_cnt, scan_scratch, flags, job.hints, job.hint_spt,
                                       job.hint_inv_spt, job.hint_tiles, job.hint_stride);
#endif
      __syncthreads();
      if (threadIdx.x < 4) s_cnt[g][threadIdx.x] = base_cnt[threadIdx.x];
    }
  }
  __syncthreads();
  if (threadIdx.x < kOlsBands * 4 && b0 + (threadIdx.x >> 2) < nb) {
    const BandJob job = jobs[(int64_t)u * nb + b0 + (threadIdx.x >> 2)];
    const int c = s_cnt[threadIdx.x >> 2][threadIdx.x & 3];
    job.counts[threadIdx.x & 3] = c;
    if (job.hints) {  // frame tiles that start behind the last block: everything lies in front of them
      int64_t T = (int64_t)ceil((double)(tiles * kOlsValid) / job.hint_spt);
      if ((int64_t)floor((double)T * job.hint_spt) < tiles * kOlsValid) ++T;
      for (; T < job.hint_tiles; ++T) job.hints[T * job.hint_stride + (threadIdx.x & 3)] = c;
    }
  }
}

// Two channels per inverse transform.  The filtered tiles y_a, y_b of two channels are real, so the complex sequence
// y_a + i*y_b is the inverse DFT of Y_a + i*Y_b (Hermitian extensions): ONE full-size complex inverse FFT (four
// radix-8 passes, two butterflies per thread) instead of two half-size ones with their extra real-transform pass —
// half the barrier phases per channel, and the split into the two channels is free (real and imaginary parts).  The
// crossing detector then reads one component of the interleaved buffer (stride 2).
static __global__ __launch_bounds__(256, 2) void band_events_ols2_kernel(const BandJob* __restrict__ jobs, int nb, int H,
                                                                      const int32_t* __restrict__ half,
                                                                      const double2* __restrict__ tspec,
                                                                      const double2* __restrict__ zspec,
                                                                      const int64_t* __restrict__ tile_off,
                                                                      const double2* __restrict__ tw_base,
                                                                      int32_t* __restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = kOlsN / 2 + 1;
  constexpr int PER = (KS + 255) / 256;
  double2* ybuf = reinterpret_cast<double2*>(smem);  // kOlsN complex
  const double* comp = reinterpret_cast<const double*>(smem);
  unsigned long long* scan_scratch = reinterpret_cast<unsigned long long*>(ybuf + kOlsN);  // 8
  // utterance-fastest workgroup order: the workgroups in flight at any time share a few channel groups, so the
  // 33 KB tap spectra they stream stay in every XCD's L2 (channel-fastest, each XCD cycled through all 5 MB of them
  // and half of the 8.6 GB requested per launch came from HBM)
  const int u = blockIdx.x;
  const int b0 = blockIdx.y * kOlsBands;
  const int64_t M = jobs[(int64_t)u * nb + b0].M;
  const int64_t tiles = (M + kOlsValid - 1) / kOlsValid;
  __shared__ int s_cnt[kOlsBands][4];
  if (threadIdx.x < kOlsBands * 4) s_cnt[threadIdx.x >> 2][threadIdx.x & 3] = 0;
  __syncthreads();
#pragma unroll 1
  for (int64_t tile = 0; tile < tiles; ++tile) {
    const int64_t t0 = tile * kOlsValid;
    double2 zr[PER];
    const double2* zs = zspec + (tile_off[u] + tile) * KS;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int k = threadIdx.x + q * 256;
      zr[q] = k < KS ? zs[k] : make_double2(0.0, 0.0);
    }
#pragma unroll 1
    for (int g = 0; g < kOlsBands; g += 2) {
      const int ba = b0 + g;
      if (ba >= nb) break;
      const bool has_b = ba + 1 < nb;
      const double2* ta = tspec + (int64_t)ba * KS;
      const double2* tb = tspec + (int64_t)(has_b ? ba + 1 : ba) * KS;
      __syncthreads();  // the previous pair's crossings have been read out of the buffer
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int k = threadIdx.x + q * 256;
        if (k < KS) {
          const double2 ya = cmul(zr[q], ta[k]);
          double2 yb = cmul(zr[q], tb[k]);
          if (!has_b) yb = make_double2(0.0, 0.0);
          // Y[k] = Ya[k] + i*Yb[k];  Y[N-k] = conj(Ya[k]) + i*conj(Yb[k])
          ybuf[k] = make_double2(ya.x - yb.y, ya.y + yb.x);
          if (k > 0 && k < kOlsN / 2) ybuf[kOlsN - k] = make_double2(ya.x + yb.y, yb.x - ya.y);
        }
      }
      __syncthreads();
      fft_lds<kOlsN, true, 256>(ybuf, tw_base + kOlsN);
      for (int c = 0; c < 2; ++c) {
        const int b = ba + c;
        if (b >= nb) break;
        const BandJob job = jobs[(int64_t)u * nb + b];
        int base_cnt[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) base_cnt[t] = s_cnt[g + c][t];
        // output i of the block is s[t0 + i - (H + h + 1)]; component c of complex sample i sits at comp[2 i + c]
        const double* sig = comp + 2 * (H + half[b] + 1) + c;
        __syncthreads();
        emit_crossings_block<2, kOlsPer>(sig, t0, M, job.edges, job.cap, base_cnt, scan_scratch, flags);
        __syncthreads();
        if (threadIdx.x < 4) s_cnt[g + c][threadIdx.x] = base_cnt[threadIdx.x];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < kOlsBands * 4 && b0 + (threadIdx.x >> 2) < nb)
    jobs[(int64_t)u * nb + b0 + (threadIdx.x >> 2)].counts[threadIdx.x & 3] = s_cnt[threadIdx.x >> 2][threadIdx.x & 3];
}

// ws_spec: device scratch of (n_bands + total_tiles) * (kOlsN/2+1) complex; h_tile_off[n_utt+1] tile offsets (HOST).
inline int launch_band_events_ols(wh_ctx* ctx, hipStream_t st, const BandJob* d_jobs, int nb, int n_utt, int pad, int H,
                                  const double* d_taps, const int32_t* d_tap_off, const int32_t* d_tap_len,
                                  const int32_t* d_half, const int64_t* d_tile_off, int64_t max_tiles, double2* d_tspec,
                                  double* d_tre, double2* d_zspec, int32_t* d_flag) {
  const size_t lds_fft = sizeof(double) * (kOlsN + 2);
  { KernelTimer _kt(ctx, st, "band_taps_fft_kernel"); hipLaunchKernelGGL(band_taps_fft_kernel, dim3(nb), dim3(256), lds_fft, st, d_taps, d_tap_off, d_tap_len, ctx->d_twiddle, d_tspec, d_tre); }
  { KernelTimer _kt(ctx, st, "band_tile_fft_kernel"); hipLaunchKernelGGL(band_tile_fft_kernel, dim3((unsigned)max_tiles, n_utt), dim3(256), lds_fft, st, d_jobs, nb, pad, H, d_tile_off, ctx->d_twiddle, d_zspec); }
#ifndef WH_OLS_PAIR
#define WH_OLS_PAIR 0  // measured: 7.2 ms against 6.7 ms for the one-channel-per-transform walker at config 3
#endif
#if WH_OLS_PAIR
  const size_t lds = sizeof(double2) * kOlsN + 64;
  if (int rc = allow_lds(&band_events_ols2_kernel, lds)) return rc;
  { KernelTimer _kt(ctx, st, "band_events_kernel"); hipLaunchKernelGGL(band_events_ols2_kernel, dim3(n_utt, (nb + kOlsBands - 1) / kOlsBands), dim3(256), lds, st, d_jobs, nb, H, d_half, d_tspec, d_zspec, d_tile_off, ctx->d_twiddle, d_flag); }
#else
  const size_t lds = sizeof(double2) * (kOlsN / 2 + 2) + 64;
  // one channel per workgroup while four-channel workgroups would be fewer than ~16 rounds of the chip (3 per CU): measured better at 64 and 256 utterances, worse at 1024
  const bool single = WH_OLS_BANDS == 1 || (WH_OLS_BANDS == 0 && (int64_t)n_utt * ((nb + 3) / 4) < 16 * 3 * 256);
  {
    KernelTimer _kt(ctx, st, "band_events_kernel");
#if WH_OLS_XCD
#ifndef WH_OLS_GROUP
#define WH_OLS_GROUP 4  // channels per workgroup of the large-batch form
#endif
    const int n_xcd = n_utt >= 8 ? 8 : 1;
    const unsigned u8 = (unsigned)(((n_utt + n_xcd - 1) / n_xcd) * n_xcd);
    if (single) hipLaunchKernelGGL(band_events_ols_kernel<1>, dim3(u8 * nb), dim3(256), lds, st, d_jobs, nb, n_utt, n_xcd, H, d_half, d_tre, d_zspec, d_tile_off, ctx->d_twiddle, d_flag);
    else hipLaunchKernelGGL(band_events_ols_kernel<WH_OLS_GROUP>, dim3(u8 * ((nb + WH_OLS_GROUP - 1) / WH_OLS_GROUP)), dim3(256), lds, st, d_jobs, nb, n_utt, n_xcd, H, d_half, d_tre, d_zspec, d_tile_off, ctx->d_twiddle, d_flag);
#else
    if (single) hipLaunchKernelGGL(band_events_ols_kernel<1>, dim3(n_utt, nb), dim3(256), lds, st, d_jobs, nb, n_utt, 1, H, d_half, d_tre, d_zspec, d_tile_off, ctx->d_twiddle, d_flag);
    else hipLaunchKernelGGL(band_events_ols_kernel<4>, dim3(n_utt, (nb + 3) / 4), dim3(256), lds, st, d_jobs, nb, n_utt, 1, H, d_half, d_tre, d_zspec, d_tile_off, ctx->d_twiddle, d_flag);
#endif
  }
#endif
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("band_events_ols_kernel", e);
  return 0;
}

}  // namespace wh
