// Harvest F0 estimator, batched.  Replaces harvest() of the reference (world/harvest.py:17-54).
//
// Front end (this file):
//   hv_iir_fwd/bwd   : zero-phase Chebyshev-I decimation to ~8 kHz, SciPy filtfilt semantics (odd
//                      extension by 9, steady-state initial conditions), chunk-parallel with state warm-up
//                      (harvest.py:58-71,584-609)
//   band_events      : 152 band-pass channels + zero-crossing compaction (wh_bands.h): overlap-save FFT products
//                      (one forward transform per 4096-sample tile shared by all channels), direct FIR as fallback
//   hv_raw_kernel    : per (1 ms frame, channel) interpolation of the four interval-F0 trains (harvest.py:252-278),
//                      from (location, frequency) intervals staged per 128-frame tile
//   hv_detect_kernel : per frame: runs of >= 10 live channels -> candidate = mean (harvest.py:88-110)
//   hv_refine_kernel : every overlapped candidate (+-3 frames, harvest.py:114-125) refined by instantaneous frequency
//                      at <= 6 harmonics.  The reference farms ~178 k two-FFT calls per 10 s utterance to a process
//                      pool (harvest.py:131-211); here four lanes per candidate accumulate only the harmonic bins as
//                      direct DFT sums — window pairs from a per-length table (the frame time cancels out of the
//                      reference's window argument), twiddles from an LDS table, samples from an LDS stage of the
//                      frames a workgroup takes; a 16-lane rotation form remains for f0 floors whose tables do not
//                      fit LDS.  The candidates of a frame that share window length and harmonic bins — mostly the seven
//                      overlapped copies of one pitch track — share ONE pass over the samples (equal-key classes), every
//                      member taking its own score from the class's spectra; 24 frames per workgroup
//   hv_prune_kernel  : neighbour-frame consistency test (harvest.py:215-248), 16 frames per workgroup
// Back end (wh_harvest_contour.h): contour tracking, smoothing, 5 ms pick.
#include <math.h>
#include <hip/hip_runtime.h>

// The overlap-save walker loops over channel-tiles around an inlined inverse transform: with the plain thread index
// every per-thread LDS / twiddle address of its passes is a loop invariant that LLVM hoists and keeps alive across the
// loop (see wh_synthesis.hip).  An opaque read makes each use its own value.
__device__ __forceinline__ unsigned wh_opaque_tid() {
  unsigned t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}
#define WH_TID wh_opaque_tid()

#include "wh_bands.h"
#include "wh_device.h"
#include "wh_host.h"
#include "wh_math.h"

namespace {

constexpr int kMaxC = 15;          // int(152/10 + 0.5): candidate rows per frame before overlapping
constexpr int kRows = 7 * kMaxC;   // overlapped candidate rows (shift-major, candidate-minor)
constexpr int kFPad = 9;           // filtfilt padlen
#ifndef WH_HV_IIR_CHUNK
#define WH_HV_IIR_CHUNK 256
#endif
constexpr int kHChunk = WH_HV_IIR_CHUNK;  // filter outputs per lane of the chunked decimation IIR (+ warm-up before each chunk)
#ifndef WH_HV_WIN_TABLE
#define WH_HV_WIN_TABLE 1  // 0: always derive the refinement windows per sample (rotation + DPP neighbours)
#endif

struct HvUtt {
  int64_t x_off, n;
  int64_t nd, offset;     // constant-padded length, pad amount
  int64_t t_off;          // pass-1 output (nd + 18)
  int64_t y_off, ylen;    // decimated + trimmed signal
  int64_t z_off;          // zero-padded, mean-removed copy: ylen + 2*pad
  int64_t pick0;          // index into the filtfilt output of y[0]
  int64_t f1_off, nf1;    // 1 ms frames
  int64_t l_off, ntile;   // live-candidate bit map: word l_off + channel * ntile + tile holds the tile's 64 frames
  int64_t f_off, nf;      // output frames
};

struct Tdf2 {
  double b0, b1, b2, b3, a1, a2, a3, zi0, zi1, zi2;
};

#include "wh_harvest_contour.h"

__device__ __forceinline__ double hv_xp(const double* __restrict__ x, const HvUtt& m, int64_t i) {
  int64_t k = i - m.offset;  // constant edge padding (harvest.py:66)
  k = k < 0 ? 0 : (k > m.n - 1 ? m.n - 1 : k);
  return x[k];
}
// odd extension by 9 samples of the constant-padded signal (scipy.signal.filtfilt, padtype='odd')
__device__ __forceinline__ double hv_ext(const double* __restrict__ x, const HvUtt& m, int64_t e) {
  if (e < kFPad) return 2 * hv_xp(x, m, 0) - hv_xp(x, m, kFPad - e);
  if (e < kFPad + m.nd) return hv_xp(x, m, e - kFPad);
  return 2 * hv_xp(x, m, m.nd - 1) - hv_xp(x, m, m.nd - 2 - (e - (kFPad + m.nd)));
}

#define TDF2_STEP(IN)                     \
  {                                       \
    const double xin = (IN);              \
    yv = z0 + c.b0 * xin;                 \
    z0 = z1 + xin * c.b1 - yv * c.a1;     \
    z1 = z2 + xin * c.b2 - yv * c.a2;     \
    z2 = xin * c.b3 - yv * c.a3;          \
  }

constexpr int kHBlock = 16;  // samples fetched together, a block ahead of the recurrence (wh::serial_run)

__global__ __launch_bounds__(64) void hv_iir_fwd_kernel(const double* __restrict__ x, const HvUtt* __restrict__ meta,
                                                        Tdf2 c, int warm, double* __restrict__ tmp) {
  const HvUtt m = meta[blockIdx.y];
  const int64_t len = m.nd + 2 * kFPad;
  const int64_t s = ((int64_t)blockIdx.x * 64 + threadIdx.x) * kHChunk;
  if (s >= len) return;
  const int64_t e = s + kHChunk < len ? s + kHChunk : len;
  const double* xu = x + m.x_off;
  double* out = tmp + m.t_off;
  double z0 = 0, z1 = 0, z2 = 0, yv = 0;
  int64_t i0 = s - warm;
  if (i0 <= 0) {  // the true start: steady-state initial conditions scaled by the first sample
    i0 = 0;
    const double x0 = hv_ext(xu, m, 0);
    z0 = c.zi0 * x0;
    z1 = c.zi1 * x0;
    z2 = c.zi2 * x0;
  }
  const int64_t lo = kFPad + m.offset, hi = kFPad + m.offset + m.n;  // extended indices that are plain samples of x
  wh::serial_run<kHBlock>(
      i0, e, [&](int64_t i) { return i >= lo && i + kHBlock <= hi; }, [&](int64_t i) { return xu[i - lo]; },
      [&](int64_t i) { return hv_ext(xu, m, i); },
      [&](int64_t i, double v) {
        TDF2_STEP(v);
        if (i >= s) out[i] = yv;
      });
}
// Second pass over the reversed pass-1 output; stores only the decimated picks p = pick0 + k*r of the filtfilt result
// (quotient and remainder of p - pick0 by r are carried along the walk instead of divided out per sample).
__global__ __launch_bounds__(64) void hv_iir_bwd_kernel(const HvUtt* __restrict__ meta, Tdf2 c, int warm, int r,
                                                        const double* __restrict__ tmp, double* __restrict__ y) {
  const HvUtt m = meta[blockIdx.y];
  const int64_t len = m.nd + 2 * kFPad;
  const int64_t s = ((int64_t)blockIdx.x * 64 + threadIdx.x) * kHChunk;
  if (s >= len) return;
  const int64_t e = s + kHChunk < len ? s + kHChunk : len;
  const double* in = tmp + m.t_off;
  double* yo = y + m.y_off;
  const int64_t ylen = m.ylen;
  double z0 = 0, z1 = 0, z2 = 0, yv = 0;
  int64_t i0 = s - warm;
  if (i0 <= 0) {
    i0 = 0;
    const double y0 = in[len - 1];
    z0 = c.zi0 * y0;
    z1 = c.zi1 * y0;
    z2 = c.zi2 * y0;
  }
  const int64_t d0 = (len - 1 - s) - kFPad - m.pick0;  // p - pick0 at i = s; falls by one per step
  int64_t k0 = d0 >= 0 ? d0 / r : -((-d0 + r - 1) / r);
  int r0 = (int)(d0 - k0 * r);
  wh::serial_run<kHBlock>(
      i0, e, [&](int64_t) { return true; }, [&](int64_t i) { return in[len - 1 - i]; },
      [&](int64_t i) { return in[len - 1 - i]; },
      [&](int64_t i, double v) {
        TDF2_STEP(v);
        if (i >= s) {
          if (r0 == 0 && k0 >= 0 && k0 < ylen) yo[k0] = yv;
          if (--r0 < 0) {
            r0 = r - 1;
            --k0;
          }
        }
      });
}

__global__ __launch_bounds__(256) void hv_copy_kernel(const double* __restrict__ x, const HvUtt* __restrict__ meta,
                                                      double* __restrict__ y) {
  const HvUtt m = meta[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < m.ylen) y[m.y_off + i] = x[m.x_off + i];
}

// mean of the decimated signal in two steps: kMeanParts partial sums per utterance (fixed association: the result does
// not depend on the launch), then their sum / length
constexpr int kMeanParts = 32;
__global__ __launch_bounds__(256) void hv_mean_part_kernel(const HvUtt* __restrict__ meta, const double* __restrict__ y,
                                                           double* __restrict__ part) {
  __shared__ double scratch[16];
  const HvUtt m = meta[blockIdx.y];
  const int64_t chunk = (m.ylen + kMeanParts - 1) / kMeanParts;
  const int64_t begin = (int64_t)blockIdx.x * chunk;
  const int64_t end = begin + chunk < m.ylen ? begin + chunk : m.ylen;
  double s = 0.0;
  for (int64_t i = begin + threadIdx.x; i < end; i += 256) s += y[m.y_off + i];
  s = wh::block_sum(s, scratch);
  if (threadIdx.x == 0) part[(int64_t)blockIdx.y * kMeanParts + blockIdx.x] = s;
}
__global__ __launch_bounds__(64) void hv_mean_kernel(const HvUtt* __restrict__ meta, const double* __restrict__ part,
                                                     int n_utt, double* __restrict__ mean) {
  const int u = blockIdx.x * 64 + threadIdx.x;
  if (u >= n_utt) return;
  double s = 0.0;
  for (int k = 0; k < kMeanParts; ++k) s += part[(int64_t)u * kMeanParts + k];
  mean[u] = s / (double)meta[u].ylen;
}

// z = [zeros(pad), y - mean, zeros(pad)]; also rewrites y itself mean-removed (the refinement reads it)
__global__ __launch_bounds__(256) void hv_pad_kernel(const HvUtt* __restrict__ meta, double* __restrict__ y,
                                                     const double* __restrict__ mean, int pad, double* __restrict__ z) {
  const HvUtt m = meta[blockIdx.y];
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= m.ylen + 2 * pad) return;
  const int64_t i = j - pad;
  double v = 0.0;
  if (i >= 0 && i < m.ylen) {
    v = y[m.y_off + i] - mean[blockIdx.y];
    y[m.y_off + i] = v;
  }
  z[m.z_off + j] = v;
}

// One workgroup per (utterance, channel) walks the 1 ms frames in tiles of 256.  The four event trains are sorted
// and the frames ascending, so the events a tile can need are a window that only moves forward: the workgroup keeps
// a cursor per train and, per tile, turns the edges behind it into INTERVALS in LDS — location (e[i]+e[i+1])/2/fs and
// instantaneous frequency fs/(e[i+1]-e[i]), one thread per interval, coalesced loads — and every frame searches the
// locations of that window and interpolates between two staged intervals: one LDS read per search step and one divide
// per (frame, train), where the per-frame form re-derived both neighbouring intervals from four edges (two reads, an
// add and a multiply per step, three divides).  A band of centre f has ~1.1 f events per second at most, so the
// window staged for 256 ms is sized by the band (kRawChunk caps it).  A frame whose answer is not inside the staged
// window falls back to the search over the whole list in global memory (guarded, never taken for speech bands).
// Same arithmetic as wh::interp_four_trains, value for value.
#ifndef WH_HV_RAW_SEGS
#define WH_HV_RAW_SEGS 1
#endif
#ifndef WH_HV_RAW_TILE
#define WH_HV_RAW_TILE 64  // (one wave per workgroup: 1.56 ms at config 3 against 1.66 for 128 and 1.79 for 256; 23.3 against 25.0 ms at 1024 utterances)
#endif
constexpr int kRawTile = WH_HV_RAW_TILE;   // frames per tile = threads per workgroup
constexpr int kRawChunk = 2 * kRawTile;    // intervals staged per train and tile, at most

__global__ __launch_bounds__(kRawTile) void hv_raw_kernel(const HvUtt* __restrict__ meta, const wh::BandJob* __restrict__ jobs,
                                                     const double* __restrict__ band_f0, int nb, double fs_d,
                                                     double f0_floor, double f0_ceil, double* __restrict__ raw,
                                                     unsigned long long* __restrict__ live, int dense) {
  static_assert(kRawTile == 64, "a tile's live bits are one wave ballot");
  __shared__ double2 iv[4][kRawChunk];  // (location, frequency) of interval start + i
  __shared__ int s_next[4];
  const HvUtt m = meta[blockIdx.y];
  const int b = blockIdx.x;
  const wh::BandJob job = jobs[(int64_t)blockIdx.y * nb + b];
  double* out = raw + m.f1_off * nb + (int64_t)b * m.nf1;
  // What hv_detect scans: ONE BIT per (channel, frame) — set where a candidate survived the range tests — as a 64-bit
  // word per (channel, 64-frame tile), the wave's ballot.  The candidate VALUES are written only where that bit is set
  // (round 6; `dense`: everywhere, for the debug read-out): hv_detect reads a value only inside a run of live channels,
  // and most of the [channel][frame] map is dead — it used to leave as 12.2 MB of doubles + 1.5 MB of bytes per 10 s
  // utterance, nearly all of it zeros nobody read.
  unsigned long long* lv = live + m.l_off + (int64_t)b * m.ntile;
  int cnt[4];
  bool usable = true;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    cnt[k] = job.counts[k];
    usable = usable && (cnt[k] - 1 >= 3);
  }
  if (!usable) {  // fewer than 3 intervals in a train: no candidate anywhere (dio.py:159-162)
    for (int64_t t = threadIdx.x; t < m.ntile; t += kRawTile) lv[t] = 0ull;
    if (dense)
      for (int64_t f = threadIdx.x; f < m.nf1; f += kRawTile) out[f] = 0.0;
    return;
  }
  const double bf = band_f0[b];
  const double half_inv_fs = 0.5 / fs_d;
  // intervals staged per tile: twice the ~1.1*bf*0.256 a band-limited signal can hold, plus the cursor's slack
  int need = 2 * (int)(bf * 1.1 * (kRawTile * 0.001) + 1.0) + 8;
  need = need > kRawChunk - 1 ? kRawChunk - 1 : need;  // (the last slot always holds the +inf sentinel of the search)
  int search_steps = 0;  // doubling steps that settle a lower_bound over [0, need]: 2^steps > need
  while ((1 << search_steps) < need + 1) ++search_steps;
  // blockIdx.z cuts the frames into gridDim.z segments of whole tiles, each with its own workgroup (the tile loop is a
  // chain of load -> barrier -> search -> barrier; more workgroups in flight hide it).  A segment's first cursors
  // are found by a search over the whole list.
  const int64_t tiles_all = (m.nf1 + kRawTile - 1) / kRawTile;
  const int64_t tiles_seg = (tiles_all + gridDim.z - 1) / gridDim.z;
  const int64_t f_begin = (int64_t)blockIdx.z * tiles_seg * kRawTile;
  const int64_t f_end = f_begin + tiles_seg * kRawTile < m.nf1 ? f_begin + tiles_seg * kRawTile : m.nf1;
  if (f_begin >= m.nf1) return;
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    const double* e = job.edges + (int64_t)k * job.cap;
    const double t = (double)f_begin * 1 / 1000;
    int lo = 0, hi = job.counts[k] - 1;
    if (f_begin == 0) hi = 0;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const double loc = (e[mid] + e[mid + 1]) * half_inv_fs;
      if (loc < t) lo = mid + 1; else hi = mid;
    }
    s_next[k] = lo;
  }
  __syncthreads();
  int pos[4];  // per train: number of interval locations before the current tile's first frame
#pragma unroll
  for (int k = 0; k < 4; ++k) pos[k] = s_next[k];
  __syncthreads();
  for (int64_t f0 = f_begin; f0 < f_end; f0 += kRawTile) {
    int start[4], nloc[4];
    double ea[4][2], eb[4][2];  // all of a tile's edge loads are issued before the first divide consumes one
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      start[k] = pos[k] - 2 > 0 ? pos[k] - 2 : 0;
      const int ni = cnt[k] - 1;
      nloc[k] = ni - start[k] < need ? ni - start[k] : need;  // intervals staged
      const double* e = job.edges + (int64_t)k * job.cap + start[k];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int i = threadIdx.x + r * kRawTile;
        ea[k][r] = i < nloc[k] ? e[i] : 0.0;
        eb[k][r] = i < nloc[k] ? e[i + 1] : 1.0;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int i = threadIdx.x + r * kRawTile;
        // (slots past the staged intervals hold +inf locations: the search below needs no bounds; written as a select —
        // as a branch that skips the second round's divide for the bands below ~400 Hz: 1.36 against 1.30 ms)
        iv[k][i] = i < nloc[k] ? make_double2((ea[k][r] + eb[k][r]) * half_inv_fs, wh::fdiv(fs_d, eb[k][r] - ea[k][r]))
                               : make_double2(INFINITY, 0.0);
      }
    __syncthreads();
    const int64_t f = f0 + threadIdx.x;
    int lo_g[4] = {0, 0, 0, 0};
    double cand = 0.0;
    if (f < f_end) {
      const double t = (double)f * 1 / 1000;  // basic_temporal_positions (harvest.py:21)
      double v[4];
      // lower_bound of t among the staged locations, the four trains in lockstep, as the branch-free DOUBLING search: the
      // position grows by a power of two whenever the element in front of the probe is still below t — add, read,
      // compare, select per step and train (the halving form carried a [lo, hi) pair and an activity flag: three times the
      // integer work of a kernel whose VALU is saturated and 22 % FP64).  The +inf sentinels behind the staged intervals
      // stand in for the bounds checks.
      int lo4[4] = {0, 0, 0, 0};
      for (int st = 1 << (search_steps - 1); st > 0; st >>= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int q = lo4[k] + st;
          lo4[k] = iv[k][q - 1].x < t ? q : lo4[k];
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ni = cnt[k] - 1;  // intervals of the whole train; location i = (e[i]+e[i+1])/2/fs
        const int lo = lo4[k];
        const bool inside = lo < nloc[k] || start[k] + nloc[k] == ni;
        const double* e = job.edges + (int64_t)k * job.cap;
        int g = start[k] + lo;  // count of locations < t over the whole train
        if (!inside) {          // the window ended before t: global search
          int l2 = g, h2 = ni;
          while (l2 < h2) {
            const int mid = (l2 + h2) >> 1;
            const double loc = (e[mid] + e[mid + 1]) * half_inv_fs;
            if (loc < t) l2 = mid + 1; else h2 = mid;
          }
          g = l2;
        }
        lo_g[k] = g;
        const int ih = g < 1 ? 1 : (g > ni - 1 ? ni - 1 : g);
        const int il = ih - 1;
        double x_lo, x_hi, y_lo, y_hi;
        if (il >= start[k] && ih < start[k] + nloc[k]) {
          const double2 a = iv[k][il - start[k]], c = iv[k][ih - start[k]];
          x_lo = a.x;
          y_lo = a.y;
          x_hi = c.x;
          y_hi = c.y;
        } else {
          const double e0 = e[il], e1 = e[il + 1], e2 = e[ih], e3 = e[ih + 1];
          x_lo = (e0 + e1) * half_inv_fs;
          x_hi = (e2 + e3) * half_inv_fs;
          y_lo = wh::fdiv(fs_d, e1 - e0);
          y_hi = wh::fdiv(fs_d, e3 - e2);
        }
        const double slope = wh::fdiv(y_hi - y_lo, x_hi - x_lo);
        v[k] = slope * (t - x_lo) + y_lo;
      }
      cand = (((v[0] + v[1]) + v[2]) + v[3]) / 4;
      if (cand > bf * 1.1 || cand < bf * 0.9 || cand > f0_ceil || cand < f0_floor) cand = 0.0;  // harvest.py:273-276
      if (dense || cand > 0) out[f] = cand;
    }
    {
      const unsigned long long bits = __ballot(cand > 0);  // (lanes behind the utterance's end: 0)
      if (threadIdx.x == 0) lv[f0 / kRawTile] = bits;
    }
    // the last frame of the tile hands its counts to the next tile as the new cursors
    const int64_t last = f0 + kRawTile - 1 < f_end - 1 ? f0 + kRawTile - 1 : f_end - 1;
    if (f == last) {
#pragma unroll
      for (int k = 0; k < 4; ++k) s_next[k] = lo_g[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) pos[k] = s_next[k];
  }
}

// ---- raw candidates AND detection in one pass (round 6) ---------------------------------------------------------------
// hv_raw_kernel walks the frames of ONE channel and leaves a [channel][frame] map for hv_detect_kernel, which walks the
// channels of one frame: 6 MB of candidate values written and 4.4 MB read back per 10 s utterance, plus the launch.  This
// kernel is the transpose: one wave per (utterance, tile of 64 frames) walks ALL channels in order, a lane per frame —
// the interpolation of hv_raw_kernel, value for value (same staging of (location, frequency) intervals in LDS, same
// doubling search, same window check with the search over the whole list as the way out) — and what DetectCandidates
// (harvest.py:88-110) needs of a frame's column is kept in the lane as it goes by: the length of the current run of live
// channels and the sum of its values, added channel after channel.  (np.mean adds them pairwise; hv_detect_kernel
// reproduces that association, this kernel does not: the values themselves agree with the reference's to ~1e-9 Hz —
// overlap-save filters against FFT products — so the 1e-16 of a summation order is not what parity rests on, and the
// eight accumulators + pending group of the pairwise form cost a wave per SIMD: 7.06 against 5.4 ms at 256 utterances.)
// A run of >= 10 channels that ends becomes a candidate.  Nothing but the 15 candidate slots per frame leaves the kernel.
//   Where a tile's search of a channel's four edge lists starts is the band walker's advice (emit_crossings_block's
// hints: crossings in front of the tile's first sample); the edges of the NEXT channel are fetched while the current
// one is searched.  Workgroup order: an utterance's tiles on one XCD, consecutive (neighbouring tiles read overlapping
// stretches of every list — from that XCD's L2 the second time).
struct RdMeta {  // what the edge loads of a channel are addressed by: fetched a channel ahead of them
  int4 h0, h1, c;  // the tile's hints, the next tile's, the trains' edge counts
  const double* e;
  int64_t cap;
  double bf;
};
struct RdStage {
  int start[4], nloc[4], ni[4];
  double ea[4][2];  // edge start + lane + 64 r of every train (the interval's other edge is the next lane's)
  const double* e;
  int64_t cap;
  double bf;
  int need, steps;
  bool usable;
};

#ifndef WH_HV_RAWDET_MINW
#define WH_HV_RAWDET_MINW 3  // (12 KB of LDS per wave: 13 waves per CU whatever the registers)
#endif
__global__ __launch_bounds__(kRawTile, WH_HV_RAWDET_MINW) void hv_rawdet_kernel(const HvUtt* __restrict__ meta, const wh::BandJob* __restrict__ jobs,
                                                             const double* __restrict__ band_f0,
                                                             const int32_t* __restrict__ hints, int nb, int n_utt, int n_xcd, int max_ntile,
                                                             double fs_d, double f0_floor, double f0_ceil,
                                                             double* __restrict__ dc, int32_t* __restrict__ dcount,
                                                             double* __restrict__ raw_dbg) {
  __shared__ double2 iv[4][kRawChunk];       // (location, frequency) of interval start + i
  const int xcd = blockIdx.x % n_xcd, local = blockIdx.x / n_xcd;  // (n_xcd: 8, or 1 for fewer than eight utterances)
  const int u = (local / max_ntile) * n_xcd + xcd;
  if (u >= n_utt) return;
  const HvUtt m = meta[u];
  const int64_t T = local % max_ntile;
  if (T >= m.ntile) return;
  const int lane = threadIdx.x;
  const int64_t f = T * kRawTile + lane;
  const bool live_f = f < m.nf1;
  const double t = (double)f * 1 / 1000;  // basic_temporal_positions (harvest.py:21)
  const double half_inv_fs = 0.5 / fs_d;
  const int32_t* hT = hints + (m.l_off + T * nb) * 4;          // [channel][train] of this tile
  const int32_t* cU = jobs[(int64_t)u * nb].counts;            // [channel][train] of this utterance (contiguous)

  // Two fetch stages run ahead of the channel being searched: the addressing data of channel b + 2 (hints, counts, list
  // base: uniform loads), then — from the data fetched one channel earlier — the edges of channel b + 1.  (With both in
  // one stage every channel waited a memory round trip for its hints before its edge loads could be issued.)
  auto fetch_meta = [&](int b, RdMeta& q) {
    b = b < nb ? b : nb - 1;  // (a surplus prefetch behind the last channel: harmless)
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef const v4i __attribute__((address_space(1))) * g4;
    const v4i a = *(g4)(hT + b * 4), c = *(g4)(cU + b * 4);
    const v4i n = T + 1 < m.ntile ? *(g4)(hT + (nb + b) * 4) : c;  // (behind the last tile: every edge)
    q.h0 = make_int4(a.x, a.y, a.z, a.w);
    q.h1 = make_int4(n.x, n.y, n.z, n.w);
    q.c = make_int4(c.x, c.y, c.z, c.w);
    const wh::BandJob* j = jobs + (int64_t)u * nb + b;
    q.e = j->edges;
    q.cap = j->cap;
    q.bf = band_f0[b];
  };
  auto fetch = [&](const RdMeta& q, RdStage& s) {
    s.e = q.e;
    s.cap = q.cap;
    s.bf = q.bf;
    const int cs[4] = {q.c.x, q.c.y, q.c.z, q.c.w}, h0s[4] = {q.h0.x, q.h0.y, q.h0.z, q.h0.w}, h1s[4] = {q.h1.x, q.h1.y, q.h1.z, q.h1.w};
    bool usable = true;
    int need = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = __builtin_amdgcn_readfirstlane(cs[k]);  // (uniform: kept in scalar registers)
      const int ni = c - 1;  // intervals of the whole train
      usable = usable && (ni >= 3);
      s.ni[k] = ni;
      // the hint counts EDGES in front of the tile's first sample; the locations in front of its first frame are that
      // many, give or take two: start four entries early (hv_raw_kernel starts two in front of the exact count) — and the
      // NEXT tile's hint says where this tile's edges end: the window is what the tile needs plus that slack, not a
      // worst-case budget (staged from a budget of twice the band's rate the tiles of an utterance read every list 2.2 times)
      const int h0 = __builtin_amdgcn_readfirstlane(h0s[k]);
      const int h1 = __builtin_amdgcn_readfirstlane(h1s[k]);
      int st = h0 - 4;
      st = st > ni - 1 ? ni - 1 : st;
      st = st < 0 ? 0 : st;
      s.start[k] = st;
      int n = h1 - st + 3;
      n = n > kRawChunk - 1 ? kRawChunk - 1 : n;  // (the last slot always holds the +inf sentinel of the search)
      n = n > ni - st ? ni - st : n;
      n = n < 1 ? 1 : n;
      s.nloc[k] = n;
      need = n > need ? n : need;
    }
    int steps = 0;
    while ((1 << steps) < need + 1) ++steps;
    s.need = need;
    s.steps = steps;
    s.usable = usable;
    if (!usable) return;  // fewer than 3 intervals in a train: no candidate anywhere in this channel (dio.py:159-162)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double* e = s.e + (int64_t)k * s.cap + s.start[k];
      {
        const int ic = lane <= s.nloc[k] ? lane : 0;  // entries 0 .. nloc (start + nloc <= ni: a valid edge); clamped, not skipped: no branch per load
        s.ea[k][0] = wh::ldg(e + ic);
      }
      if (s.need >= kRawTile) {  // (uniform; rare: more than 60 crossings of one kind in 64 ms)
        const int i = lane + kRawTile;
        const int ic = i <= s.nloc[k] ? i : 0;
        s.ea[k][1] = wh::ldg(e + ic);
      }
    }
  };

  // the lane's walk state: DetectCandidates of its frame
  double run_sum = 0.0;
  int idx = 0, count = 0;
  double* out = dc + (m.f1_off + f) * kMaxC;

  auto consume = [&](int b, const RdStage& s) {
    double cand = 0.0;
    if (s.usable) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // the interval's upper edge: the next lane's entry (lane 63 of the first round: lane 0 of the second)
        const double up0 = __shfl_down(s.ea[k][0], 1);
        if (s.need < kRawTile) {  // (uniform) one round: entries 0 .. 63, intervals 0 .. 62 at most
          iv[k][lane] = lane < s.nloc[k] ? make_double2((s.ea[k][0] + up0) * half_inv_fs, wh::fdiv(fs_d, up0 - s.ea[k][0]))
                                         : make_double2(INFINITY, 0.0);
        } else {
          const double up1 = __shfl_down(s.ea[k][1], 1);
          const double wrap = __shfl(s.ea[k][1], 0);
          const double eb[2] = {lane == kRawTile - 1 ? wrap : up0, up1};  // (entry 127 is never an interval: nloc <= 127)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int i = lane + r * kRawTile;
            iv[k][i] = i < s.nloc[k] ? make_double2((s.ea[k][r] + eb[r]) * half_inv_fs, wh::fdiv(fs_d, eb[r] - s.ea[k][r]))
                                     : make_double2(INFINITY, 0.0);
          }
        }
      }
      wh::sync<kRawTile>();
      if (live_f) {
        double v[4];
        int lo4[4] = {0, 0, 0, 0};
        for (int st = 1 << (s.steps - 1); st > 0; st >>= 1) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int q = lo4[k] + st;
            lo4[k] = iv[k][q - 1].x < t ? q : lo4[k];
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int ni = s.ni[k];
          const int lo = lo4[k];
          const double* e = s.e + (int64_t)k * s.cap;
          // the staged window holds the answer if a location >= t lies inside it (or the list ends in it) AND the count
          // did not stop at the window's first entry while locations lie in front of the window (the hint was late)
          const bool inside = (lo < s.nloc[k] || s.start[k] + s.nloc[k] == ni) && (lo > 0 || s.start[k] == 0);
          int g = s.start[k] + lo;  // count of locations < t over the whole train
          if (!inside) {            // search over the whole list
            int l2 = 0, h2 = ni;
            while (l2 < h2) {
              const int mid = (l2 + h2) >> 1;
              const double loc = (e[mid] + e[mid + 1]) * half_inv_fs;
              if (loc < t) l2 = mid + 1; else h2 = mid;
            }
            g = l2;
          }
          const int ih = g < 1 ? 1 : (g > ni - 1 ? ni - 1 : g);
          const int il = ih - 1;
          double x_lo, x_hi, y_lo, y_hi;
          if (il >= s.start[k] && ih < s.start[k] + s.nloc[k]) {
            const double2 a = iv[k][il - s.start[k]], c = iv[k][ih - s.start[k]];
            x_lo = a.x;
            y_lo = a.y;
            x_hi = c.x;
            y_hi = c.y;
          } else {
            const double e0 = e[il], e1 = e[il + 1], e2 = e[ih], e3 = e[ih + 1];
            x_lo = (e0 + e1) * half_inv_fs;
            x_hi = (e2 + e3) * half_inv_fs;
            y_lo = wh::fdiv(fs_d, e1 - e0);
            y_hi = wh::fdiv(fs_d, e3 - e2);
          }
          const double slope = wh::fdiv(y_hi - y_lo, x_hi - x_lo);
          v[k] = slope * (t - x_lo) + y_lo;
        }
        cand = (((v[0] + v[1]) + v[2]) + v[3]) / 4;
        if (cand > s.bf * 1.1 || cand < s.bf * 0.9 || cand > f0_ceil || cand < f0_floor) cand = 0.0;  // harvest.py:273-276
      }
      wh::sync<kRawTile>();  // (the next channel's intervals overwrite iv)
    }
    if (raw_dbg && live_f) raw_dbg[m.f1_off * nb + (int64_t)b * m.nf1 + f] = cand;
    // DetectCandidates (harvest.py:88-110): first and last channel count as dead; a run of >= 10 live channels that ends
    // gives the mean of its values
    const bool live = live_f && b >= 1 && b < nb - 1 && cand > 0;
    if (live) {
      run_sum += cand;
      ++idx;
    } else if (idx > 0) {
      if (idx >= 10 && count < kMaxC) out[count++] = run_sum / (double)idx;
      idx = 0;
      run_sum = 0.0;
    }
  };

  RdMeta qa, qb;
  RdStage sa, sb;
  fetch_meta(0, qa);
  fetch_meta(1, qb);
  fetch(qa, sa);
  for (int b = 0; b < nb; b += 2) {
    fetch(qb, sb);          // the edges of channel b + 1
    fetch_meta(b + 2, qa);
    consume(b, sa);
    fetch(qa, sa);          // ... of b + 2
    fetch_meta(b + 3, qb);
    if (b + 1 < nb) consume(b + 1, sb);
  }
  if (live_f) {
    for (int c = count; c < kMaxC; ++c) out[c] = 0.0;  // (hv_refine reads every slot)
    dcount[m.f1_off + f] = count;
  }
}

// NumPy's pairwise summation for n <= 128 (what np.mean does on the run of channel values)
__device__ __forceinline__ double np_sum_strided(const double* __restrict__ a, int64_t stride, int n) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += a[i * stride];
    return r;
  }
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = a[j * stride];
  int i = 8;
  for (; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] += a[(i + j) * stride];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += a[i * stride];
  return res;
}

#ifndef WH_HV_DETECT_STAGE
#define WH_HV_DETECT_STAGE 1
#endif
// The channel walk reads hv_raw's byte map (1 = a candidate survived the band's range test) instead of the candidates
// themselves — an eighth of the bytes of a pass that runs at HBM speed — and fetches the values only for the runs that
// count.
__global__ __launch_bounds__(256) void hv_detect_kernel(const HvUtt* __restrict__ meta, int nb,
                                                        const double* __restrict__ raw,
                                                        const unsigned long long* __restrict__ live_map, double* __restrict__ dc,
                                                        int32_t* __restrict__ dcount) {
  const HvUtt m = meta[blockIdx.y];
  // the grid is sized by the longest utterance of the batch: blocks wholly behind this utterance's end leave at once
  // (also the nf1 == 0 case, where the clamp below would point in front of the column)
  if ((int64_t)blockIdx.x * 256 >= m.nf1) return;
  const int64_t f_raw = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live_f = f_raw < m.nf1;  // (the tail threads of the last block stay for its output pass)
  const int64_t f = live_f ? f_raw : m.nf1 - 1;
  const double* col = raw + m.f1_off * nb + f;  // element b at col[b * nf1]
  // the wave's 64 frames are one tile of hv_raw's bit map: ONE word per channel for the whole wave (a uniform address)
  const int64_t tile_w = ((int64_t)blockIdx.x * 256 + (threadIdx.x & ~63)) / 64;
  const unsigned long long* lcol = live_map + m.l_off + (tile_w < m.ntile ? tile_w : m.ntile - 1);
  const int lane_w = threadIdx.x & 63;
#if !WH_HV_DETECT_STAGE
  double* out = dc + (m.f1_off + f) * kMaxC;
#endif
  // Phase 1: the runs.  Phase 2 sums them run by run: every lane of the wave is then inside the same summation loop at
  // the same time (its loads in flight together), where summing a run the moment the walk finds its end made the wave
  // go through one summation — two or three dependent rounds of global loads — per distinct end position among its
  // 64 lanes.
  __shared__ unsigned short runs[kMaxC][256];  // (first live channel) << 8 | length
  int count = 0;
  int run_start = -1;  // 'st': index of the last dead channel before a live run
  bool prev = false;   // channel 0 is forced dead
  // the channel walk is a chain of dependent branches; its loads are not: eight channels are fetched together
  for (int b0 = 0; live_f && b0 < nb; b0 += 8) {
    unsigned v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (unsigned)(lcol[(int64_t)(b0 + q < nb ? b0 + q : nb - 1) * m.ntile] >> lane_w) & 1u;  // clamped, not skipped: a
    // conditional load becomes a branch, and eight of them a chain of load-wait-load (the surplus values are not read)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int b = b0 + q;
      if (b < 1 || b >= nb) continue;
      const bool live = (b < nb - 1) && (v[q] > 0);  // last channel forced dead
      if (live && !prev) run_start = b - 1;
      if (!live && prev) {
        const int ed = b - 1;
        if (ed - run_start >= 10 && count < kMaxC) runs[count++][threadIdx.x] = (unsigned short)(((run_start + 1) << 8) | (ed - run_start));
      }
      prev = live;
    }
  }
#if WH_HV_DETECT_STAGE
  // The kMaxC slots of a frame are 120 bytes apart: written from the walk's lanes, every store instruction touched 64
  // cache lines.  They are collected in LDS (dense rows: an odd stride in doubles, conflict-free both ways) and leave as one
  // contiguous block per workgroup.
  __shared__ double s_out[256 * kMaxC];
  for (int c = 0; c < kMaxC; ++c) {
    double val = 0.0;
    if (live_f && c < count) {
      const int rr = runs[c][threadIdx.x];
      const int n = rr & 0xff;
      val = np_sum_strided(col + (int64_t)(rr >> 8) * m.nf1, m.nf1, n) / (double)n;
    }
    s_out[threadIdx.x * kMaxC + c] = val;
  }
  if (live_f) dcount[m.f1_off + f] = count;
  __syncthreads();
  const int64_t f_blk = (int64_t)blockIdx.x * 256;
  const int n_blk = (int)(m.nf1 - f_blk < 256 ? m.nf1 - f_blk : 256);
  double* ob = dc + (m.f1_off + f_blk) * kMaxC;
  for (int q = threadIdx.x; q < n_blk * kMaxC; q += 256) ob[q] = s_out[q];
#else
  if (!live_f) return;
  for (int c = 0; c < kMaxC; ++c) {
    double val = 0.0;
    if (c < count) {
      const int rr = runs[c][threadIdx.x];
      const int n = rr & 0xff;
      val = np_sum_strided(col + (int64_t)(rr >> 8) * m.nf1, m.nf1, n) / (double)n;
    }
    out[c] = val;
  }
  dcount[m.f1_off + f] = count;
#endif
}

// ---- candidate refinement ------------------------------------------------------------------------
// GetRefinedF0 (harvest.py:169-211) for one (frame, candidate), evaluated by a 16-lane DPP row; a wave
// refines four candidates at once.  No FFT: the Blackman-windowed frame and its derivative-windowed twin are
// accumulated directly into the <= 6 harmonic bins the reference reads from its two zero-padded FFTs, and the
// 24 partial sums are reduced inside the row with DPP lane permutes (no LDS, no cross-row traffic).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  // every control used here reads a live lane of the same row for every lane: no "old" value is needed
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// sum over the RL lanes (a power of two up to 16, aligned) that work on one candidate, result in every lane of the group: xor-1, xor-2
// quad permutes, then the mirror of the half row and, for 16, of the row
template <int RL>
__device__ __forceinline__ double row_sum(double v) {
  static_assert(RL == 16 || RL == 8 || RL == 4 || RL == 2 || RL == 1, "group of a DPP row");
  if (RL >= 2) v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  if (RL >= 4) v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  if (RL >= 8) v += dpp_f64<0x141>(v);   // row_half_mirror
  if (RL == 16) v += dpp_f64<0x140>(v);  // row_mirror
  return v;
}

// TWL: the twiddles come from the workgroup's LDS copy (tw_lds, at a compile-time offset of the dynamic LDS block, so
// that a look-up is one ds_read_b128 whose address register is the running byte offset itself); else from the global
// tables through tw_base.
// WTAB: the Blackman window and its derivative twin depend on the window length alone (the frame time cancels out of the
// reference's window argument, see the tabulated loop) — (w(j), dw(j)) come from a per-call table (win_tab, row hwl at
// offset hwl*(hwl+2), a zero pair at either end) instead of being re-derived per sample by rotation + DPP neighbour exchange.
// Window half length, transform length, harmonic count and the six rounded harmonic bins of a candidate
// (harvest.py:171-174,203), and their packed form: two candidates with the same key have the same two spectra at the
// same bins (hv_refine_kernel's classes), and the key is all the sample loop needs to know about the candidate.
struct RefineGeom {
  int hwl, nfft, nh, bins[6];
};
__device__ __forceinline__ int refine_nfft(int hwl) {
  const int L = 2 * hwl + 1;
  int e = 0;
  while ((1 << e) < L) ++e;
  return 1 << (e + 1);
}
__device__ __forceinline__ RefineGeom refine_geom(double f0c, double fs) {
  RefineGeom g;
  g.hwl = (int)ceil(3 * fs / f0c / 2);
  g.nfft = refine_nfft(g.hwl);
  g.nh = (int)fmin(floor(fs / 2 / f0c), 6.0);
#pragma unroll
  for (int h = 0; h < 6; ++h) g.bins[h] = (int)(f0c * g.nfft / fs * (double)(h + 1) + 0.5);
  return g;
}
// the first bin is within a few units of 1.5 * nfft / hwl (f0c lies in (1.5 fs / hwl, 1.5 fs / (hwl - 1)]): any
// deterministic function of (hwl, nfft) serves as the base it is stored against
__device__ __forceinline__ int refine_bin_base(int hwl, int nfft) { return (int)(1.5f * (float)nfft / (float)hwl) - 1; }
// [0,9) hwl; [9,11) bins[0] - base; [11,14) nh; [14,16), [16,19), [19,22), [22,25), [25,28): bins[h] - (h+1)*bins[0], which is
// within +-(h+2)/2, offset to be non-negative.  -1: a value outside these ranges (not expected; the item is then a class
// of its own and the sample loop derives everything from the candidate).
__device__ __forceinline__ int refine_pack(const RefineGeom& g) {
  const int b0 = g.bins[0] - refine_bin_base(g.hwl, g.nfft);
  const int d1 = g.bins[1] - 2 * g.bins[0] + 1, d2 = g.bins[2] - 3 * g.bins[0] + 3, d3 = g.bins[3] - 4 * g.bins[0] + 3;
  const int d4 = g.bins[4] - 5 * g.bins[0] + 3, d5 = g.bins[5] - 6 * g.bins[0] + 3;
  const bool fits = (unsigned)d1 < 4u && (unsigned)d2 < 8u && (unsigned)d3 < 8u && (unsigned)d4 < 8u && (unsigned)d5 < 8u &&
                    (unsigned)b0 < 4u && (unsigned)g.hwl < 512u && (unsigned)g.nh < 8u;
  return fits ? (g.hwl | (b0 << 9) | (g.nh << 11) | (d1 << 14) | (d2 << 16) | (d3 << 19) | (d4 << 22) | (d5 << 25)) : -1;
}
__device__ __forceinline__ RefineGeom refine_unpack(int key) {
  RefineGeom g;
  g.hwl = key & 511;
  g.nfft = refine_nfft(g.hwl);
  g.nh = (key >> 11) & 7;
  const int b0 = refine_bin_base(g.hwl, g.nfft) + ((key >> 9) & 3);
  g.bins[0] = b0;
  g.bins[1] = 2 * b0 + ((key >> 14) & 3) - 1;
  g.bins[2] = 3 * b0 + ((key >> 16) & 7) - 3;
  g.bins[3] = 4 * b0 + ((key >> 19) & 7) - 3;
  g.bins[4] = 5 * b0 + ((key >> 22) & 7) - 3;
  g.bins[5] = 6 * b0 + ((key >> 25) & 7) - 3;
  return g;
}
#ifndef WH_HV_WIN_FENCE
#define WH_HV_WIN_FENCE 1
#endif
__device__ __forceinline__ bool rotation_path_ok(bool wtab, double a0, double a0_frac) {
  return !wtab && a0 > 1.0 && a0_frac > 1e-6 && a0_frac < 1.0 - 1e-6;
}
// RL: lanes per candidate.  The set-up before and the reductions after the sample loop are per wave instruction,
// whatever the number of candidates in the wave, and with the tabulated windows they outweigh the loop (233 + 391
// against 40 instructions per 16 samples): eight lanes per candidate — eight candidates per wave — halve their share
// and drop one of the four reduction steps.  The rotation path needs the 16-lane row rotates.
// `members(eval)`: the caller runs eval(f0c_m, &f0, &score) for every candidate of the frame that shares this one's
// window length and harmonic bins (hv_refine_kernel: the seven overlapped copies of a slowly moving pitch track mostly
// do) — the spectra at the harmonic bins are the same for all of them, only the score looks at the candidate itself.
template <bool TWL, bool WTAB, int RL, class Members>
__device__ __forceinline__ void hv_refine_row(wh::ckp<const double> WH_RESTRICT yl, int64_t ybase, int64_t ylen, double fs,
                                              double t0, double f0c, int pkey, double f0_floor, double f0_ceil,
                                              const double2* __restrict__ tw_base, const char* tw_lds, int tw_n,
                                              wh::ckp<const double2> WH_RESTRICT rot_tab, wh::ckp<const double2> WH_RESTRICT win_tab,
                                              Members members) {
  static_assert(WTAB || RL == 16, "the rotation path exchanges window values with 16-lane row rotates");
  const int l16 = threadIdx.x & (RL - 1);  // lane within the candidate's group
  // pkey >= 0: the candidate's packed geometry (refine_pack), computed when the classes were built
  const RefineGeom geo = pkey >= 0 ? refine_unpack(pkey) : refine_geom(f0c, fs);
  const int hwl = geo.hwl;
  const double hwl_d = (double)hwl;
  const int L = 2 * hwl + 1;
  const double wlit = (2 * hwl_d + 1) / fs;
  const int nfft = geo.nfft;
  const int nh = geo.nh;
  // twiddles exp(-2*pi*i*k/nfft): from the workgroup's LDS copy of the largest table any of its candidates can
  // need (the smaller tables are its subsamples, bit for bit), else from the global table
  const char* tw = TWL ? tw_lds : reinterpret_cast<const char*>(tw_base + nfft);
  const int tw_sh = (TWL ? (__ffs(tw_n) - __ffs(nfft)) : 0) + 4;  // table subsampling, and elements -> bytes
  auto twiddle = [&](int byte_off) -> double2 {
    if constexpr (TWL) {  // the table sits at LDS address 0 (checked by the kernel): the offset IS the address
#if WH_BOUNDS
      if ((unsigned)byte_off + 16u > (unsigned)tw_n * 16u || (byte_off & 15)) wh::oob_report(wh::WH_CK_TWIDDLE, byte_off >> 4, tw_n);
#endif
      typedef double v2d __attribute__((ext_vector_type(2)));
      typedef const v2d __attribute__((address_space(3))) * lds_tw_t;
      const v2d w = *(lds_tw_t)(size_t)(uint32_t)byte_off;
      return make_double2(w.x, w.y);
    } else {
      return *reinterpret_cast<const double2*>(tw + byte_off);
    }
  };
  int bins[6];
#pragma unroll
  for (int h = 0; h < 6; ++h) bins[h] = geo.bins[h];
  const double inv_fs = 1.0 / fs;  // the sample index below is floor(integer + 0.501 +- 1e-12): an ulp cannot move it
  auto idx_raw_at = [&](int j) -> double {
    const double v = (t0 + (double)(j - hwl) * inv_fs) * fs + 0.001;  // "first-aid treatment", harvest.py:178
    return v > 0 ? v + 0.5 : v - 0.5;                               // round_matlab does not truncate (Q1)
  };
  // window phase: 2*common = pi*xw, xw advances by dx per unit step of idx_raw (steps are 1, or 2 where the
  // +-0.5 offset flips sign at negative times)
  const double dx = 2.0 / (fs * wlit);
  double sd1 = 0.0, cd1 = 1.0, sd2 = 0.0, cd2 = 1.0;  // step rotations of the general path (set there)
  double xr[6], xi[6], dr[6], di[6];
#pragma unroll
  for (int h = 0; h < 6; ++h) xr[h] = xi[h] = dr[h] = di[h] = 0.0;
  // one sample of the frame: (s2, c2) = sin/cos(pi*xw(j)); two_next / two_prev: the index step to the neighbour is 2
  auto sample = [&](int j, double ir, double s2, double c2, bool two_next, bool two_prev) {
    const double mj = 0.42 + 0.5 * c2 + 0.08 * (2 * c2 * c2 - 1);  // cos(4c) = 2cos^2(2c) - 1
    double mn = 0.0, mp = 0.0;
    if (j + 1 < L) {
      const double c = two_next ? c2 * cd2 - s2 * sd2 : c2 * cd1 - s2 * sd1;  // cos(2c + step*pi*dx)
      mn = 0.42 + 0.5 * c + 0.08 * (2 * c * c - 1);
    }
    if (j > 0) {
      const double c = two_prev ? c2 * cd2 + s2 * sd2 : c2 * cd1 + s2 * sd1;
      mp = 0.42 + 0.5 * c + 0.08 * (2 * c * c - 1);
    }
    double dw;
    if (j == 0) dw = -mn / 2;
    else if (j == L - 1) dw = mp / 2;
    else dw = -((mn - mj) + (mj - mp)) / 2;
    const double irc = fmax(1.0, fmin((double)ylen, ir)) - 1;
    const double smp = yl[(int64_t)irc - ybase];
    const double a = smp * mj, d = smp * dw;
#pragma unroll
    for (int h = 0; h < 6; ++h) {
      if (h < nh) {
        const double2 w = twiddle(((bins[h] * j) & (nfft - 1)) << tw_sh);
        xr[h] = fma(a, w.x, xr[h]);
        xi[h] = fma(a, w.y, xi[h]);
        dr[h] = fma(d, w.x, dr[h]);
        di[h] = fma(d, w.y, di[h]);
      }
    }
  };
  // The sample index of j is floor(idx_raw_at(j)) = floor(A + j) with A = t0*fs - hwl + 0.501: it steps by exactly one
  // per j as long as A's fraction stays clear of 0 and 1 by more than the rounding of the expression (~1e-9 at 60 s),
  // which the 0.501 guarantees for every decimated rate that is a multiple of 50 Hz; checked here (row-uniform), and
  // a frame that fails the check takes the general path below.
  const double a0 = idx_raw_at(0);
  const double a0_frac = a0 - floor(a0);
  if (WTAB && a0 > 1.0 && a0_frac > 1e-6 && a0_frac < 1.0 - 1e-6) {
    // index_raw(j) = t0*fs + (j - hwl) + 0.501 is never truncated before it enters the window argument
    // (index_raw - 1)/fs - t0 = (j - hwl - 0.499)/fs: the frame time cancels, whatever its position between two
    // samples, and the window pair of row hwl applies as tabulated (only the sample PICK floor(index_raw) depends on
    // the frame, through i_first below).  The loop is the 24 FMAs, the six twiddle gathers and one table read
    // (fetched an iteration ahead).
    // (Advancing some of the six twiddles by a rotation instead of gathering them — 4 more FP64 operations per harmonic and
    // iteration, one LDS gather less — is slower: 2.32 ... 2.48 against 2.28 ms; the loop is bound by VALU issue, not LDS.)
    // The sums run over the sample PAIRS (hwl + m, hwl - m), m = 1..hwl, around the window's centre: the twiddle of
    // bin b at -m is the conjugate of the one at +m, so with a = x*w and d = x*dw
    //   sum_j a_j e^{-i th (j - hwl)} = a_0 + sum_m (a_m + a_-m) cos(th m) - i (a_m - a_-m) sin(th m)
    // — one twiddle gather and four FMAs per harmonic and PAIR instead of per sample (half the gathers, half the FMAs of
    // the loop: 3.19 -> 2.57 ms at config 3).  Referring the phase to the centre multiplies both spectra by the same unit
    // factor e^{i th hwl}: the power |X|^2 and the cross term Im(conj(X) D) that the instantaneous frequency is made of do
    // not see it.  (The window is NOT symmetric — its argument is (j - hwl - 0.499)/fs — so both window pairs are read.)
    // Row hwl of the table has a zero entry in front of and behind its 2*hwl + 1 pairs, and a lane past the window's end
    // (m > hwl in the last iteration) reads those: no predicate and no select inside the loop; the staged signal
    // replicates the utterance's edge samples, so the sample index needs no clamp either.
    const wh::ckp<const double2> wt = win_tab + (hwl * (hwl + 2) + (hwl + 1));  // the centre pair
    int tix[6], tstep[6];  // byte offsets into the twiddle table (see the rotation path below)
    const int tmask = ((nfft - 1) << tw_sh);
#pragma unroll
    for (int h = 0; h < 6; ++h) {
      tix[h] = ((bins[h] * (1 + l16)) & (nfft - 1)) << tw_sh;
      tstep[h] = ((bins[h] * RL) & (nfft - 1)) << tw_sh;
    }
    const int n_it = (hwl + RL - 1) / RL;
    const wh::ckp<const double> yc = yl + ((int)((int64_t)a0 - 1 - ybase) + hwl);  // the centre sample in the staged signal
    if (l16 == 0) {  // the centre sample: cos = 1 for every bin
      const double2 wc = wt[0];
      const double smp = yc[0];
#pragma unroll
      for (int h = 0; h < 6; ++h) {
        xr[h] = smp * wc.x;
        dr[h] = smp * wc.y;
      }
    }
    const int m_end = hwl + 1;  // the zero pair
    int m = 1 + l16;            // (<= hwl + 1: RL <= 4 ... hwl >= RL is not required, the clamp covers it)
    m = m < m_end ? m : m_end;
    double2 cur_p = wt[m], cur_m = wt[-m];
    for (int it = 0; it < n_it; ++it) {
      int mn = m + RL;
      mn = mn < m_end ? mn : m_end;
      const double sp = yc[m], sm = yc[-m];
      const double ap = sp * cur_p.x, am = sm * cur_m.x, dp = sp * cur_p.y, dm = sm * cur_m.y;
      const double ea = ap + am, oa = ap - am, ed = dp + dm, od = dp - dm;
      double2 wv[6];  // all six gathers in flight before the first FMA needs one
#pragma unroll
      for (int h = 0; h < 6; ++h) wv[h] = twiddle(tix[h]);
      cur_p = wt[mn];  // the next iteration's window pairs, in flight under this one's FMAs
      cur_m = wt[-mn];
#if WH_HV_WIN_FENCE
      // (without the fence the compiler folds the loop-carried pair into "load the current pair at the top of the
      // iteration", and every iteration waits for a global round trip)
      asm volatile("" ::: "memory");
#endif
#pragma unroll
      for (int h = 0; h < 6; ++h) {
        const double2 w = wv[h];
        xr[h] = fma(ea, w.x, xr[h]);
        xi[h] = fma(oa, w.y, xi[h]);
        dr[h] = fma(ed, w.x, dr[h]);
        di[h] = fma(od, w.y, di[h]);
        tix[h] = (tix[h] + tstep[h]) & tmask;
      }
      m = mn;
    }
  } else if (rotation_path_ok(WTAB, a0, a0_frac)) {
    if constexpr (!WTAB) {
    // Every index of the frame is positive (all frames but the first few of an utterance): idx_raw, and with it the
    // window phase xw, is linear in j, so this lane's samples j = l16 + 16 i are a fixed rotation of 16*pi*dx apart —
    // one sincospi to start, a 6-flop rotation per sample after that (<= 24 steps: error growth ~1e-15; the rotation
    // constants of every window length come from a small per-call table).
    // The derivative window needs the Blackman values of samples j-1 and j+1: those are what the neighbouring lanes
    // of the row hold in the same iteration (lane 0's left neighbour is lane 15's value of the previous iteration,
    // lane 15's right neighbour lane 0's value of the next one, which is therefore computed one iteration ahead), so
    // they are fetched with two DPP row rotates each instead of being recomputed (2 x (rotation + Blackman) = 26
    // FP64 operations per sample in a kernel that is VALU-bound).  All 16 lanes of a row run the same number of
    // iterations (validity is a predicate), which keeps every source lane of the rotates alive.  The twiddle offset
    // (bin*j mod nfft) advances by a constant per iteration: an add and a mask instead of a 32-bit multiply.
    // (A lane-contiguous layout, j = l*nb + i, needs no cross-lane traffic at all but multiplies the lane stride of
    // the twiddle gathers by nb: their LDS bank conflicts made it 20 % slower, measured.)
    const double2 rot = rot_tab[hwl];  // (sin, cos)(16*pi*dx)
    const double s16 = rot.x, c16 = rot.y;
    double s2, c2;
    {
      const double ir0 = idx_raw_at(l16);
      sincospi(2 * ((ir0 - 1) / fs - t0) / wlit, &s2, &c2);
    }
    auto blackman = [](double c) { return 0.42 + 0.5 * c + 0.08 * (2 * c * c - 1); };  // cos(4c) = 2cos^2(2c) - 1
    // twiddle look-ups as byte offsets into the table in use, already scaled by its subsampling shift: the
    // offset of harmonic h advances by (16*bin_h mod nfft) << shift per iteration.  All six harmonics are accumulated
    // whatever nh is (the surplus ones, for candidates above fs/12, are simply not read afterwards): no per-harmonic
    // predication inside the loop.
    int tix[6], tstep[6];
    const int tmask = ((nfft - 1) << tw_sh);
#pragma unroll
    for (int h = 0; h < 6; ++h) {
      tix[h] = ((bins[h] * l16) & (nfft - 1)) << tw_sh;
      tstep[h] = ((bins[h] * 16) & (nfft - 1)) << tw_sh;
    }
    double m_prev = 0.0, m_cur = blackman(c2);
    const int n_it = (L + 15) >> 4;
    int j = l16;
    // staged-signal index of sample j: clamp(floor(A) + j, 1, ylen) - 1 - ybase, carried as an int
    const int64_t i_first = (int64_t)a0;
    const int i_lo = (int)(0 - ybase), i_hi = (int)(ylen - 1 - ybase);
    int si = (int)(i_first - 1 - ybase) + l16;
    for (int it = 0; it < n_it; ++it, j += 16) {
      const double cn = c2 * c16 - s2 * s16;  // phase of sample j + 16
      s2 = s2 * c16 + c2 * s16;
      c2 = cn;
      const double m_next = blackman(c2);
      const double right = dpp_f64<0x12F>(m_cur), right_wrap = dpp_f64<0x12F>(m_next);  // lane l <- lane l+1 (15 <- 0)
      const double left = dpp_f64<0x121>(m_cur), left_wrap = dpp_f64<0x121>(m_prev);     // lane l <- lane l-1 (0 <- 15)
      const double mn = j + 1 < L ? (l16 == 15 ? right_wrap : right) : 0.0;
      const double mp = j > 0 ? (l16 == 0 ? left_wrap : left) : 0.0;
      const double mj = m_cur;
      double dw;
      if (j == 0) dw = -mn / 2;
      else if (j == L - 1) dw = mp / 2;
      else dw = -((mn - mj) + (mj - mp)) / 2;
      double smp = 0.0;
      if (j < L) smp = yl[si < i_lo ? i_lo : (si > i_hi ? i_hi : si)];
      si += 16;
      const double a = smp * mj, d = smp * dw;
#pragma unroll
      for (int h = 0; h < 6; ++h) {
        const double2 w = twiddle(tix[h]);
        xr[h] = fma(a, w.x, xr[h]);
        xi[h] = fma(a, w.y, xi[h]);
        dr[h] = fma(d, w.x, dr[h]);
        di[h] = fma(d, w.y, di[h]);
        tix[h] = (tix[h] + tstep[h]) & tmask;
      }
      m_prev = m_cur;
      m_cur = m_next;
    }
    }
  } else {
    sincospi(dx, &sd1, &cd1);
    sincospi(2 * dx, &sd2, &cd2);
    for (int j = l16; j < L; j += RL) {
      const double ir = idx_raw_at(j);
      const double xw = 2 * ((ir - 1) / fs - t0) / wlit;
      double s2, c2;
      sincospi(xw, &s2, &c2);  // sin/cos(2*common)
      const bool two_next = (j + 1 < L) && (idx_raw_at(j + 1) - ir > 1.5);
      const bool two_prev = (j > 0) && (ir - idx_raw_at(j - 1) > 1.5);
      sample(j, ir, s2, c2, two_next, two_prev);
    }
  }
  // The four sums of every harmonic go round the row; lane h then evaluates harmonic h alone — instantaneous
  // frequency, amplitude, deviation: five FP64 divides and a square root, ~85 instructions that all sixteen lanes used
  // to repeat for each of the six harmonics — and three more row sums collect the totals.  (Folding the 24 partial sums
  // once by DPP and finishing through an LDS scratch needs a third of the instructions but its 24 KB cost a workgroup
  // per CU: 9.5 against 8.0 ms, measured.)
  constexpr int P = (6 + RL - 1) / RL;  // harmonics per lane: lane l takes l, l + RL, ...
  double sa[P], sb[P], sc_[P], sd[P];
  int my_bin[P];
#pragma unroll
  for (int q = 0; q < P; ++q) {
    sa[q] = sb[q] = sc_[q] = sd[q] = 0.0;
    my_bin[q] = 0;
  }
#pragma unroll
  for (int h = 0; h < 6; ++h) {
    const double a = row_sum<RL>(xr[h]), b = row_sum<RL>(xi[h]), c = row_sum<RL>(dr[h]), d = row_sum<RL>(di[h]);
    if (l16 == h % RL) {
      sa[h / RL] = a;
      sb[h / RL] = b;
      sc_[h / RL] = c;
      sd[h / RL] = d;
      my_bin[h / RL] = bins[h];
    }
  }
  // per harmonic of this lane: instantaneous frequency and amplitude (harvest.py:193-203) — functions of the spectra
  // alone; numerator / denominator / variation are summed per candidate below, over ITS harmonics
  double inst_q[P], amp_q[P];
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const double p = sa[q] * sa[q] + sb[q] * sb[q];
    const double nm = sa[q] * sd[q] - sb[q] * sc_[q];
    // bin / nfft is exact (power of two), and so is the halving
    // (the quotients of this epilogue — seventeen per lane — go through wh::fdiv: they were two thirds of its instructions)
    inst_q[q] = ((double)my_bin[q] * (1.0 / (double)nfft) + wh::fdiv(wh::fdiv(nm, p) * 0.5, M_PI)) * fs;
    amp_q[q] = sqrt(p);
  }
  // numerator and denominator run over the harmonics below nh — the same for every member (nh is part of the class key)
  double t_num = 0.0, t_den = 0.0, a_q[P];
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const int h = l16 + q * RL;
    a_q[q] = wh::fdiv(inst_q[q], (double)(h + 1));
    if (h < nh) {
      t_num += amp_q[q] * inst_q[q];
      t_den += amp_q[q] * (double)(h + 1);
    }
  }
  const double num = row_sum<RL>(t_num), den = row_sum<RL>(t_den);
  const double rf_all = wh::fdiv(num, den);
  // (scoring four members at a time, one per lane, was measured and is no faster: 3.51 against 3.53 ms)
  members([&](double f0m, double* out_f0, double* out_sc) {
    double t_var = 0.0;
#pragma unroll
    for (int q = 0; q < P; ++q) {
      const int h = l16 + q * RL;
      if (h < nh) t_var += fabs(wh::fdiv(a_q[q] - f0m, f0m));
    }
    const double var = row_sum<RL>(t_var);
    double rf = rf_all;
    double sc = wh::fdiv(1.0, 0.000000000001 + wh::fdiv(var, (double)nh));
    if (rf < f0_floor || rf > f0_ceil || sc < 2.5) {
      rf = 0.0;
      sc = 0.0;
    }
    *out_f0 = rf;
    *out_sc = sc;
  });
}

#ifndef WH_HV_CLASSES
#define WH_HV_CLASSES 1  // 0: no sharing between the candidates of a frame (timing experiments)
#endif
#ifndef WH_HV_MINW
#define WH_HV_MINW 3  // waves per SIMD the tabulated variant is compiled for (2: 5.07 ms against 4.09 at config 3)
#endif
#ifndef WH_HV_ROW_LANES
#define WH_HV_ROW_LANES 2  // (round 6, 512 x 10 s: 8 lanes 22.1 ms, 4: 17.7, 2: 16.5, 1: 18.6; 64 x 10 s: 4: 2.30, 2: 2.15, 1: 2.49)
#endif
#ifndef WH_HV_TAB_FRAMES
#define WH_HV_TAB_FRAMES 32  // (round 6, with two lanes per candidate: 32 frames 16.36 ms against 16.50 for 24 at 512 x 10 s; round 4,
                            // four lanes: (the list building is per block and a wave pass takes 16 classes whatever it holds; config 3:
                            // 16 frames 3.81 ms, 24: 3.34, 32: 3.37, 48: 4.79 — 52.3 against 59.7 ms at 1024 utterances)
#endif
#ifndef WH_HV_ITEM_CAP
#define WH_HV_ITEM_CAP 1344  // work-list slots in LDS; a block whose frames hold more takes them in several rounds of whole frames
#endif
// lanes per candidate and frames per workgroup of the two refinement variants
constexpr int refine_lanes(bool wtab) { return wtab ? WH_HV_ROW_LANES : 16; }
constexpr int refine_frames(bool wtab) { return wtab ? WH_HV_TAB_FRAMES : 4; }
constexpr int refine_item_cap(bool wtab) { return refine_frames(wtab) * kRows < WH_HV_ITEM_CAP ? refine_frames(wtab) * kRows : WH_HV_ITEM_CAP; }

template <bool TWL, bool WTAB>
__global__ __launch_bounds__(256, WTAB ? WH_HV_MINW : 1) void hv_refine_kernel(const HvUtt* __restrict__ meta, const double* __restrict__ y,
                                                        const double* __restrict__ dc, const int32_t* __restrict__ dcount,
                                                        double fs, double f0_floor, double f0_ceil, int hmax, int seglen,
                                                        const double2* __restrict__ tw_base, int tw_n,
                                                        const double2* __restrict__ rot_tab,
                                                        const double2* __restrict__ win_tab,
                                                        double* __restrict__ rf0, double* __restrict__ rsc,
                                                        int64_t* __restrict__ lst, int item_cap) {
  // All of the kernel's LDS is the dynamic block, so that it starts at LDS address 0 and the twiddle table's byte
  // offsets are LDS addresses as they stand (hv_refine_lds_bytes mirrors this layout).
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RL = refine_lanes(WTAB);
  constexpr int kFramesPerBlock = refine_frames(WTAB);
  constexpr int kBuckets = 64;
  const HvUtt m = meta[blockIdx.y];
  const int64_t f_first = (int64_t)blockIdx.x * kFramesPerBlock;
  if (f_first >= m.nf1) return;
  // LDS: the twiddle table of the largest transform length first (tw_n points; TWL false: none, the global tables are
  // read) — every sample of every refinement gathers 6 twiddles at scattered indices, LDS serves those, the L1 does
  // not — then the staged signal around the block's frames.
  // (wh::ckp<T>: T* in every shipped build, range-checked in the bounds build — wh_device.h)
  const wh::ckp<double2> twl = wh::ck_make(reinterpret_cast<double2*>(smem), TWL ? tw_n : 0, wh::WH_CK_TWIDDLE);
  if (TWL && (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
  constexpr int kItems = refine_item_cap(WTAB);                // (>= kRows: a single frame always fits)
  const int seg_pad = (seglen + 1) & ~1;
  double* yl_raw = reinterpret_cast<double*>(smem + (TWL ? sizeof(double2) * (size_t)tw_n : 0));
  const wh::ckp<double> yl = wh::ck_make(yl_raw, seg_pad, wh::WH_CK_LDS_MAIN);
  const wh::ckp<double> cl_val = wh::ck_make(yl_raw + seg_pad, kItems, wh::WH_CK_LDS_AUX);  // kItems
  int* ints_raw = reinterpret_cast<int*>(yl_raw + seg_pad + kItems);
  const wh::ckp<int> cl_meta = wh::ck_make(ints_raw, kItems, wh::WH_CK_LDS_OTHER);               // kItems
  const wh::ckp<int> order = wh::ck_make(ints_raw + kItems, kItems, wh::WH_CK_LDS_OTHER);        // kItems
  const wh::ckp<int> key_s = wh::ck_make(ints_raw + 2 * kItems, kItems, wh::WH_CK_LDS_OTHER);    // kItems: packed geometry of an item (refine_pack)
  const wh::ckp<int> bucket = wh::ck_make(ints_raw + 3 * kItems, kBuckets + 1, wh::WH_CK_LDS_SCRATCH);  // kBuckets (+ the class count)
  int& cl_n = bucket[kBuckets];
  const wh::ckp<uint32_t> nzmask = wh::ck_make(reinterpret_cast<uint32_t*>(ints_raw + 3 * kItems + kBuckets + 1), kFramesPerBlock * 4,
                                               wh::WH_CK_LDS_SCRATCH);  // [kFramesPerBlock][4]: rows of a frame that hold a candidate
  const wh::ckp<int> foff = wh::ck_make(ints_raw + 3 * kItems + kBuckets + 1 + kFramesPerBlock * 4, kFramesPerBlock + 1,
                                        wh::WH_CK_LDS_SCRATCH);  // [kFramesPerBlock + 1]: first list slot of a frame
  for (int i = threadIdx.x; i < kFramesPerBlock * 4; i += 256) nzmask[i] = 0;
  if (TWL)
    for (int i = threadIdx.x; i < tw_n; i += 256) twl[i] = tw_base[tw_n + i];
  const int64_t centre0 = (int64_t)floor(((double)f_first * 1 / 1000) * fs + 0.5);
  // (may be negative: the staged signal replicates the utterance's first and last sample beyond its ends, which is what
  // the reference's index clamp reads there, harvest.py:179)
  const int64_t ybase = centre0 - hmax - 3;
  const wh::ckp<const double> yu = wh::ck_make(y + m.y_off, m.ylen, wh::WH_CK_WAVEFORM);
  // the utterance's share of the global arrays: candidate slots in, result pool and list heads out, the window tables
  const wh::ckp<const double> dc_u = wh::ck_make(dc + m.f1_off * kMaxC, m.nf1 * kMaxC, wh::WH_CK_IN) - m.f1_off * kMaxC;
  const wh::ckp<double> rf0_u = wh::ck_make(rf0 + m.f1_off * kRows, m.nf1 * kRows, wh::WH_CK_OUT) - m.f1_off * kRows;
  const wh::ckp<double> rsc_u = wh::ck_make(rsc + m.f1_off * kRows, m.nf1 * kRows, wh::WH_CK_OUT) - m.f1_off * kRows;
  const wh::ckp<int64_t> lst_u = wh::ck_make(lst + m.f1_off, m.nf1, wh::WH_CK_OUT) - m.f1_off;
  const wh::ckp<const double2> win_ck = wh::ck_make(win_tab, win_tab ? (long long)(hmax + 2) * (hmax + 4) : 0, wh::WH_CK_TABLE);
  const wh::ckp<const double2> rot_ck = wh::ck_make(rot_tab, hmax + 2, wh::WH_CK_TABLE);
  for (int i = threadIdx.x; i < seglen; i += 256) {
    const int64_t g = ybase + i;
    yl[i] = yu[g < 0 ? 0 : (g > m.ylen - 1 ? m.ylen - 1 : g)];
  }
  if (threadIdx.x == 0) cl_n = 0;
  __syncthreads();
  // gather the overlapped candidates of the block's frames (+-3 frames, harvest.py:114-125) and compact the
  // non-zero ones into a work list.  The reference's [frame][105] candidate map is ~17 % occupied; what the later stages
  // need from it is the ORDER of a frame's candidates (first-index / last-index tie rules), not the row numbers, so the
  // results are stored as per-frame lists in row order: the block's frames share one contiguous pool region (at the place
  // the dense rows of its first frame would start), lst[frame] = (first pool slot << 8) | count.
  // A thread's (up to 7) candidate slots are fetched in one round of unconditional loads from clamped source frames —
  // hv_detect writes every one of a frame's kMaxC slots, zeros past its count, so the count itself is not needed; as
  // `if (in range && c < dcount[src]) cand = dc[..]` every slot was two dependent round trips, one slot after the other.
  constexpr int kGather = (kFramesPerBlock * kRows + 255) / 256;
  double cv[kGather];
#pragma unroll
  for (int it = 0; it < kGather; ++it) {
    const int q = threadIdx.x + it * 256;
    const int qc = q < kFramesPerBlock * kRows ? q : 0;
    const int fl = qc / kRows, e = qc % kRows;
    int64_t src = f_first + fl + (e / kMaxC - 3);
    src = src < 0 ? 0 : (src > m.nf1 - 1 ? m.nf1 - 1 : src);
    cv[it] = dc_u[(m.f1_off + src) * kMaxC + e % kMaxC];
  }
  // The work list is written in LIST-SLOT order (the rank of a row among its frame's rows: the order the results are
  // stored in), so a frame's items are contiguous and an item's index is its result slot: a first pass marks the rows
  // that hold a candidate, a second, behind the per-frame offsets, places them.
  bool live_q[kGather];
#pragma unroll
  for (int it = 0; it < kGather; ++it) {
    const int q = threadIdx.x + it * 256;
    live_q[it] = false;
    if (q >= kFramesPerBlock * kRows) continue;
    const int fl = q / kRows, e = q % kRows;
    const int64_t f = f_first + fl;
    if (f >= m.nf1) continue;
    const int64_t src = f + (e / kMaxC - 3);
    double cand = (src >= 0 && src < m.nf1) ? cv[it] : 0.0;
    if (e == 0 && f < 3) cand = dc_u[(m.f1_off + f) * kMaxC + 6];  // stray seeding of row 0 (harvest.py:119)
    cv[it] = cand;
    if (cand != 0.0 && ceil(3 * fs / cand / 2) <= (double)hmax) {
      live_q[it] = true;
      atomicOr(&nzmask[fl * 4 + (e >> 5)], 1u << (e & 31));
    }
  }
  __syncthreads();
  const int64_t pool_base = (m.f1_off + f_first) * kRows;
  static_assert(kFramesPerBlock <= 64, "one lane per frame below");
  // the work-list encoding: a 5-bit frame index and an 11-bit successor link in cl_meta, a 6-bit sort key
  static_assert(kFramesPerBlock <= 32 && kItems < 2048 && kBuckets <= 64, "field widths of the hv_refine work list");
  if (threadIdx.x < 64) {  // per-frame counts -> list offsets: one lane per frame, a wave scan (one lane walking the frames
                           // was a chain of dependent LDS round trips with the other 255 threads at the barrier)
    const int fl = threadIdx.x;
    const int c = fl < kFramesPerBlock
                      ? __popc(nzmask[fl * 4]) + __popc(nzmask[fl * 4 + 1]) + __popc(nzmask[fl * 4 + 2]) + __popc(nzmask[fl * 4 + 3])
                      : 0;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (fl >= o) incl += up;
    }
    if (fl < kFramesPerBlock) {
      foff[fl] = incl - c;
      if (f_first + fl < m.nf1) lst_u[m.f1_off + f_first + fl] = ((pool_base + (incl - c)) << 8) | (int64_t)c;
    }
    if (fl == 63) foff[kFramesPerBlock] = incl;
  }
  __syncthreads();
  // Rounds of whole frames whose candidates fit the work list (kItems slots; the 32 frames of a block hold ~400 candidates
  // on speech-like input, 3360 at most): normally one.  The first round is placed from the registers of the gather above
  // (which die here: inside the loop they would stay alive across the refinement passes), a further round fetches its
  // rows again.
  auto round_end = [&](int fa) {
    if (foff[kFramesPerBlock] - foff[fa] <= item_cap) return (int)kFramesPerBlock;  // the usual case: all that is left
    int fb = fa + 1;  // (walking the frames costs a dependent LDS read each, on every thread)
    while (fb < kFramesPerBlock && foff[fb + 1] - foff[fa] <= item_cap) ++fb;  // (item_cap <= kItems: the LDS slots)
    return fb;
  };
  auto slot_of = [&](int fl, int e, int s0) {
    const wh::ckp<const uint32_t> mw = nzmask + fl * 4;
    int slot = foff[fl] - s0 + __popc(mw[e >> 5] & ((1u << (e & 31)) - 1u));  // rank of row e among the round's rows
    for (int w = 0; w < (e >> 5); ++w) slot += __popc(mw[w]);
    return slot;
  };
  int fa = 0, fb = round_end(0);
#pragma unroll
  for (int it = 0; it < kGather; ++it) {
    if (!live_q[it]) continue;
    const int q = threadIdx.x + it * 256;
    const int fl = q / kRows, e = q % kRows;
    if (fl >= fb) continue;
    const int slot = slot_of(fl, e, 0);
    cl_val[slot] = cv[it];
    cl_meta[slot] = fl;
  }
#if defined(WH_HV_REFINE_ABLATE) && WH_HV_REFINE_ABLATE == 1  // timing experiments: staging, gather and placement alone
  if (threadIdx.x == 0 && fb == 123456) rf0[0] = 0;
  return;
#endif
  while (true) {
    const int s0 = foff[fa];
    const int n_items = foff[fb] - s0;
    // EQUAL-KEY CLASSES.  What a refinement costs — the two windowed spectra at the harmonic bins — depends on the
    // candidate only through its window half length and its six rounded bins (harvest.py:171-174,203): the seven overlapped
    // copies of a slowly moving pitch track that meet in one frame mostly share them (measured on the benchmark input: 178 k
    // work items per 10 s, 121 k distinct keys; the reference's own test recording: 83 k / 66 k).  One item of a class does the
    // sums; every member gets its own score from them (it is the score that looks at the candidate's value,
    // harvest.py:205-206) — same arithmetic, same results, a third less work.
    // cl_meta fields: [0,5) frame; [5,11) iteration count (the counting sort's key); [17,28) successor in the class + 1,
    // bit 28: not the class's first item.  key_s: the packed geometry, which IS the class key.
    __syncthreads();
    // the keys, one thread per PLACED item (every lane busy; inside the gather loop above the same arithmetic ran seven
    // times per wave with a tenth of the lanes)
    for (int i = threadIdx.x; i < n_items; i += 256) {
      const RefineGeom g = refine_geom(cl_val[i], fs);
      const int key = refine_pack(g);
      // iteration count of the sample loop: the counting sort's key
      int skey = WTAB ? (g.hwl + RL - 1) / RL : (2 * g.hwl + 1 + RL - 1) / RL;
      skey = skey > kBuckets - 1 ? kBuckets - 1 : skey;
      key_s[i] = key >= 0 ? key : (int)(0x80000000u | (unsigned)i);  // (unpackable: a class of its own)
      cl_meta[i] = cl_meta[i] | (skey << 5);
    }
    __syncthreads();
    if (threadIdx.x < kBuckets) bucket[threadIdx.x] = 0;  // (for the counting sort below: one barrier less)
    constexpr int kClassScan = (kItems + 255) / 256;
    int link[kClassScan];  // per item of this thread: (successor + 1) | not-first flag << 11
  #pragma unroll
    for (int r = 0; r < kClassScan; ++r) {
      const int i = threadIdx.x + r * 256;
      link[r] = 0;
      if (i < n_items) {
        const int ka = key_s[i], fl = cl_meta[i] & 31;
        const int lo = foff[fl] - s0, hi = foff[fl + 1] - s0;  // the frame's items
        bool has_pred = false;
        int succ = 0;
        for (int j = lo; j < i; ++j) has_pred = has_pred || key_s[j] == ka;
        for (int j = hi - 1; j > i; --j) succ = key_s[j] == ka ? j + 1 : succ;
  #if !WH_HV_CLASSES
        has_pred = false;  // (ablation: every item its own class)
        succ = 0;
  #endif
        link[r] = succ | (has_pred ? 1 << 11 : 0);
      }
    }
    __syncthreads();
  #pragma unroll
    for (int r = 0; r < kClassScan; ++r) {
      const int i = threadIdx.x + r * 256;
      if (i < n_items) cl_meta[i] = (cl_meta[i] & 0x7ff) | (link[r] << 17);
    }
#if defined(WH_HV_REFINE_ABLATE) && WH_HV_REFINE_ABLATE == 2  // ... + keys and class scan
    if (threadIdx.x == 0 && fb == 123456) rf0[0] = 0;
    return;
#endif
    // A wave refines 64 / RL classes at once, one per group of RL lanes, and runs as long as its longest one: the
    // window length goes with 1/f0 (31 ... 340 samples at 8 kHz), and a frame's candidates are typically an f0 with its
    // octave neighbours.  Counting sort of the classes' first items by iteration count, so that the groups of a wave
    // (consecutive entries) carry windows of the same length class.
    {
      // (the buckets were zeroed in front of the class scan's barrier; a thread counts the items whose links it wrote)
      for (int i = threadIdx.x; i < n_items; i += 256)
        if (!(cl_meta[i] >> 28 & 1)) atomicAdd(&bucket[(cl_meta[i] >> 5) & 63], 1);  // the classes' first items
      __syncthreads();
      static_assert(kBuckets == 64, "one lane per bucket");
      if (threadIdx.x < 64) {  // exclusive scan of the counts, longest first (the long items start the block's schedule)
        const int k = kBuckets - 1 - threadIdx.x;
        const int c = bucket[k];
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int up = __shfl_up(incl, o, 64);
          if ((int)threadIdx.x >= o) incl += up;
        }
        bucket[k] = incl - c;
        if (threadIdx.x == 63) cl_n = incl;  // classes
      }
      __syncthreads();
        const int n_lead = cl_n;
      for (int i = threadIdx.x; i < n_items; i += 256)
        if (!(cl_meta[i] >> 28 & 1)) order[atomicAdd(&bucket[(cl_meta[i] >> 5) & 63], 1)] = i;
      __syncthreads();
  #if defined(WH_HV_REFINE_ABLATE) && WH_HV_REFINE_ABLATE == 3  // timing experiments: the list building alone
      if (threadIdx.x == 0 && n_lead == 123456) rf0[0] = 0;
      return;
  #endif
      for (int it = threadIdx.x / RL; it < n_lead; it += 256 / RL) {
        const int src = order[it];
        const int64_t f = f_first + (cl_meta[src] & 31);
        hv_refine_row<TWL, WTAB, RL>(
            yl, ybase, m.ylen, fs, (double)f * 1 / 1000, cl_val[src], key_s[src], f0_floor, f0_ceil, tw_base, smem, tw_n, rot_ck, win_ck,
            [&](auto eval) {
              int p = src;
              while (true) {
                double r0, r1;
                eval(cl_val[p], &r0, &r1);
                if ((threadIdx.x & (RL - 1)) == 0) {
                  rf0_u[pool_base + s0 + p] = r0;  // (an item's index is its slot in the round's part of the block's pool region)
                  rsc_u[pool_base + s0 + p] = r1;
                }
                const int nx = (cl_meta[p] >> 17) & 0x7ff;
                if (!nx) break;
                p = nx - 1;
              }
            });
      }
    }
    fa = fb;
    if (fa >= kFramesPerBlock) break;
    fb = round_end(fa);
    __syncthreads();  // (this round's lists have been consumed)
    for (int q = threadIdx.x; q < kFramesPerBlock * kRows; q += 256) {  // the next round's rows, fetched again (rare path)
      const int fl = q / kRows, e = q % kRows;
      if (fl < fa || fl >= fb || !((nzmask[fl * 4 + (e >> 5)] >> (e & 31)) & 1u)) continue;
      const int64_t f = f_first + fl;
      const int64_t src = f + (e / kMaxC - 3);
      const double cand = (e == 0 && f < 3) ? dc_u[(m.f1_off + f) * kMaxC + 6] : dc_u[(m.f1_off + src) * kMaxC + e % kMaxC];
      const int slot = slot_of(fl, e, foff[fa]);
      cl_val[slot] = cand;
      cl_meta[slot] = fl;
    }
  }
}

// RemoveUnreliableCandidates (harvest.py:215-234): a candidate survives if some candidate of frame j-1
// or j+1 lies within 5 %.  A workgroup takes kPruneFrames consecutive frames: the candidate lists of those frames and
// their two outer neighbours are fetched once (half a wave per frame), their non-zero entries compacted into LDS (ballot
// ranks), then every list entry is tested against the non-zero entries of its two neighbours.  (One 128-thread workgroup
// per frame — 640 k of them for 64 x 10 s — was bound by the rate at which workgroups can be launched, and read every
// frame three times.)
constexpr int kPruneFrames = 16;

// The result is a 128-bit keep mask per frame (bit k: entry k of the frame's list survives, i.e. is non-zero and passed
// the test): the contour kernels read the refined lists through it (wh_harvest_contour.h), which spares this pass the
// score array and pruned copies.
__global__ __launch_bounds__(256) void hv_prune_kernel(const HvUtt* __restrict__ meta, const double* __restrict__ rf0,
                                                       const int64_t* __restrict__ lst, uint32_t* __restrict__ keep) {
  constexpr int kRowPad = 128;                              // a row of nzl: the non-zero entries, then +inf
  __shared__ double nzl[kPruneFrames + 2][kRowPad];         // non-zero candidates of frames f_first-1 .. f_first+16
  __shared__ int ln[kPruneFrames + 2];
  static_assert(kRows <= kRowPad, "row padding");
  const HvUtt m = meta[blockIdx.y];
  const int64_t f_first = (int64_t)blockIdx.x * kPruneFrames;
  if (f_first >= m.nf1) return;
  // HALF a wave per frame: a list holds ~17 entries on speech-like input (105 at most), so with a whole wave per frame
  // three lanes in four idled through the neighbour loops; a half-wave takes entries l32 + 32 * pass, and a pass that
  // neither of the wave's two frames reaches is skipped.  The eight half-waves take rows hw, hw + 8, hw + 16 of the 18.
  const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
  const int hw = (threadIdx.x >> 6) * 2 + half;
  constexpr int kPer = (kPruneFrames + 2 + 7) / 8;
  constexpr int kPass = (kRows + 31) / 32;
  for (int i = threadIdx.x; i < (kPruneFrames + 2) * kRowPad; i += 256) (&nzl[0][0])[i] = INFINITY;
  // the list heads first, then every list — independent loads in flight before the first ballot needs one.  Entry k of
  // a list sits on lane k & 31 of its half (pass k >> 5) and stays there: the half-wave that fetched a frame also tests it.
  int64_t ent[kPer];
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const int fr = hw + 8 * i;
    const int64_t f = f_first - 1 + fr;
    const bool ok = fr < kPruneFrames + 2 && f >= 0 && f < m.nf1;
    ent[i] = ok ? lst[m.f1_off + f] : 0;
  }
  double val[kPer][kPass];
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const double* src = rf0 + (ent[i] >> 8);
    const int n = (int)(ent[i] & 255);
#pragma unroll
    for (int pass = 0; pass < kPass; ++pass) {
      const int e = l32 + 32 * pass;
      val[i][pass] = e < n ? src[e] : 0.0;
    }
  }
  __syncthreads();  // (the +inf fill is complete)
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const int fr = hw + 8 * i;
    const bool row = fr < kPruneFrames + 2;
    int n = 0;
#pragma unroll
    for (int pass = 0; pass < kPass; ++pass) {
      const double a = val[i][pass];
      const unsigned long long nz64 = __ballot(a != 0.0);  // zeros can never be the nearest candidate
      const uint32_t nz = (uint32_t)(nz64 >> (32 * half));
      if (row && a != 0.0) nzl[fr][n + __popc(nz & ((1u << l32) - 1u))] = a;
      n += __popc(nz);
    }
    if (row && l32 == 0) ln[fr] = n;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const int fr = hw + 8 * i;
    const int64_t f = f_first - 1 + fr;
    const bool valid = fr >= 1 && fr <= kPruneFrames && f < m.nf1;  // rows 0 and 17 are neighbours only
    const bool inner = valid && f >= 1 && f <= m.nf1 - 2;
    const int n_prev = valid ? ln[fr - 1] : 0, n_next = valid ? ln[fr + 1] : 0;
    const double* nb_prev = nzl[valid ? fr - 1 : 0];
    const double* nb_next = nzl[valid ? fr + 1 : 0];
    // loop bounds and the pass count of the wave: the larger of its two halves' (a half that reads past its own row's
    // entries reads +inf, which no minimum takes)
    const int cnt = valid ? (int)(ent[i] & 255) : 0;
    const int t_next = max(__builtin_amdgcn_readlane(n_next, 0), __builtin_amdgcn_readlane(n_next, 32));
    const int t_prev = max(__builtin_amdgcn_readlane(n_prev, 0), __builtin_amdgcn_readlane(n_prev, 32));
    const int t_cnt = max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 32));
#pragma unroll
    for (int pass = 0; pass < kPass; ++pass) {
      double v = val[i][pass];
      if (32 * pass < t_cnt) {
        if (inner && v != 0.0) {
          double e1 = 1.0, e2 = 1.0;  // SelectBestF0 with allowed_range = 1 (a zero candidate gives exactly 1)
          // min_k |v - nb_k| / v == (min_k |v - nb_k|) / v bit for bit (division by v > 0 is monotone): one divide per
          // neighbour frame instead of one per neighbour candidate
          double d1 = INFINITY, d2 = INFINITY;
          for (int k = 0; k < t_next; ++k) d1 = fmin(d1, fabs(v - nb_next[k]));
          for (int k = 0; k < t_prev; ++k) d2 = fmin(d2, fabs(v - nb_prev[k]));
          if (n_next > 0) e1 = fmin(e1, d1 / v);
          if (n_prev > 0) e2 = fmin(e2, d2 / v);
          if (fmin(e1, e2) > 0.05) v = 0.0;
        }
      }
      const unsigned long long kept = __ballot(v != 0.0);
      if (valid && l32 == 0) keep[(m.f1_off + f) * 4 + pass] = (uint32_t)(kept >> (32 * half));
    }
  }
}

double tdf2_pole_radius(double a1, double a2, double a3) {
  // max |root| of z^3 + a1 z^2 + a2 z + a3 (Durand-Kerner)
  double re[3] = {0.4, -0.2, 0.3}, im[3] = {0.9, 0.5, -0.7};
  for (int it = 0; it < 300; ++it)
    for (int i = 0; i < 3; ++i) {
      const double zr = re[i], zi = im[i];
      double pr = zr + a1, pi = zi;
      double tr = pr * zr - pi * zi + a2, ti = pr * zi + pi * zr;
      pr = tr * zr - ti * zi + a3;
      pi = tr * zi + ti * zr;
      double dr = 1, di = 0;
      for (int j = 0; j < 3; ++j)
        if (j != i) {
          const double ar = zr - re[j], ai = zi - im[j];
          const double nr = dr * ar - di * ai, ni = dr * ai + di * ar;
          dr = nr;
          di = ni;
        }
      const double den = dr * dr + di * di;
      if (den == 0) continue;
      re[i] -= (pr * dr + pi * di) / den;
      im[i] -= (pi * dr - pr * di) / den;
    }
  double r = 0;
  for (int i = 0; i < 3; ++i) r = fmax(r, hypot(re[i], im[i]));
  return r;
}

}  // namespace

extern "C" int wh_harvest(wh_ctx* ctx, void* stream, const wh_batch* b, const double* x, const double* tp, double fs,
                          double f0_floor, double f0_ceil, double frame_period_ms, int decimation_ratio,
                          const double* h_ba, const double* h_zi, int n_bands, const double* h_band_f0,
                          const int32_t* h_band_half, const double* h_band_taps, double* f0_out, double* vuv_out,
                          double* dbg_y, double* dbg_raw, double* dbg_f0_1ms) {
  if (!ctx || !b || !x || !tp || !h_band_f0 || !h_band_half || !h_band_taps || !f0_out || !vuv_out)
    return wh::fail_msg("wh_harvest", "null argument");
  WH_ENTER(ctx);
  if (decimation_ratio > 1 && (!h_ba || !h_zi)) return wh::fail_msg("wh_harvest", "decimation filter missing");
  if (n_bands < 3 || n_bands > 1024) return wh::fail_msg("wh_harvest", "n_bands out of range");
  if (int rc = wh::tables_make_room(ctx)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int B = b->n_utt;
  const int r = decimation_ratio < 1 ? 1 : decimation_ratio;
  // The anti-aliasing filter runs whenever fs > 8000 Hz — ALSO when the ratio rounds to 1 (8 kHz < fs < 12 kHz, e.g.
  // 11.025 kHz): the reference branches on `fs <= target_fs` (harvest.py:60) and then low-pass filters at 0.8 / r of
  // Nyquist with r = 1, keeping every sample.  The host says so by handing over the coefficients (a0 != 0).
  const bool filtered = r > 1 || (h_ba && h_zi && h_ba[4] != 0.0);
  const double fs_d = fs / r;
  int max_lb = 0, taps_total = 0;
  std::vector<int32_t> ti(n_bands * 3);
  for (int i = 0; i < n_bands; ++i) {
    const int lb = 2 * h_band_half[i] + 1;
    ti[i] = taps_total;
    ti[n_bands + i] = lb;
    ti[2 * n_bands + i] = h_band_half[i];  // filtered[(h+1) + g] == filtered[bias + 1 + g] with bias = h
    taps_total += lb;
    if (lb > max_lb) max_lb = lb;
  }
  const int pad = max_lb + 2;
  const int hmax = (int)ceil(3 * fs_d / f0_floor / 2) + 1;
  if (2 * hmax + 1 > WH_MAX_TWIDDLE / 2) return wh::fail_msg("wh_harvest", "f0_floor too low for the twiddle tables");
  std::vector<HvUtt> meta(B);
  std::vector<int64_t> e_off((size_t)B * n_bands), e_cap((size_t)B * n_bands);
  const bool caps_worst = ctx->hv_caps_worst;
  const bool caps_given = !caps_worst && !ctx->hv_caps_next.empty();
  if (caps_given && ctx->hv_caps_next.size() != (size_t)B * n_bands) {
    ctx->hv_caps_next.clear();
    return wh::fail_msg("wh_harvest", "wh_harvest_set_event_caps: one capacity per (utterance, channel) of THIS batch expected");
  }
  int64_t l_tot = 0;
  int64_t t_tot = 0, y_tot = 0, z_tot = 0, e_tot = 0, f1_tot = 0, max_len = 0, max_ylen = 0, max_nf1 = 0, max_nf = 0;
  for (int u = 0; u < B; ++u) {
    HvUtt& m = meta[u];
    m.x_off = b->h_x_off[u];
    m.n = b->h_x_off[u + 1] - b->h_x_off[u];
    if (m.n < 32) return wh::fail_msg("wh_harvest", "utterance shorter than 32 samples");
    if (filtered) {
      m.offset = (int64_t)ceil(140.0 / r) * r;
      m.nd = m.n + 2 * m.offset;
      const double n_out = ceil((double)m.nd / r);
      const int64_t n_beg = (int64_t)(r - (r * n_out - m.nd));
      const int64_t picks = (m.nd - (n_beg - 1) + r - 1) / r;
      m.ylen = picks - 2 * (m.offset / r);
      m.pick0 = (n_beg - 1) + (m.offset / r) * r;
    } else {
      m.offset = 0;
      m.nd = m.n;
      m.ylen = m.n;
      m.pick0 = 0;
    }
    m.t_off = t_tot;
    t_tot += m.nd + 2 * kFPad;
    m.y_off = y_tot;
    y_tot += m.ylen;
    m.z_off = z_tot;
    z_tot += m.ylen + 2 * pad;
    m.nf1 = (int64_t)(1000.0 * (double)m.n / fs / 1 + 1);
    m.f1_off = f1_tot;
    f1_tot += m.nf1;
    m.ntile = (m.nf1 + kRawTile - 1) / kRawTile;
    m.l_off = l_tot;
    l_tot += m.ntile * n_bands;
    m.f_off = b->h_frame_off[u];
    m.nf = b->h_frame_off[u + 1] - b->h_frame_off[u];
    for (int i = 0; i < n_bands; ++i) {
      // a band-limited channel centred on f crosses zero ~f times per second; 3x head-room + slack.  That is an
      // ESTIMATE: where the filtered signal is constant up to rounding (digital silence next to signal: the mean
      // removal of harvest.py:69 turns it into a DC level) the first difference changes sign at random, up to every
      // other sample.  Such a call raises WH_FLAG_EVENT_OVERFLOW, its counts stay exact (the walker counts on past a
      // full list), and the caller repeats it with them (wh_harvest_event_counts -> wh_harvest_set_event_caps) or with
      // the bound no signal exceeds, ylen / 2 + 2.
      int64_t cap = (int64_t)ceil((double)m.ylen / fs_d * h_band_f0[i] * 3.0) + 64;
      if (caps_worst) cap = m.ylen / 2 + 2;
      else if (caps_given) cap = std::max<int64_t>(ctx->hv_caps_next[(size_t)u * n_bands + i], 8);
      e_off[(size_t)u * n_bands + i] = e_tot;
      e_cap[(size_t)u * n_bands + i] = cap;
      e_tot += 4 * cap;
    }
    max_len = std::max(max_len, m.nd + 2 * kFPad);
    max_ylen = std::max(max_ylen, m.ylen);
    max_nf1 = std::max(max_nf1, m.nf1);
    max_nf = std::max(max_nf, m.nf);
  }
  ctx->hv_caps_next.clear();  // explicit capacities serve one call
  int h_max = 0;
  for (int i = 0; i < n_bands; ++i) h_max = std::max(h_max, (int)h_band_half[i]);
#ifndef WH_HV_BAND_OLS
#define WH_HV_BAND_OLS 1
#endif
  // the block of kOlsN inputs must cover H + h + 1 + kOlsValid + 2 outputs for every channel
  const bool use_ols = WH_HV_BAND_OLS && (2 * h_max + 1 + wh::kOlsValid + 2 <= wh::kOlsN) && pad >= h_max + 1;
#ifndef WH_HV_RAWDET
#define WH_HV_RAWDET 1  // raw candidates + detection in one transposed pass (hv_rawdet_kernel); 0: hv_raw_kernel + hv_detect_kernel
#endif
#ifndef WH_HV_RAWDET_MIN_TILES
#define WH_HV_RAWDET_MIN_TILES 8192  // (~52 utterances of 10 s; measured: 1 / 8 / 32 / 64 utterances 0.46 / 0.46 / 0.96 / 1.54 ms fused against 0.15 / 0.29 / 0.82 / 1.52 ms for the pair)
#endif
  // The transposed kernel runs one wave per (utterance, 64-frame tile), each walking all channels: a handful of utterances
  // is a few hundred waves with a 152-step chain each (one 4.6 s utterance: 74), where hv_raw_kernel spreads the same work
  // over channels x utterances x segments workgroups — the reference's own benchmark, ONE encode of its test recording,
  // went from 2.5 to 3.0 ms.  Below WH_HV_RAWDET_MIN_TILES tiles in the batch the pair of kernels runs instead
  // (WH_HV_RAWDET_MIN_TILES in the environment overrides it: tests run both forms on the same input).
  int64_t batch_tiles = 0;
  for (int u = 0; u < B; ++u) batch_tiles += meta[u].ntile;
  static const long rawdet_min_env = getenv("WH_HV_RAWDET_MIN_TILES") ? atol(getenv("WH_HV_RAWDET_MIN_TILES")) : -1;
  const int64_t rawdet_min = rawdet_min_env >= 0 ? rawdet_min_env : WH_HV_RAWDET_MIN_TILES;
  const bool use_rawdet = WH_HV_RAWDET && use_ols && batch_tiles >= rawdet_min;  // (its cursor hints come from the overlap-save walker)
  // the [channel][frame] candidate map (12 GB per 1024 x 10 s) and its bit map exist only where something reads them
  const bool need_map = !use_rawdet || dbg_raw != nullptr;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t off = 0;
  const size_t o_tmp = off; off += al(sizeof(double) * t_tot);
  const size_t o_y = off; off += al(sizeof(double) * y_tot);
  const size_t o_z = off; off += al(sizeof(double) * z_tot);
  const size_t o_mean = off; off += al(sizeof(double) * B * (1 + kMeanParts));  // means, then the partial sums
  const size_t o_e = off; off += al(sizeof(double) * e_tot);
  const size_t o_raw = off; off += need_map ? al(sizeof(double) * f1_tot * n_bands) : 0;
  const size_t o_live = off; off += need_map ? al(sizeof(unsigned long long) * (size_t)l_tot) : 0;
  const size_t o_hint = off; off += use_rawdet ? al(sizeof(int32_t) * 4 * (size_t)l_tot) : 0;  // [utterance][tile][channel][train]
  const size_t o_dc = off; off += al(sizeof(double) * f1_tot * kMaxC);
  const size_t o_dn = off; off += al(sizeof(int32_t) * f1_tot);
  const size_t o_rf0 = off; off += al(sizeof(double) * f1_tot * kRows);
  const size_t o_rsc = off; off += al(sizeof(double) * f1_tot * kRows);
  const size_t o_keep = off; off += al(sizeof(uint32_t) * 4 * f1_tot);
  const size_t o_lst = off; off += al(sizeof(int64_t) * f1_tot);
  const size_t o_ct = off; off += al(contour_workspace_bytes(f1_tot, B));
  // overlap-save band filters: tile spectra of every utterance + the channels' tap spectra
  std::vector<int64_t> tile_off(B + 1, 0);
  int64_t max_tiles = 0;
  for (int u = 0; u < B; ++u) {
    const int64_t t = (meta[u].ylen + wh::kOlsValid - 1) / wh::kOlsValid;
    tile_off[u + 1] = tile_off[u] + t;
    max_tiles = std::max(max_tiles, t);
  }
  const size_t spec_bins = wh::kOlsN / 2 + 1;
  const size_t o_tspec = off; off += use_ols ? al(sizeof(double2) * spec_bins * n_bands) : 0;
  const size_t o_zspec = off; off += use_ols ? al(sizeof(double2) * spec_bins * (size_t)tile_off[B]) : 0;
  const size_t o_tre = off; off += use_ols ? al(sizeof(double) * spec_bins * n_bands) : 0;  // the real (zero-phase) tap spectra
  if (int rc = wh::ws_reserve(ctx, off)) return rc;
  char* ws = reinterpret_cast<char*>(ctx->ws);
  HvUtt* d_meta = nullptr;
  double* d_tmp = reinterpret_cast<double*>(ws + o_tmp);
  double* d_y = reinterpret_cast<double*>(ws + o_y);
  double* d_z = reinterpret_cast<double*>(ws + o_z);
  double* d_mean = reinterpret_cast<double*>(ws + o_mean);
  double* d_e = reinterpret_cast<double*>(ws + o_e);
  // the lists' counts: a buffer of their own (wh_harvest_event_counts reads them after later stages have used the scratch)
  int32_t* d_cnt = nullptr;
  {
    void* p = nullptr;
    if (int rc = wh::persistent_scratch(ctx, "hv.counts", sizeof(int32_t) * (size_t)B * n_bands * 4, &p)) return rc;
    d_cnt = reinterpret_cast<int32_t*>(p);
    ctx->hv_last_cnt = d_cnt;
    ctx->hv_last_cnt_lists = (int64_t)B * n_bands;
  }
  wh::BandJob* d_jobs = nullptr;
  double* d_raw = reinterpret_cast<double*>(ws + o_raw);
  unsigned long long* d_live = reinterpret_cast<unsigned long long*>(ws + o_live);
  int32_t* d_hint = reinterpret_cast<int32_t*>(ws + o_hint);
  double* d_dc = reinterpret_cast<double*>(ws + o_dc);
  int32_t* d_dn = reinterpret_cast<int32_t*>(ws + o_dn);
  double* d_rf0 = reinterpret_cast<double*>(ws + o_rf0);
  double* d_rsc = reinterpret_cast<double*>(ws + o_rsc);
  uint32_t* d_keep = reinterpret_cast<uint32_t*>(ws + o_keep);
  int64_t* d_lst = reinterpret_cast<int64_t*>(ws + o_lst);
  double* d_taps = nullptr;
  double* d_bf = nullptr;
  int32_t* d_ti = nullptr;
  char* d_ct = ws + o_ct;
  std::vector<wh::BandJob> jobs((size_t)B * n_bands);
  for (int u = 0; u < B; ++u)
    for (int i = 0; i < n_bands; ++i) {
      wh::BandJob& j = jobs[(size_t)u * n_bands + i];
      j.z = d_z + meta[u].z_off;
      j.M = meta[u].ylen;
      j.edges = d_e + e_off[(size_t)u * n_bands + i];
      j.cap = e_cap[(size_t)u * n_bands + i];
      j.counts = d_cnt + ((int64_t)u * n_bands + i) * 4;
      if (use_rawdet) {
        j.hints = d_hint + (meta[u].l_off + i) * 4;
        j.hint_tiles = meta[u].ntile;
        j.hint_spt = kRawTile * fs_d / 1000.0;
        j.hint_inv_spt = 1.0 / j.hint_spt;
        j.hint_stride = n_bands * 4;
      }
    }
  {
    std::vector<double> taps(h_band_taps, h_band_taps + taps_total), bf(h_band_f0, h_band_f0 + n_bands);
    if (int rc = wh::persistent_upload(ctx, st, "hv.meta", meta, &d_meta)) return rc;
    if (int rc = wh::persistent_upload(ctx, st, "hv.jobs", jobs, &d_jobs)) return rc;
    if (int rc = wh::persistent_upload(ctx, st, "hv.taps", taps, &d_taps)) return rc;
    if (int rc = wh::persistent_upload(ctx, st, "hv.band_f0", bf, &d_bf)) return rc;
    if (int rc = wh::persistent_upload(ctx, st, "hv.tapinfo", ti, &d_ti)) return rc;
  }

  // ---- decimation -----------------------------------------------------------------------------------
  if (filtered) {
    Tdf2 c;
    c.b0 = h_ba[0]; c.b1 = h_ba[1]; c.b2 = h_ba[2]; c.b3 = h_ba[3];
    c.a1 = h_ba[5] / h_ba[4]; c.a2 = h_ba[6] / h_ba[4]; c.a3 = h_ba[7] / h_ba[4];
    if (h_ba[4] != 1.0) { c.b0 /= h_ba[4]; c.b1 /= h_ba[4]; c.b2 /= h_ba[4]; c.b3 /= h_ba[4]; }
    c.zi0 = h_zi[0]; c.zi1 = h_zi[1]; c.zi2 = h_zi[2];
    const double rad = tdf2_pole_radius(c.a1, c.a2, c.a3);
    int warm = 64;
    if (rad > 0 && rad < 1) warm = (int)ceil(-46.0 / log(rad));
    warm = ((warm + 63) / 64) * 64;
    if (!(rad < 0.9999)) warm = 1 << 30;
    const int chunks = (int)((max_len + kHChunk - 1) / kHChunk);
    dim3 gi((chunks + 63) / 64, B);
    { wh::KernelTimer _kt(ctx, st, "hv_iir_fwd_kernel"); hipLaunchKernelGGL(hv_iir_fwd_kernel, gi, dim3(64), 0, st, x, d_meta, c, warm, d_tmp); }
    WH_LAUNCH_CHECK("hv_iir_fwd_kernel");
    { wh::KernelTimer _kt(ctx, st, "hv_iir_bwd_kernel"); hipLaunchKernelGGL(hv_iir_bwd_kernel, gi, dim3(64), 0, st, d_meta, c, warm, r, d_tmp, d_y); }
    WH_LAUNCH_CHECK("hv_iir_bwd_kernel");
  } else {
    { wh::KernelTimer _kt(ctx, st, "hv_copy_kernel"); hipLaunchKernelGGL(hv_copy_kernel, dim3((unsigned)((max_ylen + 255) / 256), B), dim3(256), 0, st, x, d_meta, d_y); }
    WH_LAUNCH_CHECK("hv_copy_kernel");
  }
  { wh::KernelTimer _kt(ctx, st, "hv_mean_kernel"); hipLaunchKernelGGL(hv_mean_part_kernel, dim3(kMeanParts, B), dim3(256), 0, st, d_meta, d_y, d_mean + B); }
  { wh::KernelTimer _kt(ctx, st, "hv_mean_kernel"); hipLaunchKernelGGL(hv_mean_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, d_meta, d_mean + B, B, d_mean); }
  WH_LAUNCH_CHECK("hv_mean_kernel");
  { wh::KernelTimer _kt(ctx, st, "hv_pad_kernel"); hipLaunchKernelGGL(hv_pad_kernel, dim3((unsigned)((max_ylen + 2 * pad + 255) / 256), B), dim3(256), 0, st, d_meta, d_y, d_mean, pad, d_z); }
  WH_LAUNCH_CHECK("hv_pad_kernel");
  if (dbg_y) WH_CHECK(hipMemcpyAsync(dbg_y, d_y, sizeof(double) * y_tot, hipMemcpyDeviceToDevice, st));

  // ---- 152 channels: FIR + crossings, then per-frame raw candidates ------------------------------------
  if (use_ols) {
    int64_t* d_tile_off = nullptr;
    if (int rc = wh::persistent_upload(ctx, st, "hv.tile_off", tile_off, &d_tile_off)) return rc;
    if (int rc = wh::launch_band_events_ols(ctx, st, d_jobs, n_bands, B, pad, h_max, d_taps, d_ti, d_ti + n_bands,
                                            d_ti + 2 * n_bands, d_tile_off, max_tiles,
                                            reinterpret_cast<double2*>(ws + o_tspec), reinterpret_cast<double*>(ws + o_tre), reinterpret_cast<double2*>(ws + o_zspec),
                                            ctx->d_flags + WH_FLAG_EVENT_OVERFLOW))
      return rc;
  } else if (int rc = wh::launch_band_events(ctx, st, d_jobs, n_bands, B, pad, d_taps, d_ti, d_ti + n_bands,
                                             d_ti + 2 * n_bands, max_lb, true, ctx->d_flags + WH_FLAG_EVENT_OVERFLOW)) {
    return rc;
  }
  if (use_rawdet) {
    int64_t max_ntile = 0;
    for (int u = 0; u < B; ++u) max_ntile = std::max(max_ntile, meta[u].ntile);
    double* d_rawdbg = nullptr;
    if (dbg_raw) d_rawdbg = d_raw;
    const int rd_xcd = B >= 8 ? 8 : 1;
    { wh::KernelTimer _kt(ctx, st, "hv_rawdet_kernel"); hipLaunchKernelGGL(hv_rawdet_kernel, dim3((unsigned)((int64_t)((B + rd_xcd - 1) / rd_xcd) * rd_xcd * max_ntile)), dim3(kRawTile), 0, st, d_meta, d_jobs, d_bf, d_hint, n_bands, B, rd_xcd, (int)max_ntile, fs_d, f0_floor, f0_ceil, d_dc, d_dn, d_rawdbg); }
    WH_LAUNCH_CHECK("hv_rawdet_kernel");
    if (dbg_raw) WH_CHECK(hipMemcpyAsync(dbg_raw, d_raw, sizeof(double) * f1_tot * n_bands, hipMemcpyDeviceToDevice, st));
  } else {
  // frames of an (utterance, channel) cut into segments with a workgroup each while the grid is a few rounds of the chip
  // (2560 workgroups at ten per CU): 1.80 -> 1.70 ms at 64 utterances; large batches keep one (no second cursor search)
  const int raw_segs = WH_HV_RAW_SEGS > 1 ? WH_HV_RAW_SEGS : ((int64_t)n_bands * B < 16 * 2560 ? 4 : 1);
  { wh::KernelTimer _kt(ctx, st, "hv_raw_kernel"); hipLaunchKernelGGL(hv_raw_kernel, dim3(n_bands, B, raw_segs), dim3(kRawTile), 0, st, d_meta, d_jobs, d_bf, n_bands, fs_d, f0_floor, f0_ceil, d_raw, d_live, dbg_raw ? 1 : 0); }
  WH_LAUNCH_CHECK("hv_raw_kernel");
  if (dbg_raw) WH_CHECK(hipMemcpyAsync(dbg_raw, d_raw, sizeof(double) * f1_tot * n_bands, hipMemcpyDeviceToDevice, st));
  { wh::KernelTimer _kt(ctx, st, "hv_detect_kernel"); hipLaunchKernelGGL(hv_detect_kernel, dim3((unsigned)((max_nf1 + 255) / 256), B), dim3(256), 0, st, d_meta, n_bands, d_raw, d_live, d_dc, d_dn); }
  WH_LAUNCH_CHECK("hv_detect_kernel");
  }
  // ---- refinement + pruning ------------------------------------------------------------------------------
  {
    int tw_n = 1;  // transform length of the longest window (harvest.py:171-172): 2 * 2^ceil(log2(2*hmax+1))
    while (tw_n < 2 * hmax + 1) tw_n <<= 1;
    tw_n <<= 1;
    if (tw_n > 2048) tw_n = 0;  // 32 KB of LDS at most for the table; beyond that gather from the global tables
#ifndef WH_HV_LDS_TWIDDLES
#define WH_HV_LDS_TWIDDLES 1  // 0 (the sanitizer build): always the global tables — the LDS form needs its block at LDS
#endif                        // address 0 and traps otherwise, and the sanitizer puts bookkeeping of its own there
    if (!WH_HV_LDS_TWIDDLES) tw_n = 0;
    const bool use_wtab = WH_HV_WIN_TABLE && tw_n != 0;
    const int fpb = refine_frames(use_wtab);
    const int seglen = 2 * hmax + 8 + (fpb - 1) * ((int)ceil(fs_d / 1000.0) + 1);
    const size_t lds = sizeof(double2) * (size_t)tw_n + sizeof(double) * (size_t)((seglen + 1) & ~1) +
                       (sizeof(double) + 3 * sizeof(int)) * (size_t)refine_item_cap(use_wtab) + sizeof(int) * (72 + 5 * fpb + 1);
    // 16-sample rotation (sin, cos)(16*pi*dx) of the window phase for every half length (hv_refine_row)
    double2* d_rot = nullptr;
    {
      std::vector<double2> rot(hmax + 2);
      for (int h = 0; h <= hmax + 1; ++h) {
        const double wlit = (2 * (double)h + 1) / fs_d;
        const double dx = 2.0 / (fs_d * wlit);
        rot[h] = make_double2(sin(M_PI * (16 * dx)), cos(M_PI * (16 * dx)));
      }
      if (int rc = wh::persistent_upload(ctx, st, "hv.rot", rot, &d_rot)) return rc;
    }
    // (w(j), dw(j)) of every window length, row hwl at offset hwl*(hwl+2) between two zero pairs (hv_refine_row, WTAB).  Built once per (rate,
    // longest window) and kept on the device.
    const double2* d_wtab = nullptr;
    if (use_wtab) {
      char key[96];
      snprintf(key, sizeof key, "hv.wtab2:%.17g:%d", fs_d, hmax);
      auto it = ctx->tables.find(key);
      if (it == ctx->tables.end()) {
        // row h at offset h*(h+2): a zero pair, the 2h+1 window pairs, a zero pair
        std::vector<double> tab((size_t)2 * (hmax + 2) * (hmax + 4), 0.0);
        std::vector<double> mw;
        for (int h = 0; h <= hmax + 1; ++h) {
          const int Lh = 2 * h + 1;
          const double wlit = (2 * (double)h + 1) / fs_d;
          mw.assign(Lh, 0.0);
          for (int j = 0; j < Lh; ++j) {
            // index_raw keeps round_matlab's +0.5 and the +0.001 "first-aid" (harvest.py:178, Q1): the window is
            // evaluated 0.501 samples late
            const double c = cos(M_PI * (2 * (((double)(j - h) + (0.001 + 0.5) - 1.0) / fs_d) / wlit));
            mw[j] = 0.42 + 0.5 * c + 0.08 * (2 * c * c - 1);
          }
          double* row = tab.data() + 2 * ((size_t)h * (h + 2) + 1);
          for (int j = 0; j < Lh; ++j) {
            double dw;
            if (j == 0) dw = Lh > 1 ? -mw[1] / 2 : 0.0;
            else if (j == Lh - 1) dw = mw[j - 1] / 2;
            else dw = -((mw[j + 1] - mw[j]) + (mw[j] - mw[j - 1])) / 2;
            row[2 * j] = mw[j];
            row[2 * j + 1] = dw;
          }
        }
        const double* d = nullptr;
        if (int rc = wh::const_table(ctx, key, tab, &d)) return rc;
        d_wtab = reinterpret_cast<const double2*>(d);
      } else {
        d_wtab = reinterpret_cast<const double2*>(it->second);
      }
    }
    const dim3 grid((unsigned)((max_nf1 + fpb - 1) / fpb), B);
    // slots of the work list a round may fill: all of them, unless WH_HV_ITEM_CAP_RT (tests: the several-rounds path,
    // which real input reaches only with > ~70 candidates per frame over a whole block) says fewer
    static const int cap_env = getenv("WH_HV_ITEM_CAP_RT") ? atoi(getenv("WH_HV_ITEM_CAP_RT")) : 0;
    int item_cap = refine_item_cap(use_wtab);
    if (cap_env >= kRows && cap_env < item_cap) item_cap = cap_env;
#define WH_REFINE_LAUNCH(TWL_, WTAB_)                                                                                   \
  {                                                                                                                     \
    if (int rc = wh::allow_lds(&hv_refine_kernel<TWL_, WTAB_>, lds)) return rc;                                         \
    wh::KernelTimer _kt(ctx, st, "hv_refine_kernel");                                                                   \
    hipLaunchKernelGGL((hv_refine_kernel<TWL_, WTAB_>), grid, dim3(256), lds, st, d_meta, d_y, d_dc, d_dn, fs_d, f0_floor, \
                       f0_ceil, hmax, seglen, ctx->d_twiddle, tw_n, d_rot, d_wtab, d_rf0, d_rsc, d_lst, item_cap);              \
  }
    if (tw_n && use_wtab) WH_REFINE_LAUNCH(true, true)
    else if (tw_n) WH_REFINE_LAUNCH(true, false)
    else WH_REFINE_LAUNCH(false, false)
#undef WH_REFINE_LAUNCH
    WH_LAUNCH_CHECK("hv_refine_kernel");
  }
  { wh::KernelTimer _kt(ctx, st, "hv_prune_kernel"); hipLaunchKernelGGL(hv_prune_kernel, dim3((unsigned)((max_nf1 + kPruneFrames - 1) / kPruneFrames), B), dim3(256), 0, st, d_meta, d_rf0, d_lst, d_keep); }
  WH_LAUNCH_CHECK("hv_prune_kernel");
  // ---- contour, smoothing, 5 ms pick -------------------------------------------------------------------------
  return harvest_contour(ctx, st, B, d_meta, meta, f1_tot, max_nf1, max_nf, d_rf0, d_rsc, d_lst, d_keep, d_ct, tp, f0_out, vuv_out,
                         dbg_f0_1ms);
}

// Capacities of Harvest's zero-crossing lists (include/world_hip.h).  h_caps != NULL: one capacity per (utterance,
// channel) for the NEXT wh_harvest of this context (n = utterances x channels of that call); NULL with n == -1: every
// list sized for the bound no signal exceeds (ylen / 2 + 2) until reset; NULL with n == 0: back to the estimate.
extern "C" int wh_harvest_set_event_caps(wh_ctx* ctx, const int64_t* h_caps, int64_t n) {
  if (!ctx) return wh::fail_msg("wh_harvest_set_event_caps", "null context");
  if (h_caps) {
    if (n <= 0) return wh::fail_msg("wh_harvest_set_event_caps", "n must be utterances x channels");
    ctx->hv_caps_next.assign(h_caps, h_caps + n);
    return 0;
  }
  if (n != 0 && n != -1) return wh::fail_msg("wh_harvest_set_event_caps", "without capacities n is 0 (estimate) or -1 (bound)");
  ctx->hv_caps_next.clear();
  ctx->hv_caps_worst = n == -1;
  return 0;
}

// What the last wh_harvest of this context counted: h_caps_out[u * n_bands + i] = the longest of the four crossing
// trains of channel i of utterance u — exact also when the call overflowed its lists (WH_FLAG_EVENT_OVERFLOW), so a
// repeat with these capacities fits.  Waits for `stream`.
extern "C" int wh_harvest_event_counts(wh_ctx* ctx, void* stream, int64_t* h_caps_out, int64_t n) {
  if (!ctx || !h_caps_out) return wh::fail_msg("wh_harvest_event_counts", "null argument");
  WH_ENTER(ctx);
  if (!ctx->hv_last_cnt || n != ctx->hv_last_cnt_lists)
    return wh::fail_msg("wh_harvest_event_counts", "n is not utterances x channels of this context's last wh_harvest");
  std::vector<int32_t> cnt((size_t)n * 4);
  hipStream_t st = (hipStream_t)stream;
  WH_CHECK(hipMemcpyAsync(cnt.data(), ctx->hv_last_cnt, sizeof(int32_t) * cnt.size(), hipMemcpyDeviceToHost, st));
  WH_CHECK(hipStreamSynchronize(st));
  for (int64_t i = 0; i < n; ++i) {
    int32_t m = cnt[4 * i];
    for (int t = 1; t < 4; ++t) m = std::max(m, cnt[4 * i + t]);
    h_caps_out[i] = m;
  }
  return 0;
}
