// Zero-crossing event extraction and event→frame interpolation shared by DIO and Harvest.
// Reference semantics: ZeroCrossingEngine (world/dio.py:190-204 = world/harvest.py:283-297) and
// get_f0_candidates / GetF0Candidates (world/dio.py:156-185, world/harvest.py:499-529).
// The reference builds four ragged NumPy arrays per band and four interp1d objects; here the
// crossings of a tile are flagged in registers, compacted in order with ONE packed 4x16-bit block
// scan, and appended to per-(band, train) edge lists; frames later binary-search those lists.
#pragma once
#include "wh_device.h"
#include "wh_math.h"

#ifndef WH_OLS_ABLATE
#define WH_OLS_ABLATE 0  // timing experiments on the overlap-save band walker (never shipped; results differ): 1: neither the inverse
                         // transform nor the crossing pass, 2: no crossing pass, 3: the crossing pass without its second walk
#endif


namespace wh {

__device__ __forceinline__ unsigned long long wave_scan_incl_u64(unsigned long long v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long u = (unsigned long long)__shfl_up((long long)v, o, 64);
    if (lane >= o) v += u;
  }
  return v;
}

// Append the four crossing trains of one signal tile to their edge lists, preserving order.
//   sig[i] = s[t0+i] for i in [0, tile+2) (LDS; values beyond M are ignored)
//   train 0: negative-going crossings of s, 1: positive-going, 2: negative-going of diff(s), 3: positive-going
//   edge value = (1-based sample position) - v[i]/(v[i+1]-v[i])   (dio.py:201)
// edges: [4][cap]; base_cnt[4]: running counts (block-uniform registers, updated).
// 256 threads, tile == 1024 (4 positions per thread).  Contains barriers.
// STRIDE: distance in doubles between consecutive samples of sig (2: one component of an interleaved complex buffer).
template <int STRIDE = 1>
__device__ __forceinline__ void emit_crossings(const double* sig, int64_t t0, int64_t M, int tile, double* edges,
                                               int64_t cap, int* base_cnt, unsigned long long* scratch,
                                               int32_t* overflow_flag) {
  const int tid = threadIdx.x;
  double fine[4][4];
  unsigned mask[4] = {0, 0, 0, 0};
  unsigned long long packed = 0;
  const int per = tile / WH_BLOCK;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q >= per) break;
    const int i = tid * per + q;
    const int64_t g = t0 + i;
    const double a = sig[i * STRIDE], b = sig[(i + 1) * STRIDE], c = sig[(i + 2) * STRIDE];
    if (g + 1 < M && a * b < 0) {  // crossing of s between g and g+1
      const double fe = (double)(g + 1) - wh::fdiv(a, b - a);
      if (b < a) {
        mask[0] |= 1u << q;
        fine[0][q] = fe;
      } else if (b > a) {
        mask[1] |= 1u << q;
        fine[1][q] = fe;
      }
    }
    if (g + 2 < M) {
      const double d0 = b - a, d1 = c - b;
      if (d0 * d1 < 0) {
        const double fe = (double)(g + 1) - wh::fdiv(d0, d1 - d0);
        if (d1 < d0) {
          mask[2] |= 1u << q;
          fine[2][q] = fe;
        } else if (d1 > d0) {
          mask[3] |= 1u << q;
          fine[3][q] = fe;
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) packed |= (unsigned long long)__popc(mask[t]) << (16 * t);
  const unsigned long long incl = wave_scan_incl_u64(packed);
  const int w = tid >> 6;
  __syncthreads();
  if ((tid & 63) == 63) scratch[w] = incl;
  __syncthreads();
  unsigned long long excl = incl - packed;
  unsigned long long total = 0;
#pragma unroll
  for (int i = 0; i < WH_BLOCK / 64; ++i) {
    if (i < w) excl += scratch[i];
    total += scratch[i];
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    int pos = base_cnt[t] + (int)((excl >> (16 * t)) & 0xFFFF);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (mask[t] & (1u << q)) {
        if (pos < cap) edges[(int64_t)t * cap + pos] = fine[t][q];
        else atomicOr(overflow_flag, 1);
        ++pos;
      }
    }
    base_cnt[t] += (int)((total >> (16 * t)) & 0xFFFF);
  }
}

// First walk of the crossing pass over one thread's PER consecutive positions (s[0 .. PER + 2) readable): which of them
// carry a crossing, per train.  Position q is sample g = g_first + q; a crossing of the signal needs g + 1 < M, one of
// its first difference g + 2 < M, and q < n_pos: `left` = M - g_first and n_pos turn into two bit masks applied once,
// after the walk, instead of two 64-bit compares per position.  With a*b < 0 the two values differ, so ONE compare
// tells the direction (it was two); the difference d1 of a position is d0 of the next.  All samples come in one round
// of LDS reads (left to the compiler the walk was read - wait - test, PER / 2 times over).
template <int STRIDE, int PER>
__device__ __forceinline__ void crossing_flags(const double* s_ptr, int64_t left, int n_pos, unsigned* m01_out,
                                               unsigned* m23_out) {
  double s[PER + 2];
#pragma unroll
  for (int q = 0; q < PER + 2; ++q) s[q] = s_ptr[q * STRIDE];
  unsigned m01 = 0, m23 = 0;
  double d0 = s[1] - s[0];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const double a = s[q], b = s[q + 1], c = s[q + 2];
    const unsigned f01 = a * b < 0 ? (b < a ? 1u : 0x10000u) : 0u;
    const double d1 = c - b;
    const unsigned f23 = d0 * d1 < 0 ? (d1 < d0 ? 1u : 0x10000u) : 0u;
    m01 |= f01 << q;
    m23 |= f23 << q;
    d0 = d1;
  }
  // positions q with g + 1 < M  <=>  q < left - 1;  g + 2 < M  <=>  q < left - 2;  and q < n_pos
  int64_t l1 = left - 1, l2 = left - 2;
  l1 = l1 < 0 ? 0 : (l1 > n_pos ? n_pos : l1);
  l2 = l2 < 0 ? 0 : (l2 > n_pos ? n_pos : l2);
  const unsigned k1 = (1u << (int)l1) - 1u, k2 = (1u << (int)l2) - 1u;  // (<= 16 bits)
  *m01_out = m01 & (k1 | (k1 << 16));
  *m23_out = m23 & (k2 | (k2 << 16));
}

// Second walk: the flagged positions of one thread, ascending, one divide per crossing — the signal's crossings in one
// loop, the first difference's in another (a wave runs a loop as long as its busiest lane: two loops of one divide take
// max + max rounds, one loop with both bodies took max-of-sums rounds of two).  put(train, slot, edge) stores; pos[4] are
// the thread's first slots.  Edge = (1-based position) - v[i] / (v[i+1] - v[i])  (dio.py:201).
template <int STRIDE, class Put>
__device__ __forceinline__ void crossing_edges(const double* s_ptr, int64_t g_first, unsigned m01, unsigned m23, int* pos,
                                               Put put) {
  unsigned any = (m01 | (m01 >> 16)) & 0xFFFFu;
  while (any) {
    const int q = __ffs(any) - 1;
    any &= any - 1;
    const double a = s_ptr[q * STRIDE], b = s_ptr[(q + 1) * STRIDE];
    const int t = (m01 >> q) & 1u ? 0 : 1;
    const double fe = (double)(g_first + q + 1) - wh::fdiv(a, b - a);
    put(t, pos[t], fe);
    ++pos[t];
  }
  any = (m23 | (m23 >> 16)) & 0xFFFFu;
  while (any) {
    const int q = __ffs(any) - 1;
    any &= any - 1;
    const double a = s_ptr[q * STRIDE], b = s_ptr[(q + 1) * STRIDE], c = s_ptr[(q + 2) * STRIDE];
    const int t = (m23 >> q) & 1u ? 2 : 3;
    const double d0 = b - a, d1 = c - b;
    const double fe = (double)(g_first + q + 1) - wh::fdiv(d0, d1 - d0);
    put(t, pos[t], fe);
    ++pos[t];
  }
}

// The same for a tile of PER * 256 positions in ONE pass (one block scan and two barriers whatever the tile length):
// the first walk only flags the crossings (no divides, nothing kept but four PER-bit masks), the second re-reads the
// flagged samples and writes the edge positions at the offsets the scan produced.  Used by the overlap-save band
// walker, whose tiles are 3584 samples (14 positions per thread).
//
// Cursor hints (Harvest, round 6; hint == nullptr: none).  The frames are cut into tiles of 64 (hv_rawdet_kernel: one
// wave per tile walks all channels); tile T starts at sample s_T = floor(T * hint_spt), hint_spt = 64 ms in samples.
// hint[T * hint_stride + train] receives the number of this train's crossings at positions before s_T — where that
// tile's search of the edge list starts, give or take the few entries the consumer allows for.  The thread whose
// positions hold s_T knows it: its first slot plus the crossings it flagged in front of s_T.  A hint is advice: the
// consumer clamps it into the list and checks every frame's answer against the window it staged (exact either way).
template <int STRIDE, int PER>
__device__ __forceinline__ void emit_crossings_block(const double* sig, int64_t t0, int64_t M, double* edges, int64_t cap,
                                                     int* base_cnt, unsigned long long* scratch,
                                                     int32_t* overflow_flag, int32_t* hint = nullptr,
                                                     double hint_spt = 0.0, double hint_inv_spt = 0.0,
                                                     int64_t hint_tiles = 0, int hint_stride = 0) {
  static_assert(PER <= 16, "masks are 16 bits, counts 16 bits per train");
  const int tid = threadIdx.x;
  const int i0 = tid * PER;
  unsigned m01, m23;  // bits [0,16): negative-going, [16,32): positive-going
  crossing_flags<STRIDE, PER>(sig + (int64_t)i0 * STRIDE, M - (t0 + i0), PER, &m01, &m23);
  const unsigned long long packed = (unsigned long long)__popc(m01 & 0xFFFFu) | ((unsigned long long)__popc(m01 >> 16) << 16) |
                                    ((unsigned long long)__popc(m23 & 0xFFFFu) << 32) |
                                    ((unsigned long long)__popc(m23 >> 16) << 48);
  const unsigned long long incl = wave_scan_incl_u64(packed);
  const int w = tid >> 6;
  __syncthreads();
  if ((tid & 63) == 63) scratch[w] = incl;
  __syncthreads();
  unsigned long long excl = incl - packed;
  unsigned long long total = 0;
#pragma unroll
  for (int i = 0; i < WH_BLOCK / 64; ++i) {
    if (i < w) excl += scratch[i];
    total += scratch[i];
  }
  int pos[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) pos[t] = base_cnt[t] + (int)((excl >> (16 * t)) & 0xFFFF);
  if (hint) {
    // the tile boundary in [g0, g0 + PER), if there is one (tiles are hundreds of samples long): the first tile that
    // starts at or behind g0, found with a multiply by the reciprocal and corrected by one.  32-bit indices (a signal of
    // 2^31 decimated samples is days long): the 64-bit conversions are instruction sequences, and this runs per thread
    // and channel-tile
    const int g0 = (int)(t0 + i0);
    int T = (int)((double)g0 * hint_inv_spt);
    int sT = (int)((double)T * hint_spt);
    if (sT < g0) {
      ++T;
      sT = (int)((double)T * hint_spt);
    }
    if (sT >= g0 && sT < g0 + PER && T < hint_tiles) {
      const unsigned below = (1u << (int)(sT - g0)) - 1u;
      int32_t* h = hint + (int64_t)T * hint_stride;
      h[0] = pos[0] + __popc(m01 & below);
      h[1] = pos[1] + __popc((m01 >> 16) & below);
      h[2] = pos[2] + __popc(m23 & below);
      h[3] = pos[3] + __popc((m23 >> 16) & below);
    }
  }
  bool over = false;
#if WH_OLS_ABLATE == 3
  over = m01 == 0xDEADBEEFu && m23 == m01;  // (timing experiment: the pass without its second walk)
#else
  crossing_edges<STRIDE>(sig + (int64_t)i0 * STRIDE, t0 + i0, m01, m23, pos, [&](int t, int at, double fe) {
    if (at < cap) stg(edges + (int64_t)t * cap + at, fe);
    else over = true;
  });
#endif
  if (over) atomicOr(overflow_flag, 1);
#pragma unroll
  for (int t = 0; t < 4; ++t) base_cnt[t] += (int)((total >> (16 * t)) & 0xFFFF);
}

// Interpolate the four interval-F0 trains at time t (linear, end-segment extrapolation — SciPy's
// interp1d(..., fill_value='extrapolate') arithmetic) and reduce: mean and, optionally, ddof=1 std.
// Fewer than 3 intervals in any train → (0, 1000) (dio.py:159-162,182-184).
__device__ __forceinline__ void interp_four_trains(const double* __restrict__ edges, int64_t cap,
                                                   const int32_t* __restrict__ cnt, double fs, double t, bool want_dev,
                                                   double* cand, double* dev) {
  bool usable = true;
#pragma unroll
  for (int k = 0; k < 4; ++k) usable = usable && (cnt[k] - 1 >= 3);
  if (!usable) {
    *cand = 0.0;
    *dev = 1000.0;
    return;
  }
  double v[4];
  // interval locations (e[i]+e[i+1])/2/fs are formed with one multiply by 0.5/fs: the divide would sit in every step
  // of the binary searches (~40 FP64 divides per call); the interpolant is continuous across its nodes, so the last
  // bit of a location cannot change the result by more than its own rounding
  const double half_inv_fs = 0.5 / fs;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double* e = edges + (int64_t)k * cap;
    const int ni = cnt[k] - 1;  // intervals; location i = (e[i]+e[i+1])/2/fs
    int lo = 0, hi = ni;        // lower_bound: count of locations < t
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const double loc = (e[mid] + e[mid + 1]) * half_inv_fs;
      if (loc < t) lo = mid + 1; else hi = mid;
    }
    int ih = lo < 1 ? 1 : (lo > ni - 1 ? ni - 1 : lo);
    const int il = ih - 1;
    const double x_lo = (e[il] + e[il + 1]) * half_inv_fs;
    const double x_hi = (e[ih] + e[ih + 1]) * half_inv_fs;
    const double y_lo = fs / (e[il + 1] - e[il]);
    const double y_hi = fs / (e[ih + 1] - e[ih]);
    const double slope = (y_hi - y_lo) / (x_hi - x_lo);
    v[k] = slope * (t - x_lo) + y_lo;
  }
  const double mean = (((v[0] + v[1]) + v[2]) + v[3]) / 4;
  *cand = mean;
  if (want_dev) {
    const double d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
    *dev = sqrt((((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3) / 3);
  } else {
    *dev = 0.0;
  }
}

}  // namespace wh
