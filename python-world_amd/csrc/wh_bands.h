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
};

constexpr int kBandTile = 1024;
// staging index with 2 doubles of padding every 32: keeps the 16-byte pair reads of lanes 4 samples apart on
// different LDS banks
__device__ __forceinline__ int zpad(int i) { return i + 2 * (i >> 5); }

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
  double* taps = reinterpret_cast<double*>(smem);      // lb
  double* zt = taps + ((lb + 1) & ~1);                 // kBandTile + 2 + lb
  double* sig = zt + ((zpad(kBandTile + 2 + lb) + 3) & ~1);  // kBandTile + 2
  unsigned long long* scan_scratch = reinterpret_cast<unsigned long long*>(sig + kBandTile + 2);  // 8
  for (int k = threadIdx.x; k < ((lb + 1) & ~1); k += 256) taps[k] = k < lb ? taps_all[tap_off[b] + k] : 0.0;
  int base_cnt[4] = {0, 0, 0, 0};
  const int64_t M = job.M;
  // segment blockIdx.z of gridDim.z: a run of whole tiles with its own lists (gridDim.z == 1: the whole signal,
  // straight into the final lists)
  const int nseg = gridDim.z;
  const int64_t tiles = (M + kBandTile - 1) / kBandTile;
  const int64_t tps = (tiles + nseg - 1) / nseg;
  const int64_t t_begin = (int64_t)blockIdx.z * tps * kBandTile;
  const int64_t t_end = (t_begin + tps * kBandTile) < M ? t_begin + tps * kBandTile : M;
  double* edges_out = nseg > 1 ? job.seg_edges + (int64_t)blockIdx.z * 4 * job.seg_cap : job.edges;
  const int64_t cap_out = nseg > 1 ? job.seg_cap : job.cap;
  int32_t* counts_out = nseg > 1 ? job.seg_counts + blockIdx.z * 4 : job.counts;
  for (int64_t t0 = t_begin; t0 < t_end; t0 += kBandTile) {
    __syncthreads();
    const int64_t zlo = t0 + bias[b] + 1 - (lb - 1);
    for (int i = threadIdx.x; i < kBandTile + 2 + lb - 1; i += 256) {
      const int64_t j = zlo + i + pad;
      zt[zpad(i)] = (j >= 0 && j < M + 2 * pad) ? job.z[j] : 0.0;
    }
    __syncthreads();
    if (FMA) {
      // register-tiled FIR: each thread owns 4 consecutive outputs and slides a 4-wide window over the
      // staged input, two taps per step (one 16-byte LDS read for the tap pair, one for the two new inputs)
      const int i0 = threadIdx.x * 4;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      const int top = i0 + (lb - 1);  // input index of output i0 for tap 0
      double r0 = zt[zpad(top)], r1 = zt[zpad(top + 1)], r2 = zt[zpad(top + 2)], r3 = zt[zpad(top + 3)];
      const int lbe = (lb + 1) & ~1;  // taps are zero padded to an even count
      for (int k = 0; k < lbe; k += 2) {
        const double2 tk = *reinterpret_cast<const double2*>(taps + k);
        const int inew = top - k - 2;  // even → aligned pair (z[inew], z[inew+1])
        double2 fresh = make_double2(0.0, 0.0);
        if (inew >= 0) fresh = *reinterpret_cast<const double2*>(zt + zpad(inew));
        a0 = fma(tk.x, r0, a0); a1 = fma(tk.x, r1, a1); a2 = fma(tk.x, r2, a2); a3 = fma(tk.x, r3, a3);
        r3 = r2; r2 = r1; r1 = r0; r0 = fresh.y;
        a0 = fma(tk.y, r0, a0); a1 = fma(tk.y, r1, a1); a2 = fma(tk.y, r2, a2); a3 = fma(tk.y, r3, a3);
        r3 = r2; r2 = r1; r1 = r0; r0 = fresh.x;
      }
      sig[i0] = t0 + i0 < M ? a0 : 0.0;
      sig[i0 + 1] = t0 + i0 + 1 < M ? a1 : 0.0;
      sig[i0 + 2] = t0 + i0 + 2 < M ? a2 : 0.0;
      sig[i0 + 3] = t0 + i0 + 3 < M ? a3 : 0.0;
      if (threadIdx.x < 2) {  // the two look-ahead samples the crossing detector needs
        const int i = kBandTile + threadIdx.x;
        double acc = 0.0;
        if (t0 + i < M)
          for (int k = 0; k < lb; ++k) acc = fma(taps[k], zt[zpad(i + (lb - 1) - k)], acc);
        sig[i] = acc;
      }
    } else {
      for (int i = threadIdx.x; i < kBandTile + 2; i += 256) {
        double acc = 0.0;
        if (t0 + i < M) {
          for (int k = 0; k < lb; ++k) acc += taps[k] * zt[zpad(i + (lb - 1) - k)];
        }
        sig[i] = acc;
      }
    }
    __syncthreads();
    emit_crossings(sig, t0, M, kBandTile, edges_out, cap_out, base_cnt, scan_scratch, flags);
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
  const int zlen = kBandTile + 2 + max_lb;
  const size_t lds = sizeof(double) * (((max_lb + 1) & ~1) + ((zlen + 2 * (zlen >> 5) + 3) & ~1) + kBandTile + 2) + 64;
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

}  // namespace wh
