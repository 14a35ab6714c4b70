// Band FIR + zero-crossing event extraction kernel shared by DIO (7 Nuttall low-pass bands on the
// low-cut filtered 4 kHz signal) and Harvest (152 cosine-modulated Nuttall band-pass channels on the
// 8 kHz signal).  One workgroup per (band, utterance): the signal is walked in 1024-sample tiles, each
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
};

constexpr int kBandTile = 1024;

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
  double* sig = zt + ((kBandTile + 2 + lb + 1) & ~1);  // kBandTile + 2
  unsigned long long* scan_scratch = reinterpret_cast<unsigned long long*>(sig + kBandTile + 2);  // 8
  for (int k = threadIdx.x; k < lb; k += 256) taps[k] = taps_all[tap_off[b] + k];
  int base_cnt[4] = {0, 0, 0, 0};
  const int64_t M = job.M;
  for (int64_t t0 = 0; t0 < M; t0 += kBandTile) {
    __syncthreads();
    const int64_t zlo = t0 + bias[b] + 1 - (lb - 1);
    for (int i = threadIdx.x; i < kBandTile + 2 + lb - 1; i += 256) {
      const int64_t j = zlo + i + pad;
      zt[i] = (j >= 0 && j < M + 2 * pad) ? job.z[j] : 0.0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBandTile + 2; i += 256) {
      double acc = 0.0;
      if (t0 + i < M) {
        const double* zp = zt + i + (lb - 1);
        if (FMA) {
          double a0 = 0.0, a1 = 0.0;  // two chains hide the FP64 FMA latency
          int k = 0;
          for (; k + 1 < lb; k += 2) {
            a0 = fma(taps[k], zp[-k], a0);
            a1 = fma(taps[k + 1], zp[-k - 1], a1);
          }
          if (k < lb) a0 = fma(taps[k], zp[-k], a0);
          acc = a0 + a1;
        } else {
          for (int k = 0; k < lb; ++k) acc += taps[k] * zp[-k];
        }
      }
      sig[i] = acc;
    }
    __syncthreads();
    emit_crossings(sig, t0, M, kBandTile, job.edges, job.cap, base_cnt, scan_scratch, flags);
  }
  if (threadIdx.x < 4) job.counts[threadIdx.x] = base_cnt[threadIdx.x];
}

inline int launch_band_events(wh_ctx* ctx, hipStream_t st, const BandJob* d_jobs, int nb, int n_utt, int pad,
                              const double* d_taps, const int32_t* d_tap_off, const int32_t* d_tap_len,
                              const int32_t* d_bias, int max_lb, bool use_fma, int32_t* d_flag) {
  const size_t lds = sizeof(double) * (((max_lb + 1) & ~1) + ((kBandTile + 2 + max_lb + 1) & ~1) + kBandTile + 2) + 64;
  if (use_fma) {
    if (int rc = allow_lds(&band_events_kernel<true>, lds)) return rc;
    { KernelTimer _kt(ctx, st, "band_events_kernel"); hipLaunchKernelGGL(band_events_kernel<true>, dim3(nb, n_utt), dim3(256), lds, st, d_jobs, pad, d_taps, d_tap_off, d_tap_len, d_bias, nb, d_flag); }
  } else {
    if (int rc = allow_lds(&band_events_kernel<false>, lds)) return rc;
    { KernelTimer _kt(ctx, st, "band_events_kernel"); hipLaunchKernelGGL(band_events_kernel<false>, dim3(nb, n_utt), dim3(256), lds, st, d_jobs, pad, d_taps, d_tap_off, d_tap_len, d_bias, nb, d_flag); }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("band_events_kernel", e);
  return 0;
}

}  // namespace wh
