// Requiem seed signals generated on the device.  Replaces the host producer get_seeds_signals()
// (world/get_seeds_signals.py:8-73) for the batched decode: the band pulses are deterministic and are evaluated
// exactly; the modified velvet noise (get_seeds_signals.py:40-73: short segments of three lengths chosen at random,
// one +-2 impulse per 4-sample cell at a random offset, signs balanced per segment and shuffled) draws from a
// counter-based Philox-4x32-10 stream instead of Python's `random` / NumPy's global generator — the same
// construction, statistically equivalent, not the same samples (tests/test_hip_seeds.py documents what is compared).
// The noise seeds are the circular convolution of that velvet noise with each band pulse, summed directly (the
// reference multiplies two FFTs; equal up to rounding).
#include <math.h>

#include "wh_device.h"
#include "wh_host.h"

namespace {

__device__ __forceinline__ void philox4(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t* out) {
  uint32_t c2 = 0x9E3779B9u, c3 = 0x243F6A88u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

// Segment layout: lengths drawn uniformly from the three short periods until the noise is covered
// (get_seeds_signals.py:45-52).  Inherently sequential, ~100-400 steps: one thread.
__global__ void velvet_layout_kernel(int n, int len0, int len1, int len2, uint64_t seed, int max_seg,
                                     int32_t* __restrict__ seg_start, int32_t* __restrict__ seg_len,
                                     int32_t* __restrict__ n_seg) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const int lens[3] = {len0, len1, len2};
  int index = 0, count = 0;
  while (count < max_seg) {
    uint32_t r[4];
    philox4((uint32_t)count, 0x5E65u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const int ln = lens[r[0] % 3u];
    seg_start[count] = index;
    seg_len[count] = ln;
    ++count;
    index += ln;
    if (index >= n - 1) break;
  }
  *n_seg = count;
}

// One workgroup per segment: r = len/4 cells (<= 256), one impulse per cell at a random offset in the cell; the signs
// are the balanced pool (+2 for the first r/2 ranks, -2 for the rest, get_seeds_signals.py:60-63) in a random order
// (the rank of a random key per cell).
__global__ __launch_bounds__(256) void velvet_fill_kernel(const int32_t* __restrict__ seg_start,
                                                          const int32_t* __restrict__ seg_len,
                                                          const int32_t* __restrict__ n_seg, int n, uint64_t seed,
                                                          double* __restrict__ velvet) {
  __shared__ unsigned long long keys[256];
  const int s = blockIdx.x;
  if (s >= *n_seg) return;
  const int t = threadIdx.x;
  const int len = seg_len[s], start = seg_start[s];
  const int cells = len / 4;  // int(N // td + 0.5)
  uint32_t r[4];
  philox4((uint32_t)s, 0x1000u + (uint32_t)t, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  const unsigned long long key = ((unsigned long long)r[0] << 32) | r[1];
  keys[t] = key;
  __syncthreads();
  if (t < cells) {
    int rank = 0;
    for (int o = 0; o < cells; ++o) {
      const unsigned long long other = keys[o];
      rank += (other < key || (other == key && o < t)) ? 1 : 0;
    }
    const int pos = start + 4 * t + (int)(r[2] & 3u);
    if (pos < n) velvet[pos] = rank < cells / 2 ? 2.0 : -2.0;
  }
}

// Band pulses (get_seeds_signals.py:27-35): raised-cosine band shapes on the half spectrum, inverse real DFT as a
// direct cosine sum, fftshift.  One workgroup per band.  (Band 0's DC correction follows in seed_band0_dc_kernel,
// after the noise seeds have been formed from the uncorrected pulse — the reference's order, :36-38.)
__global__ __launch_bounds__(256) void seed_pulse_kernel(double fs, int fft_size, int nb, double* __restrict__ pulse) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* shape = reinterpret_cast<double*>(smem);  // fft_size/2 + 1
  const int b = blockIdx.x;
  const int half = fft_size / 2;
  const double step = 3000.0;
  for (int k = threadIdx.x; k <= half; k += 256) {
    const double w = (double)k * fs / fft_size;
    double v = 0.5 + 0.5 * cos(((w - (step * b)) / (step * 2)) * 2 * M_PI);
    if (w > step * (b + 1)) v = 0.0;
    if (w < step * (b - 1)) v = 0.0;
    if (b == nb - 1 && w > step * b) v = 1.0;  // the top band is a high-pass
    shape[k] = v;
  }
  __syncthreads();
  for (int n = threadIdx.x; n < fft_size; n += 256) {
    const int m = (n + half) & (fft_size - 1);  // fftshift
    double acc = shape[0] + ((m & 1) ? -shape[half] : shape[half]);
    for (int k = 1; k < half; ++k) acc += 2.0 * shape[k] * cospi(2.0 * (double)((long long)k * m % fft_size) / fft_size);
    pulse[(long long)n * nb + b] = acc / fft_size;
  }
}

// noise[:, b] = ifft(fft(velvet) * fft(pulse[:, b], N)).real  ==  circular convolution, summed directly.
__global__ __launch_bounds__(256) void seed_noise_kernel(const double* __restrict__ velvet, int n,
                                                         const double* __restrict__ pulse, int fft_size, int nb,
                                                         double* __restrict__ noise) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= n) return;
  double acc = 0.0;
  for (int j = 0; j < fft_size; ++j) {
    const double v = velvet[(i - j) & (n - 1)];
    if (v != 0.0) acc += v * pulse[(long long)j * nb + b];
  }
  noise[(long long)i * nb + b] = acc;
}

}  // namespace

// pulse[:, 0] -= mean(pulse[:, 0]) * h / mean(h), h = hanning(fft_size + 2)[1:-1]  (get_seeds_signals.py:36-37)
namespace {
__global__ __launch_bounds__(256) void seed_band0_dc_kernel(int fft_size, int nb, double* __restrict__ pulse) {
  __shared__ double scratch[16];
  double mean_p = 0.0, mean_h = 0.0;
  for (int n = threadIdx.x; n < fft_size; n += 256) {
    mean_p += pulse[(long long)n * nb];
    mean_h += 0.5 - 0.5 * cospi(2.0 * (double)(n + 1) / (double)(fft_size + 1));
  }
  wh::block_sum2<256>(mean_p, mean_h, scratch);
  mean_p /= fft_size;
  mean_h /= fft_size;
  for (int n = threadIdx.x; n < fft_size; n += 256) {
    const double h = 0.5 - 0.5 * cospi(2.0 * (double)(n + 1) / (double)(fft_size + 1));
    pulse[(long long)n * nb] = pulse[(long long)n * nb] - mean_p * h / mean_h;
  }
}
}  // namespace

extern "C" int wh_requiem_seeds(wh_ctx* ctx, void* stream, double fs, int fft_size, int64_t noise_length, int n_bands,
                                uint64_t seed, double* pulse_seed, double* noise_seed, double* velvet_out) {
  if (!ctx || !pulse_seed || !noise_seed) return wh::fail_msg("wh_requiem_seeds", "null argument");
  WH_ENTER(ctx);
  if (fft_size < 64 || (fft_size & (fft_size - 1)) || noise_length < fft_size || (noise_length & (noise_length - 1)) ||
      noise_length > (1 << 24))
    return wh::fail_msg("wh_requiem_seeds", "fft_size / noise_length must be powers of two, fft_size <= noise_length");
  if (n_bands < 2 || n_bands > 8) return wh::fail_msg("wh_requiem_seeds", "n_bands must be in [2, 8]");
  hipStream_t st = (hipStream_t)stream;
  const int n = (int)noise_length;
  // short periods 8 * "round"(p * fs / 48000): the reference's round helper only offsets by 0.5
  int lens[3];
  const double base[3] = {8, 30, 60};
  for (int i = 0; i < 3; ++i) lens[i] = (int)(8 * (base[i] * fs / 48000 + 0.5));
  if (lens[0] < 4 || lens[2] > 1024) return wh::fail_msg("wh_requiem_seeds", "sampling rate outside the supported range");
  const int max_seg = n / lens[0] + 2;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t off = 0;
  const size_t o_vel = off; off += al(sizeof(double) * n);
  const size_t o_ss = off; off += al(sizeof(int32_t) * max_seg);
  const size_t o_sl = off; off += al(sizeof(int32_t) * max_seg);
  const size_t o_ns = off; off += 256;
  if (int rc = wh::ws_reserve(ctx, off)) return rc;
  char* ws = reinterpret_cast<char*>(ctx->ws);
  double* d_vel = reinterpret_cast<double*>(ws + o_vel);
  int32_t* d_ss = reinterpret_cast<int32_t*>(ws + o_ss);
  int32_t* d_sl = reinterpret_cast<int32_t*>(ws + o_sl);
  int32_t* d_ns = reinterpret_cast<int32_t*>(ws + o_ns);
  WH_CHECK(hipMemsetAsync(d_vel, 0, sizeof(double) * n, st));
  { wh::KernelTimer _kt(ctx, st, "velvet_layout_kernel"); hipLaunchKernelGGL(velvet_layout_kernel, dim3(1), dim3(64), 0, st, n, lens[0], lens[1], lens[2], seed, max_seg, d_ss, d_sl, d_ns); }
  WH_LAUNCH_CHECK("velvet_layout_kernel");
  { wh::KernelTimer _kt(ctx, st, "velvet_fill_kernel"); hipLaunchKernelGGL(velvet_fill_kernel, dim3(max_seg), dim3(256), 0, st, d_ss, d_sl, d_ns, n, seed, d_vel); }
  WH_LAUNCH_CHECK("velvet_fill_kernel");
  const size_t lds = sizeof(double) * (fft_size / 2 + 1);
  { wh::KernelTimer _kt(ctx, st, "seed_pulse_kernel"); hipLaunchKernelGGL(seed_pulse_kernel, dim3(n_bands), dim3(256), lds, st, fs, fft_size, n_bands, pulse_seed); }
  WH_LAUNCH_CHECK("seed_pulse_kernel");
  { wh::KernelTimer _kt(ctx, st, "seed_noise_kernel"); hipLaunchKernelGGL(seed_noise_kernel, dim3((n + 255) / 256, n_bands), dim3(256), 0, st, d_vel, n, pulse_seed, fft_size, n_bands, noise_seed); }
  WH_LAUNCH_CHECK("seed_noise_kernel");
  { wh::KernelTimer _kt(ctx, st, "seed_band0_dc_kernel"); hipLaunchKernelGGL(seed_band0_dc_kernel, dim3(1), dim3(256), 0, st, fft_size, n_bands, pulse_seed); }
  WH_LAUNCH_CHECK("seed_band0_dc_kernel");
  if (velvet_out) WH_CHECK(hipMemcpyAsync(velvet_out, d_vel, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  return 0;
}
