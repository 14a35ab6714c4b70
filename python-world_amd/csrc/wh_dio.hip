// DIO F0 estimator, batched over utterances.  Replaces dio() of the reference (world/dio.py:10-55).
//
// The reference filters by whole-utterance FFT products; here every filter is a short direct FIR
// (161-tap low-cut staged per 256 outputs, <=80-tap Nuttall low-pass per band) evaluated from LDS tiles, with the same
// circular-convolution indexing the reference's zero-padded FFT implies, so no utterance-length
// FFT is needed and the work is tile-parallel:
//   iir_fwd / iir_bwd : zero-phase 3-pole decimation low-pass (dio.py:359-476).  The serial
//                       recurrence is cut into chunks; each lane warms its state up over W samples
//                       before its chunk (|pole|^W < 1e-20), so chunks run in parallel.
//   lowcut_kernel     : Hann-derived low-cut FIR (dio.py:74-88)
//   band_events_kernel: per (utterance, band, segment): low-pass FIR from LDS, four zero-crossing trains,
//                       ordered stream compaction by packed block scans into segment-private lists
//                       (dio.py:128-140,190-204); band_concat_kernel joins the segments
//   cand_kernel       : per (frame, band): binary search + linear inter/extrapolation of the four
//                       interval-F0 trains, mean / sample-std, range masks, stability (dio.py:92-185)
//   sort_kernel       : per frame insertion sort by stability (dio.py:113-124)
//   contour_kernel    : per utterance serial 4-step contour fix (dio.py:216-326)
#include <math.h>

#include "wh_host.h"
#include "wh_device.h"
#include "wh_events.h"
#include "wh_bands.h"

namespace {

struct DioUtt {
  int64_t x_off, n;       // waveform
  int64_t tmp_off;        // pass-1 output, n+18 doubles
  int64_t y_off, ylen;    // decimated signal
  int64_t z_off;          // low-cut filtered, ylen + 2*pad doubles, index 0 <-> m = -pad
  int64_t e_off, cap;     // edge lists [nb][4][cap]
  int64_t f_off, nf;      // frames
  int64_t nbeg;           // first picked sample (may be negative, dio.py:470)
  int64_t fftmod;         // length of the reference's zero-padded FFT (circular indexing)
};

struct IirCoef {
  double a0, a1, a2, b0, b1;
};

// (a0,a1,a2,b0,b1) literals of the reference's decimation low-pass for r = 2..12 (data, world/dio.py:365-436).
const IirCoef kDecimate[13] = {
    {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0},
    {0.041156734567757189, -0.42599112459189636, 0.041037215479961225, 0.16797464681802227, 0.50392394045406674},
    {0.95039378983237421, -0.67429146741526791, 0.15412211621346475, 0.071221945171178636, 0.21366583551353591},
    {1.4499664446880227, -0.98943497080950582, 0.24578252340690215, 0.036710750339322612, 0.11013225101796784},
    {1.7610939654280557, -1.2554914843859768, 0.3237186507788215, 0.021334858522387423, 0.06400457556716227},
    {1.9715352749512141, -1.4686795689225347, 0.3893908434965701, 0.013469181309343825, 0.040407543928031475},
    {2.1225239019534703, -1.6395144861046302, 0.44469707800587366, 0.0090366882681608418, 0.027110064804482525},
    {2.2357462340187593, -1.7780899984041358, 0.49152555365968692, 0.0063522763407111993, 0.019056829022133598},
    {2.3236003491759578, -1.8921545617463598, 0.53148928133729068, 0.0046331164041389372, 0.013899349212416812},
    {2.3936475118069387, -1.9873904075111861, 0.5658879979027055, 0.0034818622251927556, 0.010445586675578267},
    {2.450743295230728, -2.06794904601978, 0.59574774438332101, 0.0026822508007163792, 0.0080467524021491377},
    {2.4981398605924205, -2.1368928194784025, 0.62187513816221485, 0.0021097275904709001, 0.0063291827714127002}};

constexpr int kPad = 9;       // kNFact, dio.py:452
#ifndef WH_IIR_CHUNK
#define WH_IIR_CHUNK 256
#endif
constexpr int kChunk = WH_IIR_CHUNK;  // samples of IIR output per lane (each lane also re-runs `warm` samples before its chunk)

// mirror-padded input of the first pass (dio.py:458-463)
__device__ __forceinline__ double padded(const double* __restrict__ x, int64_t n, int64_t i) {
  if (i < kPad) return 2 * x[0] - x[kPad - i];
  if (i < kPad + n) return x[i - kPad];
  return 2 * x[n - 1] - x[n - 2 - (i - (kPad + n))];
}

#define IIR_STEP(IN)                                            \
  {                                                             \
    const double wt = (IN) + c.a0 * w0 + c.a1 * w1 + c.a2 * w2; \
    yv = c.b0 * wt + c.b1 * w0 + c.b1 * w1 + c.b0 * w2;         \
    w2 = w1;                                                    \
    w1 = w0;                                                    \
    w0 = wt;                                                    \
  }

#ifndef WH_IIR_BLOCK
#define WH_IIR_BLOCK 32  // (16: 0.0865 + 0.0859 ms for the forward + backward pass at config 2; 32: 0.0801 + 0.0792; 64: 0.0787 + 0.0828)
#endif
constexpr int kIirBlock = WH_IIR_BLOCK;  // samples fetched together, a block ahead of the recurrence (wh::serial_run)

__global__ __launch_bounds__(64) void iir_fwd_kernel(const double* __restrict__ x, const DioUtt* __restrict__ meta,
                                                     IirCoef c, int warm, double* __restrict__ tmp) {
  const DioUtt m = meta[blockIdx.y];
  const int64_t len = m.n + 2 * kPad;
  const int64_t chunk = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int64_t s = chunk * kChunk;
  if (s >= len) return;
  const int64_t e = s + kChunk < len ? s + kChunk : len;
  const double* xu = x + m.x_off;
  double* out = tmp + m.tmp_off;
  const int64_t n = m.n;
  double w0 = 0, w1 = 0, w2 = 0, yv = 0;
  wh::serial_run<kIirBlock>(
      s - warm > 0 ? s - warm : 0, e, [&](int64_t i) { return i >= kPad && i + kIirBlock <= kPad + n; },
      [&](int64_t i) { return xu[i - kPad]; }, [&](int64_t i) { return padded(xu, n, i); },
      [&](int64_t i, double v) {
        IIR_STEP(v);
        if (i >= s) out[i] = yv;
      });
}

// Second pass over the time-reversed pass-1 output; only the decimated picks are stored (dio.py:465-476):
// y[k] = pass2[q + 8] for q = nbeg + k*r, q < n + 9, a negative q + 8 wrapping like a Python index.  The lane walks
// natural indices j downwards, so (quotient, remainder) of j - 8 - nbeg by r are carried along instead of divided out
// per sample; the wrapped picks (nbeg + 8 < 0: r > 8) are the same walk shifted by len.
__global__ __launch_bounds__(64) void iir_bwd_kernel(const DioUtt* __restrict__ meta, IirCoef c, int warm, int r,
                                                     const double* __restrict__ tmp, double* __restrict__ y) {
  const DioUtt m = meta[blockIdx.y];
  const int64_t len = m.n + 2 * kPad;
  const int64_t chunk = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int64_t s = chunk * kChunk;
  if (s >= len) return;
  const int64_t e = s + kChunk < len ? s + kChunk : len;
  const double* in = tmp + m.tmp_off;
  double* yo = y + m.y_off;
  const int64_t ylen = m.ylen, qmax = m.n + kPad;
  const bool wraps = m.nbeg + (kPad - 1) < 0;
  auto floordiv = [](int64_t a, int64_t b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
  // walk state at i = s (j = len - 1 - s), for the plain and the wrapped pick
  const int64_t d0 = (len - 1 - s) - (kPad - 1) - m.nbeg;
  int64_t k0 = floordiv(d0, r), k1 = floordiv(d0 - len, r);
  int r0 = (int)(d0 - k0 * r), r1 = (int)(d0 - len - k1 * r);
  double w0 = 0, w1 = 0, w2 = 0, yv = 0;
  wh::serial_run<kIirBlock>(
      s - warm > 0 ? s - warm : 0, e, [&](int64_t) { return true; }, [&](int64_t i) { return in[len - 1 - i]; },
      [&](int64_t i) { return in[len - 1 - i]; },
      [&](int64_t i, double v) {
        IIR_STEP(v);
        if (i >= s) {
          if (r0 == 0 && k0 >= 0 && k0 < ylen && (len - 1 - i) - (kPad - 1) < qmax) yo[k0] = yv;
          if (--r0 < 0) {
            r0 = r - 1;
            --k0;
          }
          if (wraps) {
            if (r1 == 0 && k1 >= 0 && k1 < ylen) yo[k1] = yv;
            if (--r1 < 0) {
              r1 = r - 1;
              --k1;
            }
          }
        }
      });
}

// z[m mod fft] = sum_k h[k] * yext[(m-k) mod fft], stored for m in [-pad, ylen+pad)   (dio.py:74-88)
// One workgroup per 256 outputs: the 256 + 2*half circularly indexed inputs they touch are staged in LDS once
// (one 64-bit modulo per staged sample instead of one per tap), taps in ascending k like the direct sum.
__global__ __launch_bounds__(256) void lowcut_kernel(const DioUtt* __restrict__ meta, const double* __restrict__ y,
                                                     const double* __restrict__ h, int half, int pad,
                                                     double* __restrict__ z) {
  extern __shared__ __attribute__((aligned(16))) double lc_sh[];  // 256 + 2*half
  const DioUtt m = meta[blockIdx.y];
  const int64_t j0 = (int64_t)blockIdx.x * 256;
  if (j0 >= m.ylen + 2 * pad) return;
  const double* yu = y + m.y_off;
  const int64_t lo = j0 - pad - half;  // first staged index (before the modulo)
  for (int i = threadIdx.x; i < 256 + 2 * half; i += 256) {
    int64_t idx = (lo + i) % m.fftmod;
    if (idx < 0) idx += m.fftmod;
    lc_sh[i] = idx < m.ylen ? yu[idx] : 0.0;
  }
  __syncthreads();
  const int64_t j = j0 + threadIdx.x;
  if (j >= m.ylen + 2 * pad) return;
  double acc = 0.0;
  const double* sh = lc_sh + threadIdx.x + 2 * half;  // (mm - k) - lo = tid + half - k + half... k = -half → +2*half
  for (int t = 0; t <= 2 * half; ++t) acc += h[t] * sh[-t];
  z[m.z_off + j] = acc;
}

__global__ __launch_bounds__(256) void cand_kernel(const DioUtt* __restrict__ meta, const double* __restrict__ tp,
                                                   const double* __restrict__ edges, const int32_t* __restrict__ counts,
                                                   const double* __restrict__ band_f0, int nb, double fs_d,
                                                   double f0_floor, double f0_ceil, double* __restrict__ raw,
                                                   double* __restrict__ stab) {
  const DioUtt m = meta[blockIdx.z];
  const int b = blockIdx.y;
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= m.nf) return;
  const double t = tp[m.f_off + f];
  const int32_t* cnt = counts + ((int64_t)blockIdx.z * nb + b) * 4;
  const double* eb = edges + m.e_off + (int64_t)b * 4 * m.cap;
  double cand, dev;
  wh::interp_four_trains(eb, m.cap, cnt, fs_d, t, true, &cand, &dev);
  const double bf = band_f0[b];
  if (cand > bf || cand < bf / 2 || cand > f0_ceil || cand < f0_floor) cand = 0.0;  // dio.py:146-149
  if (cand == 0.0) dev = 100000.0;
  const int64_t o = m.f_off * nb + (int64_t)b * m.nf + f;
  raw[o] = cand;
  stab[o] = exp(-(dev / fmax(cand, 0.0000001)));
}

constexpr int kMaxBands = 32;
#ifndef WH_DIO_BAND_SEGS
#define WH_DIO_BAND_SEGS 4
#endif
constexpr int kBandSegsMin = WH_DIO_BAND_SEGS;  // workgroups per (band, utterance) in the event extraction, at least
constexpr int kBandSegsMax = 32;
constexpr int kBandTilesPerSeg = 10;  // long signals: more segments, about this many 1024-sample tiles each

__global__ __launch_bounds__(256) void sort_kernel(const DioUtt* __restrict__ meta, int nb, const double* __restrict__ raw,
                                                   const double* __restrict__ stab, double* __restrict__ sorted,
                                                   double* __restrict__ sorted_copy) {
  const DioUtt m = meta[blockIdx.y];
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= m.nf) return;
  double c[kMaxBands], s[kMaxBands];
  for (int b = 0; b < nb; ++b) {
    const int64_t o = m.f_off * nb + (int64_t)b * m.nf + f;
    const double cv = raw[o], sv = stab[o];
    int p = b;  // stable insertion, descending stability (argsort(-stability), dio.py:118)
    while (p > 0 && s[p - 1] < sv) {
      s[p] = s[p - 1];
      c[p] = c[p - 1];
      --p;
    }
    s[p] = sv;
    c[p] = cv;
  }
  for (int b = 0; b < nb; ++b) {
    const int64_t o = m.f_off * nb + (int64_t)b * m.nf + f;
    sorted[o] = c[b];
    if (sorted_copy) sorted_copy[o] = c[b];
  }
}

// float("%.6f" % v): correctly rounded 6-decimal value of the exact binary v (dio.py:243)
__device__ __forceinline__ double round6(double v) {
  const double p = v * 1e6;
  const double err = fma(v, 1e6, -p);  // exact residual of the product
  double r0 = floor(p);
  double frac = (p - r0) + err;
  if (frac >= 1.0) {
    r0 += 1.0;
    frac -= 1.0;
  } else if (frac < 0.0) {
    r0 -= 1.0;
    frac += 1.0;
  }
  if (frac > 0.5 || (frac == 0.5 && fmod(r0, 2.0) != 0.0)) r0 += 1.0;
  return r0 / 1e6;
}

__device__ __forceinline__ double select_best(double cur, double past, const double* __restrict__ cands, int nb,
                                              int64_t stride, double allowed) {
  const double ref = (cur * 3 - past) / 2;
  double best = cands[0];
  double err = fabs(ref - best);
  for (int i = 1; i < nb; ++i) {
    const double c = cands[i * stride];
    const double e = fabs(ref - c);
    if (e < err) {
      err = e;
      best = c;
    }
  }
  if (fabs(1 - best / (ref + 2.220446049250313e-16)) > allowed) best = 0.0;
  return best;
}

// One workgroup per utterance: dio.py:216-326.  Element-wise steps (end zeroing, 6-decimal jump test,
// erosion, copies) run on all 256 threads, the change-point list is built by ordered ballot compaction,
// and only the data-dependent forward/backward extensions (a few frames per voiced section) are walked
// by one lane.
// The walk is a chain of dependent reads (seven candidates and two contour values per frame stepped over): when the
// utterance's candidate rows and the contour fit the CU's LDS (lds_cap doubles: 2001 frames x 7 bands = 128 KB) they are
// staged there first and the walk runs at LDS latency instead of L2 latency (185 -> 60 us at 64 x 10 s).
__global__ __launch_bounds__(256) void contour_kernel(const DioUtt* __restrict__ meta, int n_utt, int nb,
                                                      double frame_period, double f0_floor, double allowed,
                                                      double* __restrict__ cands_all, double* __restrict__ work,
                                                      double* __restrict__ f0_out, double* __restrict__ vuv_out,
                                                      int64_t lds_cap) {
  extern __shared__ __attribute__((aligned(16))) double ct_sh[];  // [nb + 1][n] when it fits
  __shared__ int sh[8];
  __shared__ int sh_first;
  const int u = blockIdx.x;
  const DioUtt m = meta[u];
  const int64_t n = m.nf;
  const int tid = threadIdx.x;
  double* cands = cands_all + m.f_off * nb;  // [nb][n]
  double* s1 = work + m.f_off * 5 + 8 * u;   // 3 rows + a 2n+8 row for the boundary list
  double* s2 = s1 + n;
  double* s3 = s2 + n;
  double* bl = s3 + n;                       // boundary list (as doubles)
  double* f0 = f0_out + m.f_off;
  double* vuv = vuv_out + m.f_off;
  const int64_t vrm = (int64_t)(1 / (frame_period / 1000) / f0_floor + 0.5) * 2 + 1;
  if (n < 2 * vrm + 2) {  // too short for the reference's slicing to leave anything voiced
    for (int64_t i = tid; i < n; i += 256) {
      f0[i] = 0.0;
      vuv[i] = 0.0;
    }
    return;
  }
  double* base = cands;  // row 0, mutated like the reference's view (Q6)
  for (int64_t i = tid; i < vrm; i += 256) {
    base[i] = 0.0;
    base[n - vrm + i] = 0.0;
  }
  __threadfence_block();
  __syncthreads();
  // step 1: jumps between 6-decimal rounded neighbours
  for (int64_t i = tid; i < n; i += 256) {
    double v = base[i];
    if (i >= vrm - 1) {
      const double cur = round6(v), prev = round6(base[i - 1]);
      if (fabs((cur - prev) / (0.000001 + cur)) > allowed) v = 0.0;
    }
    s1[i] = v;
  }
  __threadfence_block();
  __syncthreads();
  // step 2: erode by hw frames on both sides
  const int64_t hw = (vrm - 1) / 2;
  for (int64_t i = tid; i < n; i += 256) {
    double v = s1[i];
    if (i >= hw && i < n - hw) {
      for (int64_t j = -hw; j <= hw; ++j)
        if (s1[i + j] == 0.0) {
          v = 0.0;
          break;
        }
    }
    s2[i] = v;
    s3[i] = v;
  }
  __threadfence_block();
  __syncthreads();
  // boundary list of s2: 0, every i with vuv[i] != vuv[i+1], n-2  (dio.py:318)
  int nbl = 1;
  if (tid == 0) bl[0] = 0.0;
  for (int64_t t0 = 0; t0 < n - 1; t0 += 256) {
    const int64_t i = t0 + tid;
    const bool chg = i < n - 1 && ((s2[i] != 0.0) != (s2[i + 1] != 0.0));
    const unsigned long long mk = __ballot(chg);
    const int w = tid >> 6, lane = tid & 63;
    __syncthreads();
    if (lane == 0) sh[w] = __popcll(mk);
    __syncthreads();
    int off = nbl, tot = 0;
    for (int k = 0; k < 4; ++k) {
      if (k < w) off += sh[k];
      tot += sh[k];
    }
    if (chg) bl[off + __popcll(mk & (lane == 0 ? 0ull : (~0ull >> (64 - lane))))] = (double)i;
    nbl += tot;
  }
  if (tid == 0) {
    bl[nbl] = (double)(n - 2);
    const int64_t bl1 = (int64_t)bl[1];
    const int d1 = (int)(s2[bl1 + 1] != 0.0) - (int)(s2[bl1] != 0.0);
    sh_first = (int)ceil(-0.5 * d1);  // first_section
  }
  ++nbl;
  __threadfence_block();
  __syncthreads();
  const int first = sh_first;
  const int64_t nsec = (int64_t)floor((double)(nbl - (1 - first)) / 2);
  auto sec_start = [&](int64_t i) { return 1 + (int64_t)bl[(i - 1) * 2 + 1 + (1 - first) + 1]; };
  auto sec_end = [&](int64_t i) { return (int64_t)bl[i * 2 + (1 - first) + 1]; };
  if ((int64_t)(nb + 1) * n <= lds_cap) {  // uniform: generic pointers into LDS from here on
    for (int64_t i = tid; i < (int64_t)nb * n; i += 256) ct_sh[i] = cands[i];
    for (int64_t i = tid; i < n; i += 256) ct_sh[(int64_t)nb * n + i] = s3[i];
    cands = ct_sh;
    s3 = ct_sh + (int64_t)nb * n;
    __syncthreads();
  }
  if (tid == 0) {
    // step 3 forward extension
    for (int64_t i = 0; i < nsec; ++i) {
      const int64_t limit = (i == nsec - 1) ? n - 1 : sec_start(i + 1) + 1;
      for (int64_t j = sec_end(i); j < limit; ++j) {
        s3[j + 1] = select_best(s3[j], s3[j - 1], cands + (j + 1), nb, n, allowed);
        if (s3[j + 1] == 0.0) break;
      }
    }
    // step 4 backward extension, in place (the reference copies step 3 first; nothing else reads it)
    for (int64_t i = nsec - 1; i >= 0; --i) {
      const int64_t limit = (i == 0) ? 1 : sec_end(i - 1);
      for (int64_t j = sec_start(i); j >= limit; --j) {
        s3[j - 1] = select_best(s3[j], s3[j + 1], cands + (j - 1), nb, n, allowed);
        if (s3[j - 1] == 0.0) break;
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int64_t i = tid; i < n; i += 256) {
    const double v = s3[i];
    f0[i] = v;
    vuv[i] = v != 0.0 ? 1.0 : 0.0;
  }
}

double pole_radius(const IirCoef& c) {
  // roots of z^3 - a0 z^2 - a1 z - a2 by Durand-Kerner (tiny, host side)
  double re[3] = {0.4, -0.2, 0.3}, im[3] = {0.9, 0.5, -0.7};
  for (int it = 0; it < 200; ++it) {
    for (int i = 0; i < 3; ++i) {
      // p(z)
      const double zr = re[i], zi = im[i];
      double pr = zr - c.a0, pi = zi;              // z - a0
      double tr = pr * zr - pi * zi - c.a1, ti = pr * zi + pi * zr;  // (z-a0)z - a1
      pr = tr * zr - ti * zi - c.a2;
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
  }
  double r = 0;
  for (int i = 0; i < 3; ++i) r = fmax(r, hypot(re[i], im[i]));
  return r;
}

}  // namespace

extern "C" int wh_dio(wh_ctx* ctx, void* stream, const wh_batch* b, const double* x, const double* tp, double fs,
                      double f0_floor, double f0_ceil, double target_fs, double frame_period_ms, double allowed_range,
                      int n_bands, const double* h_band_f0, const int32_t* h_band_bias, const int32_t* h_band_len,
                      const double* h_band_taps, const double* h_lowcut, int lowcut_half, double* f0_out,
                      double* vuv_out, double* cand_out, double* raw_out) {
  if (!ctx || !b || !x || !tp || !h_band_f0 || !h_band_bias || !h_band_len || !h_band_taps || !h_lowcut || !f0_out ||
      !vuv_out)
    return wh::fail_msg("wh_dio", "null argument");
  WH_ENTER(ctx);
  if (n_bands < 1 || n_bands > kMaxBands) return wh::fail_msg("wh_dio", "n_bands must be in [1, 32]");
  hipStream_t st = (hipStream_t)stream;
  const int B = b->n_utt;
  const int r = (int)(fs / target_fs);
  if (r < 1) return wh::fail_msg("wh_dio", "fs below target_fs");
  const IirCoef coef = (r >= 2 && r <= 12) ? kDecimate[r] : kDecimate[0];  // unknown ratio → all-zero filter (Q4)
  const double fs_d = target_fs;                                           // the true ratio is ignored (Q4)
  // The band filters run on the register-tiled FIR of wh_bands.h, which wants odd tap counts (16-byte aligned input
  // pairs): an even-length filter gets one trailing zero tap (a + 0*z == a, the sums are unchanged).
#ifndef WH_DIO_BAND_TILED
#define WH_DIO_BAND_TILED 1
#endif
  int max_lb = 0, taps_total = 0;
  std::vector<int32_t> tap_off(n_bands), tap_len(n_bands);
  for (int i = 0; i < n_bands; ++i) {
    if (h_band_len[i] < 1) return wh::fail_msg("wh_dio", "empty band filter");
    tap_len[i] = WH_DIO_BAND_TILED ? (h_band_len[i] | 1) : h_band_len[i];
    tap_off[i] = taps_total;
    taps_total += tap_len[i];
    if (tap_len[i] > max_lb) max_lb = tap_len[i];
  }
  const int pad = max_lb + 2;
  const int hfl_pad = (int)(fs_d / f0_floor / 2 + 0.5) * 4;
  // ---- per-utterance metadata + workspace carve -------------------------------------------------
  std::vector<DioUtt> meta(B);
  int64_t tmp_tot = 0, y_tot = 0, z_tot = 0, e_tot = 0, max_len = 0, max_ylen = 0, max_nf = 0;
  for (int u = 0; u < B; ++u) {
    DioUtt& m = meta[u];
    m.x_off = b->h_x_off[u];
    m.n = b->h_x_off[u + 1] - b->h_x_off[u];
    if (m.n < 2 * kPad + 2) return wh::fail_msg("wh_dio", "utterance shorter than 20 samples");
    m.f_off = b->h_frame_off[u];
    m.nf = b->h_frame_off[u + 1] - b->h_frame_off[u];
    const double nout = ceil((double)m.n / r + 1);
    m.nbeg = (int64_t)(r - r * nout + m.n);
    m.ylen = (m.n + kPad - m.nbeg + r - 1) / r;
    m.tmp_off = tmp_tot;
    tmp_tot += m.n + 2 * kPad;
    m.y_off = y_tot;
    y_tot += m.ylen;
    m.z_off = z_tot;
    z_tot += m.ylen + 2 * pad;
    m.cap = m.ylen / 2 + 2;
    m.e_off = e_tot;
    e_tot += (int64_t)n_bands * 4 * m.cap;
    m.fftmod = (int64_t)llround(pow(2.0, ceil(log2((double)(m.ylen + hfl_pad)))));
    max_len = std::max(max_len, m.n + 2 * kPad);
    max_ylen = std::max(max_ylen, m.ylen);
    max_nf = std::max(max_nf, m.nf);
  }
  const int64_t F = b->total_frames;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t off = 0;
  const size_t o_tmp = off; off += al(sizeof(double) * tmp_tot);
  const size_t o_y = off; off += al(sizeof(double) * y_tot);
  const size_t o_z = off; off += al(sizeof(double) * z_tot);
  const size_t o_e = off; off += al(sizeof(double) * e_tot);
  const size_t o_cnt = off; off += al(sizeof(int32_t) * (size_t)B * n_bands * 4);
  // segment-private event lists: 7 bands x B utterances alone cannot fill 256 CUs with their serial tile loops
  std::vector<int64_t> seg_cap(B), seg_off(B);
  int64_t se_tot = 0;
  int kBandSegs = kBandSegsMin;
  {
    const int64_t max_tiles = (max_ylen + wh::kBandTile - 1) / wh::kBandTile;
    const int64_t want = (max_tiles + kBandTilesPerSeg - 1) / kBandTilesPerSeg;
    if (want > kBandSegs) kBandSegs = (int)(want < kBandSegsMax ? want : kBandSegsMax);
  }
  for (int u = 0; u < B; ++u) {
    const int64_t tiles = (meta[u].ylen + wh::kBandTile - 1) / wh::kBandTile;
    seg_cap[u] = ((tiles + kBandSegs - 1) / kBandSegs) * (wh::kBandTile / 2) + 2;
    seg_off[u] = se_tot;
    se_tot += (int64_t)n_bands * kBandSegs * 4 * seg_cap[u];
  }
  const size_t o_se = off; off += al(sizeof(double) * se_tot);
  const size_t o_scnt = off; off += al(sizeof(int32_t) * (size_t)B * n_bands * kBandSegs * 4);
  const size_t o_raw = off; off += al(sizeof(double) * F * n_bands);
  const size_t o_stab = off; off += al(sizeof(double) * F * n_bands);
  const size_t o_sorted = off; off += al(sizeof(double) * F * n_bands);
  const size_t o_work = off; off += al(sizeof(double) * (F * 5 + 8 * B));
  if (int rc = wh::ws_reserve(ctx, off)) return rc;
  char* ws = reinterpret_cast<char*>(ctx->ws);
  DioUtt* d_meta = nullptr;
  double* d_tmp = reinterpret_cast<double*>(ws + o_tmp);
  double* d_y = reinterpret_cast<double*>(ws + o_y);
  double* d_z = reinterpret_cast<double*>(ws + o_z);
  double* d_e = reinterpret_cast<double*>(ws + o_e);
  int32_t* d_cnt = reinterpret_cast<int32_t*>(ws + o_cnt);
  double* d_se = reinterpret_cast<double*>(ws + o_se);
  int32_t* d_scnt = reinterpret_cast<int32_t*>(ws + o_scnt);
  double* d_raw = raw_out ? raw_out : reinterpret_cast<double*>(ws + o_raw);
  double* d_stab = reinterpret_cast<double*>(ws + o_stab);
  double* d_sorted = reinterpret_cast<double*>(ws + o_sorted);
  double* d_work = reinterpret_cast<double*>(ws + o_work);
  double* d_taps = nullptr;
  double* d_lc = nullptr;
  double* d_bf = nullptr;
  int32_t* d_ti = nullptr;
  wh::BandJob* d_jobs = nullptr;
  std::vector<int32_t> ti(n_bands * 3);
  for (int i = 0; i < n_bands; ++i) {
    ti[i] = tap_off[i];
    ti[n_bands + i] = tap_len[i];
    ti[2 * n_bands + i] = h_band_bias[i];
  }
  // per-call tables live in persistent device buffers: re-uploaded (synchronously) only when they change
  {
    std::vector<double> taps(taps_total, 0.0), lc(h_lowcut, h_lowcut + 2 * lowcut_half + 1),
        bf(h_band_f0, h_band_f0 + n_bands);
    for (int i = 0, src = 0; i < n_bands; src += h_band_len[i], ++i)
      std::copy(h_band_taps + src, h_band_taps + src + h_band_len[i], taps.begin() + tap_off[i]);
    if (int rc = wh::persistent_upload(ctx, st, "dio.meta", meta, &d_meta)) return rc;
    if (int rc = wh::persistent_upload(ctx, st, "dio.taps", taps, &d_taps)) return rc;
    if (int rc = wh::persistent_upload(ctx, st, "dio.lowcut", lc, &d_lc)) return rc;
    if (int rc = wh::persistent_upload(ctx, st, "dio.band_f0", bf, &d_bf)) return rc;
    if (int rc = wh::persistent_upload(ctx, st, "dio.tapinfo", ti, &d_ti)) return rc;
  }

  // ---- decimation ---------------------------------------------------------------------------------
  const double rad = pole_radius(coef);
  int warm = 64;
  if (rad > 0 && rad < 1) warm = (int)ceil(-46.0 / log(rad));
  warm = ((warm + 63) / 64) * 64;
  if (!(rad < 0.9999)) warm = 1 << 30;  // unstable / unknown: fall back to a fully serial pass per lane
  const int chunks = (int)((max_len + kChunk - 1) / kChunk);
  dim3 gi((chunks + 63) / 64, B);
  { wh::KernelTimer _kt(ctx, st, "iir_fwd_kernel"); hipLaunchKernelGGL(iir_fwd_kernel, gi, dim3(64), 0, st, x, d_meta, coef, warm, d_tmp); }
  WH_LAUNCH_CHECK("iir_fwd_kernel");
  { wh::KernelTimer _kt(ctx, st, "iir_bwd_kernel"); hipLaunchKernelGGL(iir_bwd_kernel, gi, dim3(64), 0, st, d_meta, coef, warm, r, d_tmp, d_y); }
  WH_LAUNCH_CHECK("iir_bwd_kernel");
  // ---- low-cut + band events ----------------------------------------------------------------------
  { wh::KernelTimer _kt(ctx, st, "lowcut_kernel"); hipLaunchKernelGGL(lowcut_kernel, dim3((unsigned)((max_ylen + 2 * pad + 255) / 256), B), dim3(256), sizeof(double) * (256 + 2 * lowcut_half), st, d_meta, d_y,
                     d_lc, lowcut_half, pad, d_z); }
  WH_LAUNCH_CHECK("lowcut_kernel");
  {
    std::vector<wh::BandJob> jobs((size_t)B * n_bands);
    for (int u = 0; u < B; ++u)
      for (int i = 0; i < n_bands; ++i) {
        wh::BandJob& j = jobs[(size_t)u * n_bands + i];
        j.z = d_z + meta[u].z_off;
        j.M = meta[u].ylen;
        j.edges = d_e + meta[u].e_off + (int64_t)i * 4 * meta[u].cap;
        j.cap = meta[u].cap;
        j.counts = d_cnt + ((int64_t)u * n_bands + i) * 4;
        j.seg_edges = d_se + seg_off[u] + (int64_t)i * kBandSegs * 4 * seg_cap[u];
        j.seg_counts = d_scnt + ((int64_t)u * n_bands + i) * kBandSegs * 4;
        j.seg_cap = seg_cap[u];
      }
    if (int rc = wh::persistent_upload(ctx, st, "dio.jobs", jobs, &d_jobs)) return rc;
  }
  if (int rc = wh::launch_band_events(ctx, st, d_jobs, n_bands, B, pad, d_taps, d_ti, d_ti + n_bands, d_ti + 2 * n_bands,
                                      max_lb, WH_DIO_BAND_TILED != 0, ctx->d_flags + WH_FLAG_EVENT_OVERFLOW, kBandSegs))
    return rc;
  // ---- candidates, sort, contour ------------------------------------------------------------------
  { wh::KernelTimer _kt(ctx, st, "cand_kernel"); hipLaunchKernelGGL(cand_kernel, dim3((unsigned)((max_nf + 255) / 256), n_bands, B), dim3(256), 0, st, d_meta, tp, d_e,
                     d_cnt, d_bf, n_bands, fs_d, f0_floor, f0_ceil, d_raw, d_stab); }
  WH_LAUNCH_CHECK("cand_kernel");
  { wh::KernelTimer _kt(ctx, st, "sort_kernel"); hipLaunchKernelGGL(sort_kernel, dim3((unsigned)((max_nf + 255) / 256), B), dim3(256), 0, st, d_meta, n_bands, d_raw,
                     d_stab, d_sorted, cand_out); }
  WH_LAUNCH_CHECK("sort_kernel");
  {
    // candidate rows + contour in LDS when the longest utterance fits (see contour_kernel)
    const size_t want = sizeof(double) * (size_t)(n_bands + 1) * (size_t)max_nf;
    const size_t lds = want <= 150 * 1024 ? want : 0;
    if (int rc = wh::allow_lds(&contour_kernel, lds)) return rc;
    wh::KernelTimer _kt(ctx, st, "contour_kernel");
    hipLaunchKernelGGL(contour_kernel, dim3(B), dim3(256), lds, st, d_meta, B, n_bands, frame_period_ms, f0_floor,
                       allowed_range, d_sorted, d_work, f0_out, vuv_out, (int64_t)(lds / sizeof(double)));
  }
  WH_LAUNCH_CHECK("contour_kernel");
  return 0;
}
