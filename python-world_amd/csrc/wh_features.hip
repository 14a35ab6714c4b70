// Spectral feature heads downstream of encode(): log mel filterbank energies, mel-cepstrum and its inverse, context
// stacking.  Replaces World.encode_lfbank / encode_mcep / decode_mcep / get_context (world/main.py:305-365).
//
// All three heads are one dense product per frame — the only dense contractions in the reference
// (`np.dot(pspec, fb.T)`, world/main.py:320; the irfft / rfft of encode_mcep / decode_mcep keep 12 coefficients, i.e.
// a 513 x 12 cosine matrix with the mel warp folded in) — with an elementwise prologue and epilogue:
//     out[f][n] = epi( sum_k pro(A[f][k], k) * W[k][n] )
//   encode_lfbank : pro = (a * |H(k)|)^2 / nfft (pre-emphasis, power), W = fb^T,           epi = log (0 -> eps)
//   encode_mcep   : pro = log a,                                     W = warp + irfft rows, epi = none
//   decode_mcep   : pro = none,                                      W = rfft rows + warp,  epi = exp
// feature_matmul_kernel runs it on the FP64 matrix cores (v_mfma_f64_16x16x4_f64): a workgroup of four waves takes a
// 128 x 64 output tile, each wave 32 x 64 of it; 16-wide k strips of A (prologue applied on the way in) and of W
// (k-major, zero-padded to multiples of 32 x 16) are staged in double-buffered LDS, fetched a step ahead of the MFMAs.
// The heads are HBM-bound (4104 B in per 32 x 8 B out for the filterbank); SWIPE's products (513-1025 x 326, 326 x ~150
// per window size) are where the matrix cores carry weight.
//
// MFMA operand layout (measured, tools/ubench/mfma_check.hip): A[i][k] in lane 16k+i, B[k][j] in lane 16k+j,
// D[4r + l/16][l%16] in register r of lane l.
#include <math.h>

#include "wh_device.h"
#include "wh_host.h"

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kFeatKC = 32;   // W is zero-padded on the host side to multiples of 32 (k) x 16 (n)
constexpr int kFeatKS = 16;   // k step staged in LDS per pipeline stage
constexpr int kFeatNT = 4;    // 16-column tiles per workgroup: a 128 x 64 output tile
constexpr int kFeatRows = 128;  // rows per workgroup: each of the four waves owns 32 x 64 = 2 x 4 MFMA tiles
constexpr int kFeatAS = kFeatKS + 1;        // row stride of the staged A strip in doubles: 16 rows x 4 k land on distinct banks
constexpr int kFeatBS = 16 * kFeatNT + 16;  // row stride of the staged W strip: k rows 32 banks apart

// One product with prologue / epilogue, tiled for the FP64 matrix cores.  Per 16-wide k step a workgroup stages a
// 128 x 16 strip of A (prologue applied on the way in) and the 16 x 64 strip of W in LDS — both fetched from global
// memory into registers ONE STEP AHEAD, while the 32 MFMAs per wave of the current step run out of the other LDS buffer —
// so the matrix pipe waits neither for HBM (A) nor for L2 (W): the first version read W from L2 in front of every MFMA and
// single-buffered A (5.6 TFLOP/s on SWIPE's products).  Each wave keeps 8 accumulator tiles (64 VGPRs): an A operand
// feeds four MFMAs, a W operand two.  One barrier per step.
template <int PRO, int EPI>
__global__ __launch_bounds__(256) void feature_matmul_kernel(const double* __restrict__ A, long long n_rows, int ka,
                                                             long long lda, const double* __restrict__ P, double pscale,
                                                             const double* __restrict__ W, int kpad, int npad, int nw,
                                                             double* __restrict__ out, long long ldo) {
  __shared__ double As[2][kFeatRows * kFeatAS];
  __shared__ double Bs[2][kFeatKS * kFeatBS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long f0 = (long long)blockIdx.x * kFeatRows;
  const int nt0 = blockIdx.y * kFeatNT;
  const int ntiles = npad / 16;
  const int nt_here = ntiles - nt0 < kFeatNT ? ntiles - nt0 : kFeatNT;  // column tiles this workgroup really has
  double4_t acc[2][kFeatNT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int t = 0; t < kFeatNT; ++t) acc[mt][t] = double4_t{0.0, 0.0, 0.0, 0.0};
  // staging roles: thread t fetches 8 consecutive k of A row t/2 and 4 consecutive columns of W row t/16
  const int ar = threadIdx.x >> 1, ak = (threadIdx.x & 1) * 8;
  const int bk = threadIdx.x >> 4, bc = (threadIdx.x & 15) * 4;
  const long long arow = f0 + ar;
  const bool a_ok = arow < n_rows;
  double areg[8], breg[4];
  auto fetch = [&](int kc) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = kc + ak + i;
      double v = 0.0;
      if (a_ok && k < ka) {
        v = A[arow * lda + k];
        if (PRO == 1) {
          v = v * P[k];
          v = pscale * (v * v);  // 1 / nfft * np.square(spec * |h|)  (main.py:314-316)
        } else if (PRO == 2) {
          v = log(v);
        }
      }
      areg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = 16 * nt0 + bc + i;
      breg[i] = n < npad ? W[(long long)(kc + bk) * npad + n] : 0.0;
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) As[buf][ar * kFeatAS + ak + i] = areg[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) Bs[buf][bk * kFeatBS + bc + i] = breg[i];
  };
  fetch(0);
  stage(0);
  __syncthreads();
  const int steps = kpad / kFeatKS;
  for (int st = 0; st < steps; ++st) {
    const int buf = st & 1;
    if (st + 1 < steps) fetch((st + 1) * kFeatKS);  // in flight under this step's MFMAs
    const double* as = As[buf] + (32 * w + (lane & 15)) * kFeatAS + (lane >> 4);
    const double* bs = Bs[buf] + (lane >> 4) * kFeatBS + (lane & 15);
#pragma unroll
    for (int kk = 0; kk < kFeatKS / 4; ++kk) {
      const double a0 = as[4 * kk], a1 = as[16 * kFeatAS + 4 * kk];
#pragma unroll
      for (int t = 0; t < kFeatNT; ++t) {
        if (t < nt_here) {
          const double b = bs[4 * kk * kFeatBS + 16 * t];
          acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc[1][t], 0, 0, 0);
        }
      }
    }
    if (st + 1 < steps) stage(buf ^ 1);  // the other buffer was last read a step ago, before the barrier below
    __syncthreads();
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int t = 0; t < kFeatNT; ++t) {
      const int n = 16 * (nt0 + t) + (lane & 15);
      if (t < nt_here && n < nw) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long long f = f0 + 32 * w + 16 * mt + 4 * r + (lane >> 4);
          if (f < n_rows) {
            double v = acc[mt][t][r];
            if (EPI == 1) v = log(v == 0.0 ? 2.220446049250313e-16 : v);  // np.where(feat == 0, eps, feat); np.log
            else if (EPI == 2) v = exp(v);
            else if (EPI == 3) v = sqrt(fmax(0.0, v));  // SWIPE' loudness: np.sqrt(np.maximum(0, .)) (swipe.py:42-44)
            out[f * ldo + n] = v;
          }
        }
      }
    }
}

// get_context (main.py:360-365): out[i] = rows i-w .. i+w of X side by side, the first / last row repeated at the ends
__global__ __launch_bounds__(256) void context_kernel(const double* __restrict__ X, long long n, int d, int w,
                                                      double* __restrict__ out) {
  const long long width = (long long)(2 * w + 1) * d;
  const long long i = blockIdx.x;
  for (long long c = threadIdx.x; c < width; c += 256) {
    long long src = i + c / d - w;
    src = src < 0 ? 0 : (src > n - 1 ? n - 1 : src);
    out[i * width + c] = X[src * d + c % d];
  }
}

template <int PRO>
int launch_epi(int epi, dim3 grid, hipStream_t st, const double* A, long long n_rows, int ka, long long lda, const double* P,
               double pscale, const double* W, int kpad, int npad, int nw, double* out, long long ldo) {
  switch (epi) {
    case 0: hipLaunchKernelGGL((feature_matmul_kernel<PRO, 0>), grid, dim3(256), 0, st, A, n_rows, ka, lda, P, pscale, W, kpad, npad, nw, out, ldo); break;
    case 1: hipLaunchKernelGGL((feature_matmul_kernel<PRO, 1>), grid, dim3(256), 0, st, A, n_rows, ka, lda, P, pscale, W, kpad, npad, nw, out, ldo); break;
    case 2: hipLaunchKernelGGL((feature_matmul_kernel<PRO, 2>), grid, dim3(256), 0, st, A, n_rows, ka, lda, P, pscale, W, kpad, npad, nw, out, ldo); break;
    case 3: hipLaunchKernelGGL((feature_matmul_kernel<PRO, 3>), grid, dim3(256), 0, st, A, n_rows, ka, lda, P, pscale, W, kpad, npad, nw, out, ldo); break;
    default: return wh::fail_msg("wh_feature_matmul", "epilogue must be 0 (none), 1 (log), 2 (exp) or 3 (sqrt of the positive part)");
  }
  return 0;
}

}  // namespace

extern "C" int wh_feature_matmul_tagged(wh_ctx* ctx, void* stream, const double* a, int64_t n_rows, int ka, int64_t lda,
                                        int prologue, const double* h_p, double pscale, const double* h_w, int nw,
                                        int epilogue, double* out, int64_t ldo, uint64_t table_tag) {
  if (!ctx || !a || !h_w || !out) return wh::fail_msg("wh_feature_matmul", "null argument");
  WH_ENTER(ctx);
  if (n_rows <= 0) return 0;
  if (ka < 1 || nw < 1 || lda < ka || ldo < nw) return wh::fail_msg("wh_feature_matmul", "bad shape");
  if (prologue == 1 && !h_p) return wh::fail_msg("wh_feature_matmul", "prologue 1 needs the per-column table");
  hipStream_t st = (hipStream_t)stream;
  const int kpad = ((ka + kFeatKC - 1) / kFeatKC) * kFeatKC;
  const int npad = ((nw + 15) / 16) * 16;
  const std::string shape = std::to_string(ka) + "x" + std::to_string(nw);
  double* d_w = nullptr;
  double* d_p = nullptr;
  const std::string slot = table_tag ? "feat.t." + std::to_string(table_tag) + "." + shape
                                     : "feat.w." + std::to_string(prologue) + std::to_string(epilogue) + "." + shape;
  if (table_tag) {  // a tagged table that is already resident: a pointer look-up
    auto it = ctx->persist.find(slot);
    if (it != ctx->persist.end() && it->second.d && it->second.host.size() == (size_t)kpad * npad * sizeof(double))
      d_w = reinterpret_cast<double*>(it->second.d);
  }
  if (!d_w) {
    std::vector<double> wp((size_t)kpad * npad, 0.0);  // zero padding: the surplus k rows / n columns contribute nothing
    for (int k = 0; k < ka; ++k)
      for (int n = 0; n < nw; ++n) wp[(size_t)k * npad + n] = h_w[(size_t)k * nw + n];
    if (int rc = wh::persistent_upload(ctx, st, slot, wp, &d_w)) return rc;
  }
  if (prologue == 1) {
    std::vector<double> pv(h_p, h_p + ka);
    if (int rc = wh::persistent_upload(ctx, st, "feat.p." + std::to_string(ka), pv, &d_p)) return rc;
  }
  const dim3 grid((unsigned)((n_rows + kFeatRows - 1) / kFeatRows), (unsigned)((npad / 16 + kFeatNT - 1) / kFeatNT));
  int rc;
  {
    wh::KernelTimer _kt(ctx, st, "feature_matmul_kernel");
    switch (prologue) {
      case 0: rc = launch_epi<0>(epilogue, grid, st, a, n_rows, ka, lda, d_p, pscale, d_w, kpad, npad, nw, out, ldo); break;
      case 1: rc = launch_epi<1>(epilogue, grid, st, a, n_rows, ka, lda, d_p, pscale, d_w, kpad, npad, nw, out, ldo); break;
      case 2: rc = launch_epi<2>(epilogue, grid, st, a, n_rows, ka, lda, d_p, pscale, d_w, kpad, npad, nw, out, ldo); break;
      default: return wh::fail_msg("wh_feature_matmul", "prologue must be 0 (none), 1 (pre-emphasised power) or 2 (log)");
    }
  }
  if (rc) return rc;
  WH_LAUNCH_CHECK("feature_matmul_kernel");
  return 0;
}

extern "C" int wh_feature_matmul(wh_ctx* ctx, void* stream, const double* a, int64_t n_rows, int ka, int64_t lda,
                                 int prologue, const double* h_p, double pscale, const double* h_w, int nw, int epilogue,
                                 double* out, int64_t ldo) {
  return wh_feature_matmul_tagged(ctx, stream, a, n_rows, ka, lda, prologue, h_p, pscale, h_w, nw, epilogue, out, ldo, 0);
}

extern "C" int wh_context_frames(wh_ctx* ctx, void* stream, const double* x, int64_t n_rows, int d, int w, double* out) {
  if (!ctx || !x || !out) return wh::fail_msg("wh_context_frames", "null argument");
  WH_ENTER(ctx);
  if (n_rows <= 0) return 0;
  if (d < 1 || w < 0) return wh::fail_msg("wh_context_frames", "bad shape");
  hipStream_t st = (hipStream_t)stream;
  { wh::KernelTimer _kt(ctx, st, "context_kernel"); hipLaunchKernelGGL(context_kernel, dim3((unsigned)n_rows), dim3(256), 0, st, x, (long long)n_rows, d, w, out); }
  WH_LAUNCH_CHECK("context_kernel");
  return 0;
}
