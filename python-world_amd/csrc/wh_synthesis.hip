// Pulse-by-pulse overlap-add synthesis, batched.  Replaces synthesis() (world/synthesis.py:21-82).
//
//   prep_kernel     : per output sample: f0/vuv linear interpolation at t_i, phase increment
//                     2*pi*f0/fs (synthesis.py:121-128).  Embarrassingly parallel.
//   phase scan      : per utterance: the cumulative phase is a SEQUENTIAL float64 sum in the reference
//                     (np.cumsum); reproduced bit for bit by an integer prefix sum per binade of the running
//                     sum (exact_cumsum_block), so the pulse positions derived from it are NumPy's.
//   pulse_*_kernel  : wrap phase, detect pulses (|d wrap| > pi), ordered compaction, 1-based sample index and
//                     fractional shift per pulse, noise-stream offsets (synthesis.py:129-138, 65): tile-parallel
//                     mark / scan / emit, then a per-utterance finish.
//   response_kernel : one workgroup per pulse: interpolate the two neighbouring frames, build the
//                     minimum-phase periodic and aperiodic responses with six LDS FFTs
//                     (synthesis.py:86-116,144-180), excite the aperiodic one with zero-mean noise
//                     (direct convolution == the reference's truncated fftfilt), and scatter-add into
//                     y with the reference's clipped-index semantics.
#include "wh_host.h"
#include "wh_math.h"
// response_kernel walks a run of pulses in a loop.  With the plain thread index every per-thread LDS / twiddle address
// of the ~15 transform passes in the loop body is a loop invariant: LLVM hoists them all in front of the loop and keeps
// them alive across it (248 VGPRs, 2 waves per SIMD).  Reading the index through an empty volatile asm makes each use
// its own value; the few integer instructions that are recomputed cost nothing next to 128 spare registers.
__device__ __forceinline__ unsigned wh_opaque_tid() {
  unsigned t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}
#define WH_TID wh_opaque_tid()
#include "wh_device.h"

// -DWH_RESP_STAGE_TIMER: per-stage shader-clock cycles of response_kernel (thread 0 of every workgroup), read with
// wh_debug_resp_stages (tools/resp_stage_timer.py).
#ifdef WH_RESP_STAGE_TIMER
__device__ unsigned long long g_resp_stage[8];
#define RSTAGE_BEGIN unsigned long long _t0 = __builtin_readcyclecounter();
#define RSTAGE_MARK(i) { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long _t = __builtin_readcyclecounter(); atomicAdd(&g_resp_stage[i], _t - _t0); _t0 = _t; } }
#else
#define RSTAGE_BEGIN
#define RSTAGE_MARK(i)
#endif
namespace {

// The spectral half (pulse responses, Requiem frames) may fuse a*b+c into one FP64 instruction; the TIME BASE may not:
// pulse positions are read off its arithmetic, which must round like the reference's (the library is built with
// -ffp-contract=off for that reason).  1: `#pragma clang fp contract(fast)` inside response_pulse / min_phase_response.
#ifndef WH_SYN_CONTRACT
#define WH_SYN_CONTRACT 0  // (python-world_amd/build.py builds with 1)
#endif
#ifndef WH_RESP_TRANS_UNROLL
#define WH_RESP_TRANS_UNROLL 1
#endif
#ifndef WH_RESP_ABLATE
#define WH_RESP_ABLATE 0
#endif
#ifndef WH_RESP_CONV8
#define WH_RESP_CONV8 1  // the noise convolution with eight outputs per thread from N = 2048 up (see response_pulse); 0: four everywhere
#endif
#ifndef WH_FT_SYNTH
#define WH_FT_SYNTH 256
#endif
// Threads cooperating on one pulse / frame: 256 up to N = 1024, 512 from N = 2048 (44.1 / 48 kHz), where the 54 KB
// of LDS per pulse leave two workgroups per CU and the thread count is the occupancy (measured 58.6 -> 50.5 ms on
// config 5).
constexpr int ft_syn(int n) { return n >= 2048 ? 2 * WH_FT_SYNTH : WH_FT_SYNTH; }
// Where it pays: the long noise runs of 44.1 / 48 kHz (config 5: response_kernel 41.8 -> 39.6 ms).  At 16 kHz a pulse's run
// is ~64 samples — two 16-sample rounds per half — and the prologue and the merge cost more than the reads they save
// (config 2: 3.43 -> 3.58 ms), so N = 1024 keeps four outputs per thread.
template <int N>
constexpr bool resp_conv8() { return WH_RESP_CONV8 && N / ft_syn(N) == 4 && N >= 2048; }

struct SynUtt {
  int64_t f_off, nf;      // frames
  int64_t y_off, ny;      // output samples
  int64_t p_off, pcap;    // pulse slots
  int64_t noise_off, noise_len;  // host-supplied noise stream (if any)
  double t0, dt;          // time axis t_i = t0 + i*dt  (NumPy arange semantics, host-computed)
};

// searchsorted-left, hi clipped to [1, nf-1]: the segment SciPy's interp1d(linear, extrapolate) evaluates t on.
// The frame times are almost always an even grid (also after scale_duration), so the answer is first guessed from the
// grid's mean step and checked against its definition (tp[lo-1] < t <= tp[lo]): two rounds of independent loads
// instead of log2(nf) dependent ones; any other time axis falls through to the bisection.  Same result either way.
__device__ __forceinline__ int64_t lerp_segment(const double* __restrict__ tp, int64_t nf, double t) {
  int64_t lo = 0, hi = nf;
  if (nf >= 2) {
    const double first = tp[0], last = tp[nf - 1];
    const double g = ceil((t - first) * (double)(nf - 1) / (last - first));
    if (g >= 1.0 && g <= (double)(nf - 1)) {
      const int64_t gi = (int64_t)g;
      const double a = tp[gi - 1], b = tp[gi], c = gi + 1 < nf ? tp[gi + 1] : b;
      if (a < t && !(b < t)) return gi;                                    // already inside [1, nf-1]
      if (b < t && !(c < t) && gi + 1 <= nf - 1) return gi + 1;
      if (gi >= 2 && !(a < t) && tp[gi - 2] < t) return gi - 1;
    }
  }
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (tp[mid] < t) lo = mid + 1; else hi = mid;
  }
  return lo < 1 ? 1 : (lo > nf - 1 ? nf - 1 : lo);
}
// slope*(t-x_lo)+y_lo on that segment
__device__ __forceinline__ double lerp_on(const double* __restrict__ tp, const double* __restrict__ v, int64_t ih, double t) {
  const int64_t il = ih - 1;
  const double slope = (v[ih] - v[il]) / (tp[ih] - tp[il]);
  return slope * (t - tp[il]) + v[il];
}

// f0_low_limit > 0: f0 is the F0 stage's output and is read as World.encode leaves it after CheapTrick (unvoiced or
// below 3 fs / (fft - 3) -> 500 Hz, cheaptrick.py:26-27,32-33) and D4C (unvoiced -> 0, d4c.py:32): the time base can
// then be computed while those two kernels are still running.
__global__ __launch_bounds__(256) void prep_kernel(const SynUtt* __restrict__ meta, const double* __restrict__ tp,
                                                   const double* __restrict__ f0, const double* __restrict__ vuv,
                                                   double fs, double f0_low_limit, double* __restrict__ phase,
                                                   uint8_t* __restrict__ vuv_s) {
  const SynUtt m = meta[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m.ny) return;
  const double t = m.t0 + (double)i * m.dt;
  const double* tpu = tp + m.f_off;
  const int64_t ih = lerp_segment(tpu, m.nf, t);  // one search serves both interpolants
  double f_raw;
  if (f0_low_limit > 0.0) {
    const double* fu = f0 + m.f_off;
    const double* vu = vuv + m.f_off;
    auto final_f0 = [&](int64_t k) { return vu[k] == 0.0 ? 0.0 : (fu[k] < f0_low_limit ? 500.0 : fu[k]); };
    const int64_t il = ih - 1;
    const double slope = (final_f0(ih) - final_f0(il)) / (tpu[ih] - tpu[il]);
    f_raw = slope * (t - tpu[il]) + final_f0(il);
  } else {
    f_raw = lerp_on(tpu, f0 + m.f_off, ih, t);
  }
  const bool v = lerp_on(tpu, vuv + m.f_off, ih, t) > 0.5;
  double fi = f_raw * (v ? 1.0 : 0.0);
  if (fi == 0.0) fi = fi + 500.0;  // default_f0, synthesis.py:126
  phase[m.y_off + i] = 2 * M_PI * fi / fs;
  vuv_s[m.y_off + i] = v ? 1 : 0;
}

// In-place cumulative sum of NON-NEGATIVE doubles, bit-identical to the sequential float64 sum (np.cumsum: one
// rounding per sample, left to right) — without being sequential.
//
// While the running sum a stays inside one binade [2^k, 2^(k+1)) every partial sum is a multiple of the binade's
// ulp q = 2^(k-52), and fl(a + x) = a + RN_q(x): the rounding of each addend to a multiple of q does not depend on
// a (except for exact ties, which round to the even neighbour of a + x).  So inside a binade the sequence is an
// INTEGER prefix sum of r_j = RN(x_j / q), exact in any order.  One workgroup per utterance walks 2048-sample tiles:
//   * r_j for its 8 samples per thread, thread-local prefix, block scan  -> V_j = a/q + sum r;
//   * the first stop point of the pass — an exact tie, or V_j >= 2^53 (the sum leaves the binade) — is found with
//     min-reductions; everything before it is final (value V_j * q);
//   * the stop element itself is done as the true floating-point add, becomes the new carry, and the pass
//     repeats behind it.  There are ~17 binade crossings and ~1 tie per binade in a whole utterance, so a tile
//     takes one pass almost always.
// 5 ns per sample for the sequential add chain (tools/ubench/chain.hip) becomes ~0.5 ns.
#ifndef WH_XTILE
#define WH_XTILE 4096
#endif
constexpr int kXTile = WH_XTILE;
#ifndef WH_XTHREADS
#define WH_XTHREADS 512
#endif
constexpr int kXThreads = WH_XTHREADS;
constexpr int kXPer = kXTile / kXThreads;
constexpr int kXLds = kXTile + kXTile / kXPer;  // padded tile (xpad)
__device__ __forceinline__ int xpad(int i) { return i + i / kXPer; }  // thread-contiguous runs of kXPer: odd stride in doubles

__device__ __forceinline__ int wave_min_int(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int u = __shfl_xor(v, o, 64);
    v = u < v ? u : v;
  }
  return v;
}

// p[0..n): in place.  xin / xout: kXLds doubles of LDS each; scr: 32 doubles.  One workgroup of kXThreads.
// All integer quantities (r_j, their prefix sums, V_j < 2^53) are carried as integer-valued doubles: exact, and
// the whole pass stays on the FP64 pipe.
// carry_in: the running sum in front of p[0] (0 at the start of a sequence); returns the running sum behind p[n-1].
// (wh::ckp<T>: T* in every shipped build, a range-checked pointer in the bounds build — wh_device.h)
__device__ __forceinline__ double exact_cumsum_block(wh::ckp<double> WH_RESTRICT p, int64_t n, wh::ckp<double> xin, wh::ckp<double> xout,
                                                     wh::ckp<double> scr, double carry_in = 0.0) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  constexpr double kTop = 0x1p53;  // V reaches this: the sum has left the binade
  double carry = carry_in;         // the running sum before the current tile (uniform)
  double pre[kXPer];               // the next tile, in flight from global memory while this one is scanned
#pragma unroll
  for (int q = 0; q < kXPer; ++q) {
    const int64_t i = (int64_t)q * kXThreads + tid;
    pre[q] = i < n ? p[i] : 0.0;
  }
  for (int64_t base = 0; base < n; base += kXTile) {
    const int cnt = (int)(n - base < kXTile ? n - base : kXTile);
#pragma unroll
    for (int q = 0; q < kXPer; ++q) xin[xpad(q * kXThreads + tid)] = pre[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kXPer; ++q) {
      const int64_t i = base + kXTile + (int64_t)q * kXThreads + tid;
      pre[q] = i < n ? p[i] : 0.0;
    }
    double x[kXPer];
#pragma unroll
    for (int j = 0; j < kXPer; ++j) x[j] = xin[xpad(tid * kXPer + j)];
    int s = 0;  // first element of the tile that is not final yet (uniform)
    while (s < cnt) {
      const int ebits = (int)((__double_as_longlong(carry) >> 52) & 0x7ff);
      int jstop = s;  // carry == 0 (or subnormal): fl(carry + x) straight away
      if (ebits != 0) {
        const int sh = 52 - (ebits - 1023);     // x / q = x * 2^sh, exact
        const double c_int = ldexp(carry, sh);  // in [2^52, 2^53)
        double r[kXPer];
        double run = 0.0;
        int first_tie = kXTile;
#pragma unroll
        for (int j = 0; j < kXPer; ++j) {
          const int idx = tid * kXPer + j;
          double rj = 0.0;
          if (idx >= s && idx < cnt) {
            const double sc = fmin(ldexp(x[j], sh), 0x1p54);
            const double fl = floor(sc);
            const double fr = sc - fl;  // exact: sc has at most 53 significant bits
            rj = fl + (fr > 0.5 ? 1.0 : 0.0);
            if (fr == 0.5 && first_tie == kXTile) first_tie = idx;
          }
          run += rj;   // exact while < 2^53; beyond that only "it is >= 2^53" matters, and that it stays
          r[j] = run;  // thread-local inclusive prefix
        }
        double incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const double u = __shfl_up(incl, o, 64);
          if (lane >= o) incl += u;
        }
        if (lane == 63) scr[w] = incl;
        __syncthreads();
        // exclusive prefix from the lanes below only: incl - run would go through this thread's own run, which is
        // inexact (>= 2^53) once the thread lies behind a binade crossing — and would spoil lanes that do not
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 0.0;
        double before = c_int + excl;  // V just before this thread's first element
        for (int i = 0; i < w; ++i) before += scr[i];
        int first_x = kXTile;
#pragma unroll
        for (int j = 0; j < kXPer; ++j) {
          const int idx = tid * kXPer + j;
          if (idx >= s && idx < cnt && before + r[j] >= kTop && first_x == kXTile) first_x = idx;
        }
        const int mine = wave_min_int(first_tie < first_x ? first_tie : first_x);
        const wh::ckp<int> iscr = wh::ck_as<int>(scr + 16);
        if (lane == 0) iscr[w] = mine;
        __syncthreads();
        jstop = iscr[0];
#pragma unroll
        for (int i = 1; i < kXThreads / 64; ++i) jstop = iscr[i] < jstop ? iscr[i] : jstop;
        if (jstop > cnt) jstop = cnt;
        const double q = ldexp(1.0, -sh);
#pragma unroll
        for (int j = 0; j < kXPer; ++j) {
          const int idx = tid * kXPer + j;
          if (idx >= s && idx < jstop) xout[xpad(idx)] = (before + r[j]) * q;  // V < 2^53 times a power of two: exact
        }
        __syncthreads();
      }
      if (jstop < cnt) {  // the stop element: the floating-point add itself
        const double a = jstop == s ? carry : xout[xpad(jstop - 1)];
        const double res = a + xin[xpad(jstop)];
        __syncthreads();  // everyone has read xout[jstop - 1] / scr before they change
        if (tid == 0) xout[xpad(jstop)] = res;
        carry = res;
        s = jstop + 1;
      } else {
        carry = xout[xpad(cnt - 1)];
        s = cnt;
      }
    }
    __syncthreads();
    for (int i = tid; i < cnt; i += kXThreads) p[base + i] = xout[xpad(i)];
    __syncthreads();
  }
  return carry;
}

// The same scan over (begin, end) pairs: pairs[2i] .. pairs[2i+1].
__global__ __launch_bounds__(kXThreads) void exact_cumsum_pairs_kernel(double* __restrict__ data,
                                                                       const int64_t* __restrict__ pairs) {
  __shared__ double xin[kXLds], xout[kXLds], scr[32];
  const int64_t n = pairs[2 * blockIdx.x + 1] - pairs[2 * blockIdx.x];
  exact_cumsum_block(wh::ck_make(data + pairs[2 * blockIdx.x], n, wh::WH_CK_OUT), n, wh::ck_make(xin, kXLds, wh::WH_CK_LDS_MAIN),
                     wh::ck_make(xout, kXLds, wh::WH_CK_LDS_AUX), wh::ck_make(scr, 32, wh::WH_CK_LDS_SCRATCH));
}

// ---- the same scan, tile-parallel -----------------------------------------------------------------------------------
// One workgroup per sequence leaves the chip idle for long sequences (60 s at 48 kHz after scale_duration(2): 5.76 M
// samples on each of 16 workgroups, 4.7 ms).  Inside a binade every partial sum is carry + q * (integer prefix of
// r_j = RN(x_j / q)), and r_j does not depend on the carry unless x_j / q is an exact tie — so a tile of kXTile samples
// that (i) lies inside one binade and (ii) holds no tie needs nothing from its predecessors but the carry, as an
// additive constant.  Five passes:
//   xs_tile_sum_kernel   : plain floating-point tile sums S_t                                  (all tiles in parallel)
//   xs_prefix_kernel     : their running sums A_t per sequence: the carry in front of tile t to ~1e-12, enough to name
//                          its binade unless it sits on a power of two                          (one lane per sequence)
//   xs_tile_total_kernel : with the binade of A_t: T_t = sum r_j (exact integer), tile flagged if a tie, an over-long
//                          step or a zero / subnormal carry shows                               (all tiles in parallel)
//   xs_carry_kernel      : per sequence, in order: a lane walks the unflagged tiles with EXACT carries — checking that
//                          the true carry has the assumed exponent and that carry/q + T_t stays below 2^53 — and
//                          stores each tile's carry; at a flagged tile, or one that fails the check, the whole
//                          workgroup runs the sequential-equivalent exact_cumsum_block on that tile with the exact
//                          carry (a few dozen tiles per sequence: the binade crossings and the ties)
//   xs_apply_kernel      : carry + q * (local integer prefix) for the tiles the walk accepted   (all tiles in parallel)
// Bit-identical to np.cumsum by the same argument as exact_cumsum_block; nothing is accepted on the approximate sums
// alone.
struct XsTile {
  double T;     // sum of r_j in units of q (integer-valued), valid when flag == 0
  int32_t sh;   // x / q = x * 2^sh for the assumed binade
  int32_t flag; // 0: candidate for the closed form, 1: needs the exact block scan, 2: done by xs_carry_kernel
};

__device__ __forceinline__ int xs_find_seq(const int64_t* __restrict__ tile_base, int n_seg, int64_t t) {
  int lo = 0, hi = n_seg;  // largest s with tile_base[s] <= t
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_base[mid] <= t) lo = mid; else hi = mid;
  }
  return lo;
}

// pairs: sequence sq is data[pairs[2 sq] .. pairs[2 sq + 1]); tile_base[sq]: number of tiles in front of it.
__global__ __launch_bounds__(256) void xs_tile_sum_kernel(const double* __restrict__ data, const int64_t* __restrict__ pairs,
                                                          const int64_t* __restrict__ tile_base, int n_seg,
                                                          double* __restrict__ S) {
  __shared__ double red[8];
  const int64_t t = blockIdx.x;
  const int sq = xs_find_seq(tile_base, n_seg, t);
  const int64_t begin = pairs[2 * sq] + (t - tile_base[sq]) * kXTile;
  const int64_t end = begin + kXTile < pairs[2 * sq + 1] ? begin + kXTile : pairs[2 * sq + 1];
  double acc = 0.0;
  for (int64_t i = begin + threadIdx.x; i < end; i += 256) acc += data[i];
  acc = wh::block_sum<256>(acc, red);
  if (threadIdx.x == 0) S[t] = acc;
}

__global__ __launch_bounds__(64) void xs_prefix_kernel(const int64_t* __restrict__ tile_base, int n_seg,
                                                       double* __restrict__ S) {
  const int sq = blockIdx.x * 64 + threadIdx.x;
  if (sq >= n_seg) return;
  double run = 0.0;  // S[t] becomes the (approximate) carry in front of tile t
  wh::serial_run<16>(
      tile_base[sq], tile_base[sq + 1], [](int64_t) { return true; }, [&](int64_t t) { return S[t]; },
      [&](int64_t t) { return S[t]; },
      [&](int64_t t, double v) {
        S[t] = run;
        run += v;
      });
}

__global__ __launch_bounds__(256) void xs_tile_total_kernel(const double* __restrict__ data, const int64_t* __restrict__ pairs,
                                                            const int64_t* __restrict__ tile_base, int n_seg,
                                                            const double* __restrict__ A, XsTile* __restrict__ tiles) {
  __shared__ double red[8];
  __shared__ int bad_any;
  const int64_t t = blockIdx.x;
  const int sq = xs_find_seq(tile_base, n_seg, t);
  const int64_t begin = pairs[2 * sq] + (t - tile_base[sq]) * kXTile;
  const int64_t end = begin + kXTile < pairs[2 * sq + 1] ? begin + kXTile : pairs[2 * sq + 1];
  const double a = A[t];
  const int ebits = (int)((__double_as_longlong(a) >> 52) & 0x7ff);
  if (threadIdx.x == 0) bad_any = 0;
  __syncthreads();
  XsTile out;
  out.T = 0.0;
  out.sh = 0;
  out.flag = 1;
  if (ebits != 0 && ebits != 0x7ff && a > 0.0) {
    const int sh = 52 - (ebits - 1023);
    double acc = 0.0;
    bool bad = false;
    for (int64_t i = begin + threadIdx.x; i < end; i += 256) {
      const double x = data[i];
      const double sc = fmin(ldexp(x, sh), 0x1p54);
      const double fl = floor(sc);
      const double fr = sc - fl;
      bad = bad || fr == 0.5 || !(sc < 0x1p52) || !(x >= 0.0);  // a tie, a step of a whole binade, a negative / NaN addend
      acc += fl + (fr > 0.5 ? 1.0 : 0.0);
    }
    if (bad) bad_any = 1;
    acc = wh::block_sum<256>(acc, red);  // (two barriers: bad_any is visible behind them)
    out.T = acc;
    out.sh = sh;
    out.flag = (bad_any || !(acc < 0x1p52)) ? 1 : 0;  // partial sums below 2^52: every addition was exact
  }
  if (threadIdx.x == 0) tiles[t] = out;
}

__global__ __launch_bounds__(kXThreads) void xs_carry_kernel(double* __restrict__ data, const int64_t* __restrict__ pairs,
                                                             const int64_t* __restrict__ tile_base,
                                                             XsTile* __restrict__ tiles, double* __restrict__ C) {
  constexpr int kWin = 1024;  // tile records staged per round for the walking lane
  __shared__ double xin[kXLds], xout[kXLds], scr[32];
  __shared__ XsTile win[kWin];
  __shared__ long long sh_t;
  __shared__ double sh_a;
  const int sq = blockIdx.x;
  const int64_t t0 = tile_base[sq], t1 = tile_base[sq + 1];
  double a = 0.0;  // exact running sum in front of tile t (block-uniform)
  int64_t t = t0;
  while (t < t1) {
    const int nw = (int)(t1 - t < kWin ? t1 - t : kWin);
    for (int i = threadIdx.x; i < nw; i += kXThreads) win[i] = tiles[t + i];
    __syncthreads();
    if (threadIdx.x == 0) {
      int i = 0;
      double aa = a;
      for (; i < nw; ++i) {
        const XsTile cur = win[i];
        if (cur.flag != 0) break;
        const int ebits = (int)((__double_as_longlong(aa) >> 52) & 0x7ff);
        if (ebits == 0 || 52 - (ebits - 1023) != cur.sh) break;  // the true carry is not in the binade the tile assumed
        const double v = ldexp(aa, cur.sh) + cur.T;                // carry/q + T: both integers below 2^53, exact
        if (!(v < 0x1p53)) break;                                   // the tile would leave the binade
        C[t + i] = aa;
        aa = ldexp(v, -cur.sh);
      }
      sh_t = t + i;
      sh_a = aa;
    }
    __syncthreads();
    const int64_t tn = sh_t;
    a = sh_a;
    const bool stopped = tn < t + nw;  // inside the window: tile tn needs the sequential-equivalent scan
    t = tn;
    __syncthreads();
    if (stopped) {
      const int64_t begin = pairs[2 * sq] + (t - t0) * kXTile;
      const int64_t end = begin + kXTile < pairs[2 * sq + 1] ? begin + kXTile : pairs[2 * sq + 1];
      a = exact_cumsum_block(wh::ck_make(data + begin, end - begin, wh::WH_CK_OUT), end - begin, wh::ck_make(xin, kXLds, wh::WH_CK_LDS_MAIN),
                             wh::ck_make(xout, kXLds, wh::WH_CK_LDS_AUX), wh::ck_make(scr, 32, wh::WH_CK_LDS_SCRATCH), a);
      if (threadIdx.x == 0) tiles[t].flag = 2;
      ++t;
    }
  }
}

__global__ __launch_bounds__(256) void xs_apply_kernel(double* __restrict__ data, const int64_t* __restrict__ pairs,
                                                       const int64_t* __restrict__ tile_base, int n_seg,
                                                       const XsTile* __restrict__ tiles, const double* __restrict__ C) {
  constexpr int PER = kXTile / 256;
  __shared__ double buf[kXTile + kXTile / PER];  // thread-contiguous runs of PER at an odd stride (bank conflicts)
  __shared__ double wsum[4];
  auto pad = [](int i) { return i + i / PER; };
  const int64_t t = blockIdx.x;
  const XsTile tl = tiles[t];
  if (tl.flag != 0) return;
  const int sq = xs_find_seq(tile_base, n_seg, t);
  const int64_t begin = pairs[2 * sq] + (t - tile_base[sq]) * kXTile;
  const int64_t end = begin + kXTile < pairs[2 * sq + 1] ? begin + kXTile : pairs[2 * sq + 1];
  const int cnt = (int)(end - begin);
  const int sh = tl.sh;
  // coalesced in, thread-contiguous through LDS (the order of an integer prefix sum is free), coalesced out
  for (int i = threadIdx.x; i < cnt; i += 256) buf[pad(i)] = data[begin + i];
  __syncthreads();
  double r[PER];
  double run = 0.0;
  const int i0 = threadIdx.x * PER;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int idx = i0 + j;
    double rj = 0.0;
    if (idx < cnt) {
      const double sc = ldexp(buf[pad(idx)], sh);
      const double fl = floor(sc);
      rj = fl + (sc - fl > 0.5 ? 1.0 : 0.0);
    }
    run += rj;
    r[j] = run;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double incl = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double u = __shfl_up(incl, o, 64);
    if (lane >= o) incl += u;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  double before = ldexp(C[t], sh) + (incl - run);  // all integers below 2^53: exact
  for (int i = 0; i < w; ++i) before += wsum[i];
  const double q = ldexp(1.0, -sh);
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int idx = i0 + j;
    if (idx < cnt) buf[pad(idx)] = (before + r[j]) * q;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < cnt; i += 256) data[begin + i] = buf[pad(i)];
}

// Host side of the scan: sequences of more than kXsMinTiles tiles take the tile-parallel passes (all of them in one
// set of launches), the rest the one-workgroup-per-sequence kernel (10 s at 16 kHz is 40 tiles, a dozen of which hold a
// binade crossing or a tie: 0.21 ms either way; 120 s at 48 kHz is 1407 tiles: 1.2 against 4.7 ms).
// h_off: n_seg + 1 offsets into d_data (HOST).
#ifndef WH_XS_MIN_TILES
#define WH_XS_MIN_TILES 64
#endif
constexpr int kXsMinTiles = WH_XS_MIN_TILES;
int exact_cumsum_segments(wh_ctx* ctx, hipStream_t st, double* d_data, const int64_t* h_off, int n_seg) {
  std::vector<int64_t> s_pairs, l_pairs, tb{0};
  for (int i = 0; i < n_seg; ++i) {
    const int64_t tiles = (h_off[i + 1] - h_off[i] + kXTile - 1) / kXTile;
    std::vector<int64_t>& dst = tiles > kXsMinTiles ? l_pairs : s_pairs;
    dst.push_back(h_off[i]);
    dst.push_back(h_off[i + 1]);
    if (tiles > kXsMinTiles) tb.push_back(tb.back() + tiles);
  }
  if (!s_pairs.empty()) {
    int64_t* d_sp = nullptr;
    if (int rc = wh::persistent_upload(ctx, st, "cumsum.short", s_pairs, &d_sp)) return rc;
    { wh::KernelTimer _kt(ctx, st, "phase_kernel"); hipLaunchKernelGGL(exact_cumsum_pairs_kernel, dim3((unsigned)(s_pairs.size() / 2)), dim3(kXThreads), 0, st, d_data, d_sp); }
    WH_LAUNCH_CHECK("exact_cumsum_pairs_kernel");
  }
  if (l_pairs.empty()) return 0;
  const int ns = (int)(l_pairs.size() / 2);
  const int64_t nt = tb[ns];
  int64_t *d_lp = nullptr, *d_tb = nullptr;
  if (int rc = wh::persistent_upload(ctx, st, "cumsum.long", l_pairs, &d_lp)) return rc;
  if (int rc = wh::persistent_upload(ctx, st, "cumsum.tiles", tb, &d_tb)) return rc;
  void* d_scr = nullptr;  // per tile: S / A, C (doubles) and the tile record
  if (int rc = wh::persistent_scratch(ctx, "cumsum.scratch", (size_t)nt * (2 * sizeof(double) + sizeof(XsTile)), &d_scr)) return rc;
  double* d_S = reinterpret_cast<double*>(d_scr);
  double* d_C = d_S + nt;
  XsTile* d_tiles = reinterpret_cast<XsTile*>(d_C + nt);
  { wh::KernelTimer _kt(ctx, st, "xs_tile_sum_kernel"); hipLaunchKernelGGL(xs_tile_sum_kernel, dim3((unsigned)nt), dim3(256), 0, st, d_data, d_lp, d_tb, ns, d_S); }
  { wh::KernelTimer _kt(ctx, st, "xs_prefix_kernel"); hipLaunchKernelGGL(xs_prefix_kernel, dim3((unsigned)((ns + 63) / 64)), dim3(64), 0, st, d_tb, ns, d_S); }
  { wh::KernelTimer _kt(ctx, st, "xs_tile_total_kernel"); hipLaunchKernelGGL(xs_tile_total_kernel, dim3((unsigned)nt), dim3(256), 0, st, d_data, d_lp, d_tb, ns, d_S, d_tiles); }
  { wh::KernelTimer _kt(ctx, st, "xs_carry_kernel"); hipLaunchKernelGGL(xs_carry_kernel, dim3((unsigned)ns), dim3(kXThreads), 0, st, d_data, d_lp, d_tb, d_tiles, d_C); }
  { wh::KernelTimer _kt(ctx, st, "xs_apply_kernel"); hipLaunchKernelGGL(xs_apply_kernel, dim3((unsigned)nt), dim3(256), 0, st, d_data, d_lp, d_tb, ns, d_tiles, d_C); }
  WH_LAUNCH_CHECK("xs_apply_kernel");
  return 0;
}

// Pulse detection (synthesis.py:129-138) in four launches, none of them serial in the utterance length:
//   pulse_mark_kernel   : one workgroup per 1024-sample tile: wrap the phase, mark |d wrap| > pi, count;
//   pulse_scan_kernel   : one workgroup per utterance: exclusive scan of its tile counts, pulse count;
//   pulse_emit_kernel   : one workgroup per tile: ordered compaction into the utterance's pulse slots;
//   pulse_finish_kernel : one workgroup per utterance: fractional shifts and the noise-stream offsets
//                         (exclusive prefix sum of max(3, noise_size), synthesis.py:65).
constexpr int kPTile = 1024;
#ifndef WH_PFINISH
#define WH_PFINISH 1024  // threads of pulse_finish_kernel (one workgroup per utterance); the sanitizer build takes 256
#endif
constexpr int kPFinish = WH_PFINISH;

__device__ __forceinline__ int block_excl_scan_256(int c, int* wsum, int* total) {
  int incl = c;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int uu = __shfl_up(incl, o, 64);
    if (lane >= o) incl += uu;
  }
  __syncthreads();
  if (lane == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  int excl = incl - c, tot = 0;
  for (int w = 0; w < 4; ++w) {
    if (w < (int)(threadIdx.x >> 6)) excl += wsum[w];
    tot += wsum[w];
  }
  *total = tot;
  return excl;
}

__global__ __launch_bounds__(256) void pulse_mark_kernel(const SynUtt* __restrict__ meta, const double* __restrict__ phase,
                                                         int max_tiles, uint8_t* __restrict__ masks,
                                                         int32_t* __restrict__ tile_cnt) {
  __shared__ double wr[kPTile + 1];
  __shared__ int wsum[4];
  const SynUtt m = meta[blockIdx.y];
  const int64_t t0 = (int64_t)blockIdx.x * kPTile;
  if (t0 >= m.ny - 1) return;
  const double* ph = phase + m.y_off;
  const double two_pi = 2 * M_PI;
  for (int i = threadIdx.x; i < kPTile + 1; i += 256) {
    const int64_t g = t0 + i;
    wr[i] = g < m.ny ? fmod(ph[g], two_pi) : 0.0;  // np.remainder of a non-negative value
  }
  __syncthreads();
  unsigned mask = 0;  // 4 consecutive samples per thread
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = threadIdx.x * 4 + q;
    const int64_t g = t0 + i;
    if (g < m.ny - 1 && fabs(wr[i + 1] - wr[i]) > M_PI) mask |= 1u << q;
  }
  int total;
  (void)block_excl_scan_256(__popc(mask), wsum, &total);
  const int64_t slot = (int64_t)blockIdx.y * max_tiles + blockIdx.x;
  masks[slot * 256 + threadIdx.x] = (uint8_t)mask;
  if (threadIdx.x == 0) tile_cnt[slot] = total;
}

__global__ __launch_bounds__(256) void pulse_scan_kernel(const SynUtt* __restrict__ meta, int max_tiles,
                                                         int32_t* __restrict__ tile_cnt, int32_t* __restrict__ p_count,
                                                         int32_t* __restrict__ flags) {
  __shared__ int wsum[4];
  const SynUtt m = meta[blockIdx.x];
  const int tiles = m.ny > 1 ? (int)((m.ny - 1 + kPTile - 1) / kPTile) : 0;
  int32_t* tc = tile_cnt + (int64_t)blockIdx.x * max_tiles;
  int run = 0;
  for (int base = 0; base < tiles; base += 256) {
    const int i = base + threadIdx.x;
    const int c = i < tiles ? tc[i] : 0;
    int total;
    const int excl = block_excl_scan_256(c, wsum, &total);
    if (i < tiles) tc[i] = run + excl;
    run += total;
  }
  if (threadIdx.x == 0) {
    if (run > m.pcap) atomicOr(flags + WH_FLAG_PULSE_OVERFLOW, 1);
    if (run == 0) atomicOr(flags + WH_FLAG_NO_PULSE, 1);
    p_count[blockIdx.x] = run > m.pcap ? (int)m.pcap : run;
  }
}

__global__ __launch_bounds__(256) void pulse_emit_kernel(const SynUtt* __restrict__ meta, int max_tiles,
                                                         const uint8_t* __restrict__ masks,
                                                         const int32_t* __restrict__ tile_pos, double fs,
                                                         double* __restrict__ p_time, int64_t* __restrict__ p_idx) {
  __shared__ int wsum[4];
  const SynUtt m = meta[blockIdx.y];
  const int64_t t0 = (int64_t)blockIdx.x * kPTile;
  if (t0 >= m.ny - 1) return;
  const int64_t slot = (int64_t)blockIdx.y * max_tiles + blockIdx.x;
  const unsigned mask = masks[slot * 256 + threadIdx.x];
  int total;
  int pos = tile_pos[slot] + block_excl_scan_256(__popc(mask), wsum, &total);
  double* pt = p_time + m.p_off;
  int64_t* pi = p_idx + m.p_off;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (mask & (1u << q)) {
      if (pos < m.pcap) {
        const int64_t g = t0 + threadIdx.x * 4 + q;
        const double tt = m.t0 + (double)g * m.dt;
        pt[pos] = tt;
        pi[pos] = (int64_t)floor(tt * fs + 0.5) + 1;  // Decimal ROUND_HALF_UP then +1 (synthesis.py:132)
      }
      ++pos;
    }
  }
}

__global__ __launch_bounds__(kPFinish) void pulse_finish_kernel(const SynUtt* __restrict__ meta,
                                                                const double* __restrict__ phase, double fs,
                                                                const int64_t* __restrict__ p_idx,
                                                                const int32_t* __restrict__ p_count,
                                                                double* __restrict__ p_shift, int64_t* __restrict__ p_noff,
                                                                int32_t* __restrict__ flags) {
  __shared__ long long wsum64[kPFinish / 64];
  const SynUtt m = meta[blockIdx.x];
  const double* ph = phase + m.y_off;
  const int64_t* pi = p_idx + m.p_off;
  double* psh = p_shift + m.p_off;
  int64_t* pn = p_noff + m.p_off;
  const double two_pi = 2 * M_PI;
  const int count = p_count[blockIdx.x];
  long long run = 0;
  for (int base = 0; base < count; base += kPFinish) {
    const int i = base + threadIdx.x;
    long long d = 0;
    if (i < count) {
      int64_t id = pi[i];
      int64_t a = id - 1, b = id;  // wrap_phase[idx-1], wrap_phase[idx]
      a = a < 0 ? 0 : (a > m.ny - 1 ? m.ny - 1 : a);
      b = b < 0 ? 0 : (b > m.ny - 1 ? m.ny - 1 : b);
      const double y1 = fmod(ph[a], two_pi) - 2.0 * M_PI;
      const double y2 = fmod(ph[b], two_pi);
      psh[i] = (-y1 / (y2 - y1)) / fs;
      const int64_t nxt = pi[i + 1 < count ? i + 1 : count - 1];
      const int64_t ns = nxt - id;
      d = ns > 3 ? ns : 3;
    }
    long long incl = d;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const long long uu = __shfl_up(incl, o, 64);
      if (lane >= o) incl += uu;
    }
    __syncthreads();
    if (lane == 63) wsum64[threadIdx.x >> 6] = incl;
    __syncthreads();
    long long excl = incl - d, total = 0;
    for (int w = 0; w < kPFinish / 64; ++w) {
      if (w < (int)(threadIdx.x >> 6)) excl += wsum64[w];
      total += wsum64[w];
    }
    if (i < count) pn[i] = run + excl;
    run += total;
  }
  // (whether a host-supplied noise stream covers `run` draws is tested by wh_synthesis_render, which is the call that
  // knows the stream: noise_cover_kernel)
}

// Exclusive prefix of the per-utterance pulse counts → flat pulse numbering for the response grid.
__global__ void pulse_base_kernel(const int32_t* __restrict__ p_count, int n_utt, int64_t* __restrict__ base) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int64_t run = 0;
    for (int u = 0; u < n_utt; ++u) {
      base[u] = run;
      run += p_count[u];
    }
    base[n_utt] = run;
  }
}

// Everything response_kernel has to know about a pulse before it can touch the spectra, packed by the time base so that
// a workgroup gets it with ONE 64-byte scalar load — and gets the NEXT pulse's under the current pulse's row fetch —
// where it used to walk p_utt -> meta / p_base / p_count -> p_idx, p_shift, p_frames, p_weight, p_noff -> vuv_s: four
// dependent round trips in front of every pulse.
struct alignas(64) PulseRec {
  int64_t pidx;        // 1-based output index of the pulse (pulse_locations_index)
  int64_t rows;        // absolute spectrogram rows: (f_off + earlier frame) | (f_off + later frame) << 32
  double weight;       // of the later frame; -1: both frames are the same one
  double shift;        // pulse_locations_time_shift
  int64_t noff;        // offset of the pulse's noise run in the utterance's stream
  int32_t u;           // utterance
  int32_t noise_size;  // next pulse's index - this one's (0 for the last)
  int32_t vuv;         // interpolated vuv at the pulse (synthesis.py:69 reads it at pidx - 1)
  int32_t pad_[3];
};
static_assert(sizeof(PulseRec) == 64, "one 64-byte scalar load");

// Per pulse: the two frames it interpolates between and the weight of the later one (synthesis.py:49-51,144-180).
// One thread per pulse here, so that the 256-thread response workgroups do not each walk the same 11-deep chain of
// dependent loads (binary search over the frame times) before they can start.
__global__ __launch_bounds__(256) void pulse_frames_kernel(const SynUtt* __restrict__ meta, const double* __restrict__ tp,
                                                           const double* __restrict__ p_time,
                                                           const int64_t* __restrict__ p_idx,
                                                           const double* __restrict__ p_shift,
                                                           const int64_t* __restrict__ p_noff,
                                                           const uint8_t* __restrict__ vuv_s,
                                                           const int32_t* __restrict__ p_count,
                                                           const int64_t* __restrict__ p_base,
                                                           PulseRec* __restrict__ p_rec) {
  const SynUtt m = meta[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int count = p_count[blockIdx.y];
  if (i >= count) return;
  const double* tpu = tp + m.f_off;
  const double ptime = p_time[m.p_off + i];
  // temporal_position_index = interp(tp -> 1..F)(time), clipped to [1, F]
  int64_t lo = 0, hi = m.nf;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (tpu[mid] < ptime) lo = mid + 1; else hi = mid;
  }
  const int64_t ih = lo < 1 ? 1 : (lo > m.nf - 1 ? m.nf - 1 : lo);
  const int64_t il = ih - 1;
  const double slope = ((double)(ih + 1) - (double)(il + 1)) / (tpu[ih] - tpu[il]);
  double pos = slope * (ptime - tpu[il]) + (double)(il + 1);
  pos = fmax(1.0, fmin((double)m.nf, pos));
  const int64_t flo = (int64_t)floor(pos) - 1;
  const int64_t fhi = (int64_t)ceil(pos) - 1;
  const double t1 = tpu[flo], t2 = tpu[fhi];
  const double xq = fmax(t1, fmin(t2, ptime));
  PulseRec r;
  r.pidx = p_idx[m.p_off + i];
  r.rows = (m.f_off + flo) | ((m.f_off + fhi) << 32);
  r.weight = (t1 == t2) ? -1.0 : (xq - t1) / (t2 - t1);
  r.shift = p_shift[m.p_off + i];
  r.noff = p_noff[m.p_off + i];
  r.u = blockIdx.y;
  r.noise_size = (int32_t)(p_idx[m.p_off + (i + 1 < count ? i + 1 : count - 1)] - r.pidx);
  int64_t vi = r.pidx - 1;
  vi = vi < 0 ? 0 : (vi > m.ny - 1 ? m.ny - 1 : vi);
  r.vuv = vuv_s[m.y_off + vi] != 0 ? 1 : 0;
  r.pad_[0] = r.pad_[1] = r.pad_[2] = 0;
  p_rec[p_base[blockIdx.y] + i] = r;  // flat pulse numbering: utterance by utterance, in time order
}

inline int pulse_tiles(int64_t max_ny) { return max_ny > 1 ? (int)((max_ny - 1 + kPTile - 1) / kPTile) : 1; }
// scratch of the pulse stage: crossing masks (one byte per 4 samples) and per-tile counts
inline size_t pulse_scratch_bytes(int B, int64_t max_ny) {
  const size_t mt = (size_t)pulse_tiles(max_ny);
  return (((size_t)B * mt * 256 + 255) & ~(size_t)255) + (((size_t)B * mt * sizeof(int32_t) + 255) & ~(size_t)255);
}
int launch_pulses(wh_ctx* ctx, hipStream_t st, int B, int64_t max_ny, const SynUtt* d_meta, const double* d_phase,
                  double fs, double* d_pt, int64_t* d_pi, double* d_ps, int64_t* d_pn, int32_t* d_pc, char* scratch) {
  const int mt = pulse_tiles(max_ny);
  uint8_t* d_masks = reinterpret_cast<uint8_t*>(scratch);
  int32_t* d_tc = reinterpret_cast<int32_t*>(scratch + (((size_t)B * mt * 256 + 255) & ~(size_t)255));
  { wh::KernelTimer _kt(ctx, st, "pulse_mark_kernel"); hipLaunchKernelGGL(pulse_mark_kernel, dim3(mt, B), dim3(256), 0, st, d_meta, d_phase, mt, d_masks, d_tc); }
  WH_LAUNCH_CHECK("pulse_mark_kernel");
  { wh::KernelTimer _kt(ctx, st, "pulse_scan_kernel"); hipLaunchKernelGGL(pulse_scan_kernel, dim3(B), dim3(256), 0, st, d_meta, mt, d_tc, d_pc, ctx->d_flags); }
  WH_LAUNCH_CHECK("pulse_scan_kernel");
  { wh::KernelTimer _kt(ctx, st, "pulse_emit_kernel"); hipLaunchKernelGGL(pulse_emit_kernel, dim3(mt, B), dim3(256), 0, st, d_meta, mt, d_masks, d_tc, fs, d_pt, d_pi); }
  WH_LAUNCH_CHECK("pulse_emit_kernel");
  { wh::KernelTimer _kt(ctx, st, "pulse_finish_kernel"); hipLaunchKernelGGL(pulse_finish_kernel, dim3(B), dim3(kPFinish), 0, st, d_meta, d_phase, fs, d_pi, d_pc, d_ps, d_pn, ctx->d_flags); }
  WH_LAUNCH_CHECK("pulse_finish_kernel");
  return 0;
}

// ---- counter-based normal generator (Philox-4x32-10 + Box-Muller) for the no-host-noise mode ----
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
  c1 = (uint32_t)p1;
  c3 = (uint32_t)p0;
  c0 = n0;
  c2 = n2;
}
__device__ __attribute__((noinline)) double normal_at(uint64_t seed, uint64_t q) {  // a call: see log_call below
  uint32_t c0 = (uint32_t)(q >> 1), c1 = (uint32_t)((q >> 1) >> 32), c2 = 0x9E3779B9u, c3 = 0x243F6A88u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const double u1 = ((double)c0 * 4294967296.0 + (double)c1 + 0.5) * (1.0 / 18446744073709551616.0);
  const double u2 = ((double)c2 * 4294967296.0 + (double)c3 + 0.5) * (1.0 / 18446744073709551616.0);
  const double rr = sqrt(-2.0 * log(u1));
  double s, c;
  sincospi(2 * u2, &s, &c);  // sin/cos(2*pi*u2) without the large-argument reduction path
  return (q & 1) ? rr * s : rr * c;
}

// Both normals of one Philox block (normal_at(seed, 2*blk) and normal_at(seed, 2*blk + 1), bit for bit): the pulse's
// noise run is generated block-wise, one Box-Muller evaluation per pair instead of one per sample.
__device__ __attribute__((noinline)) double2 normal_pair(uint64_t seed, uint64_t blk) {
  uint32_t c0 = (uint32_t)blk, c1 = (uint32_t)(blk >> 32), c2 = 0x9E3779B9u, c3 = 0x243F6A88u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const double u1 = ((double)c0 * 4294967296.0 + (double)c1 + 0.5) * (1.0 / 18446744073709551616.0);
  const double u2 = ((double)c2 * 4294967296.0 + (double)c3 + 0.5) * (1.0 / 18446744073709551616.0);
  const double rr = sqrt(-2.0 * log(u1));
  double s, c;
  sincospi(2 * u2, &s, &c);
  return make_double2(rr * c, rr * s);
}

// The stream key of utterance u under `seed` (response_kernel derives the same one).
__device__ __host__ __forceinline__ uint64_t philox_key(uint64_t seed, uint64_t u) {
  return seed * 0x9E3779B97F4A7C15ull + u * 0xD1B54A32D192ED03ull + 1;
}
// out[i] = sample q0 + i of utterance u's stream: what wh_philox_normals exposes, so that the device-noise decode can be
// checked sample by sample (tests/test_hip_synthesis.py: the dumped stream fed back as host noise, and to the oracle).
__global__ __launch_bounds__(256) void philox_dump_kernel(uint64_t seed, int32_t u, int64_t q0, int64_t n,
                                                          double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = normal_at(philox_key(seed, (uint64_t)u), (uint64_t)(q0 + i));
}
// Host-supplied noise must cover every draw of the utterance: the last pulse's offset + its run (synthesis.py:93).
__global__ __launch_bounds__(64) void noise_cover_kernel(const SynUtt* __restrict__ meta, const PulseRec* __restrict__ rec,
                                                         const int64_t* __restrict__ p_base, int n_utt,
                                                         int32_t* __restrict__ flags) {
  const int u = blockIdx.x * 64 + threadIdx.x;
  if (u >= n_utt) return;
  const int64_t count = p_base[u + 1] - p_base[u];
  if (count < 1) return;
  const PulseRec& r = rec[p_base[u] + count - 1];
  const int64_t need = r.noff + (r.noise_size > 3 ? r.noise_size : 3);
  if (meta[u].noise_len >= 0 && need > meta[u].noise_len) atomicOr(flags + WH_FLAG_NOISE_SHORT, 1);
}

// Transcendentals of the per-pulse loop as real calls: inlined, their polynomial coefficients (64-bit literals live in
// VGPR pairs) are loop invariants of response_kernel's pulse loop and get parked in registers across the whole body.
#ifndef WH_FAST_MATH64
#define WH_FAST_MATH64 1  // wh_math.h's log / exp / sincospi in the minimum-phase chains (0: the device library's)
#endif
#ifndef WH_TRANS_PAIRS
#define WH_TRANS_PAIRS 1
#endif
// (exp(a0) cos(pi b0), exp(a0) sin(pi b0), exp(a1) cos(pi b1), exp(a1) sin(pi b1)): two bins of a minimum-phase spectrum
__device__ __attribute__((noinline)) double4 cis_pair_call(double a0, double b0, double a1, double b1) {
  const double e0 = exp(a0), e1 = exp(a1);
  double s0, c0, s1, c1;
  sincospi(b0, &s0, &c0);
  sincospi(b1, &s1, &c1);
  return make_double4(e0 * c0, e0 * s0, e1 * c1, e1 * s1);
}
#if WH_FAST_MATH64
__device__ __attribute__((noinline)) double2 log_pair_call(double x, double y) { return make_double2(wh::flog(x), wh::flog(y)); }
__device__ __attribute__((noinline)) double log_call(double x) { return wh::flog(x); }
// (wh::fexp / wh::fsincospi are no shorter than the library's once the compiler has materialised their coefficients — 66 / 75
// instructions against 56 / 82 — and read as a scalar table they stall on its latency: measured, not used)
__device__ __attribute__((noinline)) double exp_call(double x) { return exp(x); }
__device__ __attribute__((noinline)) double2 sincospi_call(double x) {
  double s, c;
  sincospi(x, &s, &c);
  return make_double2(s, c);
}
#else
__device__ __attribute__((noinline)) double2 log_pair_call(double x, double y) { return make_double2(log(x), log(y)); }
__device__ __attribute__((noinline)) double log_call(double x) { return log(x); }
__device__ __attribute__((noinline)) double exp_call(double x) { return exp(x); }
__device__ __attribute__((noinline)) double2 sincospi_call(double x) {
  double s, c;
  sincospi(x, &s, &c);
  return make_double2(s, c);
}
#endif

// Minimum-phase response (synthesis.py:100-116; synthesisRequiem.py:112-118) from the mirrored log-amplitude to the time
// domain, with the O(N) passes between the three transforms fused:
//   in : zr[n] = log|S[min(n, N-n)]| / 2, n < N (real, even), visible;  out: time-domain response N * h[n] in zr.
//   (1) forward transform of a real EVEN sequence: its spectrum is real, so the post-pass of the half-size transform
//       computes real parts only, and writes them where the next transform wants them — folded onto the upper half,
//       doubled (cepstrum fold) — instead of: post-pass -> copy real parts -> fold (three LDS round trips, three barriers);
//   (2) after the second transform a thread holds the pair of bins (k, N/2-k) in registers through the post-pass, the
//       complex exponential AND the pre-pass of the inverse real transform: exp(r.x/N) * cis(-r.y/N - delay*k), where
//       `delay_pi` (units of pi per bin) is the pulse's fractional delay — the reference multiplies the spectrum by
//       exp(-i*coef*shift*k) afterwards (synthesis.py:61-64); folding it into the angle saves one sincospi and one
//       complex product per bin, and the four passes over the half spectrum become one.
// `mul(k, E)`: what the minimum-phase bin k (0 <= k <= N/2) is multiplied with before the inverse transform — identity
// for the pulse responses, the excitation frame's spectrum in the Requiem filter (synthesisRequiem.py:112-118).
struct SpectrumIdentity {
  __device__ __forceinline__ double2 operator()(int, double2 e) const { return e; }
};
template <int N, int GT, class Mul = SpectrumIdentity>
__device__ __forceinline__ void min_phase_response(wh::ckp<double2> zb, wh::ckp<const double2> tw_base, double delay_pi, Mul mul = Mul()) {
#if WH_SYN_CONTRACT
#pragma clang fp contract(fast)
#endif
  constexpr int FT = ft_syn(N);
  constexpr int M = N / 2;
  constexpr int PP = (M / 2 + 1 + GT - 1) / GT;  // bin pairs (k, M-k), k <= M/2, per thread
  const wh::ckp<double> zr = wh::ck_as<double>(zb);
  const int gt = WH_TID & (GT - 1);
  const wh::ckp<const double2> WH_RESTRICT w = tw_base + N;
#if defined(WH_RESP_ABLATE_T1) && WH_RESP_ABLATE_T1
  wh::sync<FT>();  // TIMING EXPERIMENT ONLY (wrong results): the chain's first transform costs nothing — twice the upper
#else              // bound of packing the two chains' real-even first transforms into one (DCT-I)
  wh::fft_lds<M, false, GT, FT>(zb, tw_base + M);
#endif
  {
    double ck[PP], cm[PP];
#pragma unroll
    for (int p = 0; p < PP; ++p) {
      const int k = gt + p * GT;
      ck[p] = cm[p] = 0.0;
      if (k <= M / 2) {
        const double2 a = zb[k], b = zb[M - k];
        if (k == 0) {
          ck[p] = a.x + a.y;
          cm[p] = a.x - a.y;
        } else {
          const double er = 0.5 * (a.x + b.x), dr = 0.5 * (a.x - b.x), di = 0.5 * (a.y + b.y);
          const double2 wk = wh::ldg2(w + k);
          const double tr = fma(wk.x, di, wk.y * dr);
          ck[p] = er + tr;  // Re X[k]
          cm[p] = er - tr;  // Re X[M-k]
        }
      }
    }
    wh::sync<FT>();  // every pair has been read
#pragma unroll
    for (int p = 0; p < PP; ++p) {
      const int k = gt + p * GT;
      if (k <= M / 2) {
        if (k == 0) {
          zr[0] = ck[p];
          zr[M] = 2 * cm[p];
        } else {
          zr[N - k] = 2 * ck[p];
          zr[M + k] = 2 * cm[p];  // (k = M/2: the same slot, the same value)
        }
      }
    }
    for (int n = 1 + gt; n < M; n += GT) zr[n] = 0.0;
    wh::sync<FT>();
  }
  wh::fft_lds<M, false, GT, FT>(zb, tw_base + M);
#pragma unroll 1
  for (int k = gt; k <= M / 2; k += GT) {
    const double2 a = zb[k], b = zb[M - k];
    double2 x0, x1;  // R[k], R[M-k]: spectrum of the folded cepstrum
    const double2 wk = wh::ldg2(w + k);
    if (k == 0) {
      x0 = make_double2(a.x + a.y, 0.0);
      x1 = make_double2(a.x - a.y, 0.0);
    } else {
      const double er = 0.5 * (a.x + b.x), ei = 0.5 * (a.y - b.y);
      const double dr = 0.5 * (a.x - b.x), di = 0.5 * (a.y + b.y);
      const double tr = fma(wk.x, di, wk.y * dr);
      const double ti = fma(wk.y, di, -(wk.x * dr));
      x0 = make_double2(er + tr, ei + ti);
      x1 = make_double2(er - tr, ti - ei);
    }
    // minimum-phase spectrum exp(conj(R) / N) with the fractional delay in the angle (both in units of pi)
#if WH_TRANS_PAIRS
    // the pair of bins through ONE call: the library's exp / sincospi spend a third of their instructions putting polynomial
    // coefficients into registers, and two evaluations inside one function share them
    const double4 cis = cis_pair_call(x0.x / N, -x0.y / N * M_1_PI - delay_pi * (double)k, x1.x / N,
                                      -x1.y / N * M_1_PI - delay_pi * (double)(M - k));
    const double e0 = 1.0, e1 = 1.0;
    const double2 s0 = make_double2(cis.y, cis.x), s1 = make_double2(cis.w, cis.z);  // (sin, cos), already scaled by the exponential
#else
    const double e0 = exp_call(x0.x / N), e1 = exp_call(x1.x / N);
    const double2 s0 = sincospi_call(-x0.y / N * M_1_PI - delay_pi * (double)k);
    const double2 s1 = sincospi_call(-x1.y / N * M_1_PI - delay_pi * (double)(M - k));
#endif
    double2 A = mul(k, make_double2(e0 * s0.y, e0 * s0.x)), B = mul(M - k, make_double2(e1 * s1.y, e1 * s1.x));
    if (k == 0) {  // DC and Nyquist bins: only their real parts reach a real output
      A.y = 0.0;
      B.y = 0.0;
    }
    // pre-pass of the inverse real transform (wh::irfft_lds) on the pair
    const double er = A.x + B.x, ei = A.y - B.y;
    const double dr = A.x - B.x, di = A.y + B.y;
    const double orr = fma(dr, wk.x, di * wk.y);
    const double oi = fma(di, wk.x, -(dr * wk.y));
    zb[k] = make_double2(er - oi, ei + orr);
    if (k != 0) zb[M - k] = make_double2(er + oi, orr - ei);
  }
  wh::sync<FT>();
  wh::fft_lds<M, true, GT, FT>(zb, tw_base + M);
}

// padded index of the aperiodic response for the register-tiled convolution: 2 doubles of padding every 32
// keep the 16-byte pair reads of lanes that are 4..8 samples apart on different LDS banks
__device__ __forceinline__ int rap_index(int i) { return i + 2 * (i >> 5); }

// First pulse of an utterance at or behind 1-based index `lo` (its pulse indices are ascending): a 64-ary search by the
// whole wave — probes at 64 evenly spaced pulses, a ballot, the same again inside the bracket — two rounds of loads for
// the few thousand pulses of an utterance where a bisection takes twelve dependent ones.  Wave-uniform.
__device__ __forceinline__ int first_pulse_at(const int64_t* __restrict__ pi, int count, int64_t lo) {
  const int lane = threadIdx.x & 63;
  int base = 0, n = count;  // the answer is in [base, base + n]
  while (n > 0) {
    const int stride = (n + 63) / 64;
    const int idx = base + lane * stride;
    const bool below = idx < base + n && pi[idx] < lo;
    const int c = __popcll(__ballot(below));  // the probes are ascending: the first c of them are below
    if (stride == 1) {
      base += c;
      break;
    }
    if (c == 0) break;  // pi[base] >= lo
    const int nb_ = base + (c - 1) * stride + 1;
    const int left = base + n - nb_;
    n = stride - 1 < left ? stride - 1 : left;
    base = nb_;
  }
  return base;
}

// Everything one pulse needs (kernel arguments bundled so that the per-pulse body can be a real function).
struct RespArgs {
  const SynUtt* meta;
  const double* tp;
  const double* spectrogram;
  const double* aperiodicity;
  double fs;
  const PulseRec* p_rec;
  const int64_t* p_base;
  int n_utt;
  const double* noise;
  uint64_t seed;
  const double* dc_base;
  const double2* tw_base;
  double* rows;             // overlap-add rows of the runs (response_gather_kernel sums them into y)
  const int64_t* row_base;  // [B + 1]: where every utterance's region of `rows` begins (sized from ITS sample count)
  const int64_t* run_base;  // [n_utt + 1] first run of every utterance (pulse_run_base_kernel)
  const int64_t* row_off;   // [n_utt][runs_cap] where run r's row begins in the utterance's region (pulse_rows_kernel)
  int64_t runs_cap;
};

// Overlap-add of a workgroup's run of consecutive pulses OF ONE UTTERANCE: the run's contributions are accumulated, in
// pulse order, in an N-sample LDS ring that covers the window of the current pulse; when the window moves on, the
// samples that leave it are final for this run and go to the run's ROW (plain stores, zeros included) — row r of an
// utterance holds the sum of run r over the samples its pulses cover, and response_gather_kernel adds the rows that
// cover an output sample in run order.  No atomics anywhere: the decode is the same from run to run, and the same
// whether an utterance is decoded alone, in a batch or on another rank (runs are numbered per utterance).  The
// reference adds pulse after pulse into y (synthesis.py:67-81); summing runs of pulses first is another association
// of the same sum (1e-17 relative).
// Row layout (per utterance a region of row_base[u + 1] - row_base[u] doubles): the rows lie one behind the other, row r at row_off[r]
// (pulse_rows_kernel: an exclusive scan of the row lengths, which follow from the pulse positions); slot 0 = what the
// run adds to the LAST sample (Q8, below), slot 1 + (t - start_r) = its sum at the 1-based sample t < ny, start_r =
// max(1, first tap of the run's first pulse).  A region holds 12 doubles per output sample (a mean f0 up to ~fs / 16 at
// N = 1024); an utterance that needs more raises WH_FLAG_PULSE_OVERFLOW like one that runs out of pulse slots, and the
// retry with the safe pulse capacity sizes the region for it.
struct RunState {
  bool any;           // a pulse has been accumulated (the ring holds something)
  int64_t win_start;  // 1-based output index of the first sample of the ring's window
  int64_t row_start;  // start_r
  double last;        // thread FT-1: the run's contribution to the utterance's last sample
};
#ifndef WH_RESP_RUN
#define WH_RESP_RUN 0  // pulses per workgroup; 0: by transform length (resp_run below)
#endif

// Samples [a, b) (1-based, within the ring's current window) are final for this run: to the row, clear the ring.
template <int N>
__device__ __forceinline__ void ring_flush(wh::ckp<double> ring, int64_t a, int64_t b, wh::ckp<double> WH_RESTRICT row, int64_t row_start,
                                           int64_t ny) {
  constexpr int FT = ft_syn(N);
  a = a < 1 ? 1 : a;
  b = b > ny ? ny : b;
  for (int64_t tgt = a + WH_TID; tgt < b; tgt += FT) {
    const int slot = (int)(tgt & (N - 1));
    row[1 + (tgt - row_start)] = ring[slot];
    ring[slot] = 0.0;
  }
}

// One pulse of a run.
template <int N>
__device__ __forceinline__ void response_pulse(const RespArgs& A, const PulseRec& rec, char* smem, wh::ckp<double> ring, RunState& rs,
                                               wh::ckp<double> WH_RESTRICT row,
                                               const double (&dcw)[N / ft_syn(N) <= 4 ? N / ft_syn(N) : 1]) {
#if WH_SYN_CONTRACT
#pragma clang fp contract(fast)
#endif
  const SynUtt* __restrict__ meta = A.meta;
  const double* __restrict__ spectrogram = A.spectrogram;
  const double* __restrict__ aperiodicity = A.aperiodicity;
  const double fs = A.fs;
  const double* __restrict__ noise = A.noise;
  const uint64_t seed = A.seed;
  const double* __restrict__ dc_base = A.dc_base;
  const double2* __restrict__ tw_raw = A.tw_base;
  asm volatile("" : "+s"(tw_raw));  // per pulse: no twiddle address / value of one pulse survives into the next
  const wh::ckp<const double2> tw_base = wh::ck_make(tw_raw, 2 * WH_MAX_TWIDDLE, wh::WH_CK_TWIDDLE);
  constexpr int FT = ft_syn(N);
  constexpr int K = N / 2 + 1;
  constexpr int NZ = 256;
  constexpr int NZC = resp_conv8<N>() ? 252 : NZ;  // noise samples per chunk (the eight-output form walks them twelve at a time)
  constexpr int R = N / FT;  // consecutive output samples per thread
  static_assert(R % 2 == 0 && NZ % (2 * R) == 0, "pairwise reads; whole blocks of 2R noise samples");
  constexpr int GT = FT >= 256 ? FT / 2 : FT;  // threads per chain: the periodic and aperiodic chains run side by side
  constexpr int NG = FT / GT;
  // (wh::ckp<T> is T* in every shipped build; the bounds build checks each access against the range named here)
  const wh::ckp<double> lds_all = wh::ck_make(reinterpret_cast<double*>(smem), 2 * (N + 2) + (N + N / 16 + 2) + NZ + 32, wh::WH_CK_LDS_OTHER);
  const wh::ckp<double> zrA = wh::ck_sub(lds_all, 0, N + 2, wh::WH_CK_LDS_MAIN);       // N/2+1 complex: aperiodic chain
  const wh::ckp<double2> zbA = wh::ck_as<double2>(zrA);
  const wh::ckp<double> zrP = wh::ck_sub(lds_all, N + 2, N + 2, wh::WH_CK_LDS_AUX);    // N/2+1 complex: periodic chain
  const wh::ckp<double2> zbP = wh::ck_as<double2>(zrP);
  const wh::ckp<double> rap = wh::ck_sub(lds_all, 2 * (N + 2), N + N / 16 + 2, wh::WH_CK_LDS_OTHER);  // padded aperiodic response
  const wh::ckp<double> nz = wh::ck_sub(lds_all, 2 * (N + 2) + (N + N / 16 + 2), NZ, wh::WH_CK_LDS_OTHER);
  const wh::ckp<double> scratch = wh::ck_sub(lds_all, 2 * (N + 2) + (N + N / 16 + 2) + NZ, 32, wh::WH_CK_LDS_SCRATCH);

  RSTAGE_BEGIN
  wh::sync<FT>();
  const int u = rec.u;
  const SynUtt m = meta[u];  // (output / noise offsets: not needed before the noise fetch and the overlap-add)
  const int64_t pidx = rec.pidx;
  const double shift = rec.shift;
  const int64_t noise_size = rec.noise_size;

  // ---- spectral parameters of this pulse (synthesis.py:49-51,144-180) -------------------------
  // the two neighbouring frames and the interpolation weight, from pulse_frames_kernel
  const int64_t row_lo = rec.rows & 0xffffffffll, row_hi = rec.rows >> 32;
  const double bw = rec.weight;
  const bool same = bw < 0.0;
  const double b = same ? 0.0 : bw;
  const double a = 1 - b;
  const double* s_lo = spectrogram + row_lo * K;
  const double* s_hi = spectrogram + row_hi * K;
  const double* a_lo = aperiodicity + row_lo * K;
  const double* a_hi = aperiodicity + row_hi * K;
  // a thread's bins k = tid + q FT: all of their row loads are issued before the first log (a call: nothing is moved
  // across it), one global round trip per pulse instead of one per bin
  constexpr int KQ = (K + FT - 1) / FT;
  double rsl[KQ], rsh[KQ], ral[KQ], rah[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int k = WH_TID + q * FT;
    const int kc = k < K ? k : K - 1;  // (clamped: always a valid address; the surplus slot is not used)
    rsl[q] = s_lo[kc];
    rsh[q] = s_hi[kc];
    ral[q] = a_lo[kc];
    rah[q] = a_hi[kc];
  }
  // aperiodic_slice[0] decides voicing (synthesis.py:69); its two loads ride with the rows' (issued first, they put
  // two more dependent round trips in front of the rows: the compiler waited for each before going on)
  double ap0_lo = a_lo[0], ap0_hi = a_hi[0];
  asm volatile("" : "+v"(ap0_lo), "+v"(ap0_hi));  // (both issued here: else the second is sunk behind the test of `same`)
  double aper0;
  {
    const double al = ap0_lo * ap0_lo, ah = ap0_hi * ap0_hi;
    aper0 = same ? al : a * al + b * ah;
  }
  const bool voiced = (rec.vuv != 0) && (aper0 <= 0.999);
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int k = WH_TID + q * FT;
    if (k >= K) break;
    const double sl = rsl[q], sh = rsh[q];
    const double al = ral[q] * ral[q], ah = rah[q] * rah[q];
    const double pl = fmax(0.001, 1 - al), ph = fmax(0.001, 1 - ah);
    const double sp = same ? sl : a * sl + b * sh;
    const double pe = same ? pl : a * pl + b * ph;
    const double ap = same ? al : a * al + b * ah;
    double v = sp * pe;  // periodic spectrum
    if (v == 0.0) v = 2.220446049250313e-16;
    double w = voiced ? sp * ap : sp;  // aperiodic spectrum
    if (w == 0.0) w = 2.220446049250313e-16;
    // log|.| / 2 of the Hermitian-mirrored spectrum (synthesis.py:103-105), written where the chain's first
    // transform reads it: no amplitude arrays, no separate log and mirror passes
#if WH_TRANS_PAIRS
    // (a voiced pulse's two logarithms through one call, like the pair of complex exponentials in min_phase_response)
    double2 lg;
    if (voiced) lg = log_pair_call(fabs(w), fabs(v));
    else lg = make_double2(log_call(fabs(w)), 0.0);
    const double lw = lg.x / 2;
#else
    const double lw = log_call(fabs(w)) / 2;
#endif
    zrA[k] = lw;
    if (k > 0 && k < N / 2) zrA[N - k] = lw;
    if (voiced) {
#if WH_TRANS_PAIRS
      const double lv = lg.y / 2;
#else
      const double lv = log_call(fabs(v)) / 2;
#endif
      zrP[k] = lv;
      if (k > 0 && k < N / 2) zrP[N - k] = lv;
    }
  }
  // ---- noise for this pulse: max(3, noise_size) samples, zero-mean (synthesis.py:93-95) -----------
  const int64_t nd = noise_size > 3 ? noise_size : 3;
  const int64_t noff = rec.noff;
  auto noise_at = [&](int64_t j) -> double {
    if (noise) {
      const int64_t q = noff + j;
      return q < m.noise_len ? noise[m.noise_off + q] : 0.0;
    }
    return normal_at(philox_key(seed, (uint64_t)u), (uint64_t)(noff + j));
  };
  double mean;
  {
    double part = 0.0;
    if (noise) {
      for (int64_t j = WH_TID; j < nd; j += FT) {
        const double v = noise_at(j);
        part += v;
        if (j < NZ) nz[j] = v;  // the usual case nd <= NZ: generate / fetch each sample once
      }
    } else {
      // device stream: sample q of the utterance is one half of Philox block q >> 1 — walk the blocks the run touches
      const uint64_t key = philox_key(seed, (uint64_t)u);
      const int64_t b1 = (noff + nd - 1) >> 1;
      for (int64_t blk = (noff >> 1) + WH_TID; blk <= b1; blk += FT) {
        const double2 z = normal_pair(key, (uint64_t)blk);
        const int64_t j = 2 * blk - noff;  // index of the block's first half within this pulse's run (-1 .. nd-1)
        if (j >= 0) {
          part += z.x;
          if (j < NZ) nz[j] = z.x;
        }
        if (j + 1 < nd) {
          part += z.y;
          if (j + 1 < NZ) nz[j + 1] = z.y;
        }
      }
    }
    mean = wh::block_sum<FT>(part, scratch) / (double)nd;  // barriers: the log spectra and nz are visible
  }

  RSTAGE_MARK(0)
  // ---- minimum-phase responses (synthesis.py:86-116): aperiodic chain on thread group 0, periodic chain on
  //      group 1, advancing through the same barrier phases (with a single group: one after the other) --------
#if WH_RESP_ABLATE == 2
  if (WH_TID == 0) row[0] = zrA[3] + zrP[5] + mean;
  return;
#endif
  const double coef_pi = 2.0 * fs / N;  // coefficient = 2*pi*fs/N (synthesis.py:59), kept in units of pi
  if (NG == 2 && voiced) {
    const int g = WH_TID / GT;
    min_phase_response<N, GT>(g == 0 ? zbA : zbP, tw_base, g == 0 ? 0.0 : coef_pi * shift);
  } else {
    // an unvoiced pulse has no periodic response (synthesis.py:69-75): one chain, on all the threads — 40 % of the
    // pulses of speech-like input (the 500 Hz default rate of unvoiced stretches) do half the transform work
    min_phase_response<N, FT>(zbA, tw_base, 0.0);
    if (voiced) min_phase_response<N, FT>(zbP, tw_base, coef_pi * shift);
  }
  RSTAGE_MARK(2)
  // zrA[n] = N * aperiodic response, zrP[n] = N * periodic response (both before fftshift)
  for (int n = WH_TID; n < N; n += FT) rap[rap_index(n)] = zrA[(n + N / 2) & (N - 1)] / N;
  wh::sync<FT>();

  // y[m] = sum_j nz[j] * ra[m-j], m < N: each thread owns R consecutive outputs and slides an R-wide
  // register window over the response, two noise samples (one 16-byte LDS read each side) per step.
  double acc[R];
#pragma unroll
  for (int q = 0; q < R; ++q) acc[q] = 0.0;
  double acc8[resp_conv8<N>() ? 8 : 1];
#pragma unroll
  for (int q = 0; q < (resp_conv8<N>() ? 8 : 1); ++q) acc8[q] = 0.0;
  const int m0 = WH_TID * R;
#if WH_RESP_ABLATE == 1
  for (int64_t j0 = 0; j0 < 0; j0 += NZC) {
#else
  for (int64_t j0 = 0; j0 < nd; j0 += NZC) {
#endif
    const int cnt = (int)(nd - j0 < NZC ? nd - j0 : NZC);
    wh::sync<FT>();
    for (int j = WH_TID; j < NZ; j += FT) {
      double v = 0.0;
      if (j < cnt) v = (j0 == 0 ? nz[j] : noise_at(j0 + j)) - mean;
      nz[j] = v;  // zero padded to an even count
    }
    wh::sync<FT>();
    if constexpr (resp_conv8<N>()) {
      // EIGHT outputs per thread, the noise range of the chunk split over the two halves of the workgroup (round 6).
      // With four outputs per thread a block of 4 noise samples is 16 FMAs against four 16-byte LDS reads (two for the
      // noise, two for the new response group): 32 LDS cycles per 64 FMA cycles of a wave, and the CU's four SIMDs share ONE
      // LDS pipe (MI355X_MICROARCH.md) — twice what it can feed.  At 48 kHz, where a pulse's noise run is ~100-200 samples
      // against a 2048-sample response, the convolution was a third of the kernel (tools/resp_stage_timer.py 48000 16 60
      // 1.5 2.0) and LDS-bound.  Eight outputs per thread: the same four reads feed 32 FMAs.  Half h of the workgroup
      // takes the outputs m = 8 t .. 8 t + 7 (t = tid mod FT/2) over ITS half of the noise samples; the two partial sums
      // meet in LDS behind the loop.  Response samples travel as aligned groups of four through a ring of three register
      // groups: nothing is shifted.
      constexpr int HT = FT / 2;
      const int half_id = WH_TID / HT, t8 = WH_TID - half_id * HT;
      const int c_all = ((cnt + 11) / 12) * 12;              // (<= NZC = 252; nz is zero-padded to NZ)
      const int c_mid = ((c_all / 12 + 1) / 2) * 12;          // half 0: [0, c_mid), half 1: [c_mid, c_all)
      const int jb = half_id == 0 ? 0 : c_mid, je = half_id == 0 ? c_mid : c_all;
      auto load4 = [&](int base, double (&g)[4]) {  // base is a multiple of 4: all four valid or all in front of the response
        double2 v0 = make_double2(0.0, 0.0), v1 = make_double2(0.0, 0.0);
        if (base >= 0) {
          v0 = wh::ck_as<const double2>(rap + rap_index(base))[0];
          v1 = wh::ck_as<const double2>(rap + rap_index(base + 2))[0];
        }
        g[0] = v0.x; g[1] = v0.y; g[2] = v1.x; g[3] = v1.y;
      };
      // outputs q = 0..7 at noise step s = 0..3 read ra[mb + q - s]: hi = ra[mb+4 .. mb+7], mid = ra[mb .. mb+3], lo = ra[mb-4 .. mb-1]
      auto block4 = [&](int j, const double (&hi)[4], const double (&mid)[4], const double (&lo)[4]) {
#pragma unroll
        for (int sp = 0; sp < 4; sp += 2) {  // two noise samples at a time: one 16-byte read
          const double2 nn = wh::ck_as<const double2>(nz + (j + sp))[0];
#pragma unroll
          for (int sft = sp; sft < sp + 2; ++sft)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int idx = q - sft;  // -3 .. 7
              acc8[q] = fma(sft == sp ? nn.x : nn.y, idx >= 4 ? hi[idx - 4] : (idx >= 0 ? mid[idx] : lo[idx + 4]), acc8[q]);
            }
        }
      };
      // Three register groups in a ring (a fourth, fetched a block ahead, cost 28 spilled registers and 26 GB of scratch
      // traffic per config-5 step): the group a block has finished with receives the next block's lowest samples.
      const int mb0 = t8 * 8 - (int)j0 - jb;  // response index of output 0 at the half's first noise sample
      double g0[4], g1[4], g2[4];
      load4(mb0 + 4, g0);
      load4(mb0, g1);
      load4(mb0 - 4, g2);
      for (int j = jb; j < je; j += 12) {
        const int mb = t8 * 8 - (int)j0 - j;
        block4(j, g0, g1, g2);
        load4(mb - 8, g0);
        block4(j + 4, g1, g2, g0);
        load4(mb - 12, g1);
        block4(j + 8, g2, g0, g1);
        load4(mb - 16, g2);
      }
    } else if constexpr (R <= 4) {
      // Aligned groups of R response samples around the thread's outputs: hi = ra[mb .. mb+R-1], lo = ra[mb-R .. mb-1],
      // mb = m0 - j0 - j.  A block of R noise samples needs exactly these two groups (output q at step s reads
      // ra[mb + q - s]); for the next block lo becomes hi and ONE new group is fetched — into the registers of the group
      // that just died, so nothing is ever shifted (the two-step version moved 2(R-1) doubles per pair of steps).
      auto load_group = [&](int base, double (&g)[R]) {  // base is a multiple of R: a group is all-valid or all before the start
  #pragma unroll
        for (int t = 0; t < R; t += 2) {
          double2 v = make_double2(0.0, 0.0);
          if (base >= 0) v = wh::ck_as<const double2>(rap + rap_index(base + t))[0];
          g[t] = v.x;
          g[t + 1] = v.y;
        }
      };
      auto block = [&](int j, const double (&hi)[R], const double (&lo)[R]) {
        double n[R];
  #pragma unroll
        for (int t = 0; t < R; t += 2) {
          const double2 v = wh::ck_as<const double2>(nz + (j + t))[0];
          n[t] = v.x;
          n[t + 1] = v.y;
        }
  #pragma unroll
        for (int sft = 0; sft < R; ++sft)
  #pragma unroll
          for (int q = 0; q < R; ++q) acc[q] = fma(n[sft], q - sft >= 0 ? hi[q - sft] : lo[R + q - sft], acc[q]);
      };
      double ga[R], gb[R];
      const int mb0 = m0 - (int)j0;
      load_group(mb0, ga);
      load_group(mb0 - R, gb);
      const int steps = ((cnt + 2 * R - 1) / (2 * R)) * (2 * R);  // nz is zero-padded up to NZ, a multiple of 2R
      for (int j = 0; j < steps; j += 2 * R) {
        block(j, ga, gb);
        load_group(mb0 - j - 2 * R, ga);
        block(j + R, gb, ga);
        load_group(mb0 - j - 3 * R, gb);
      }
    } else {
      // (R = 8, fft size 4096: the 64-FMA blocks of the shift-free form do not fit the register budget)
      double r[R];
  #pragma unroll
      for (int q = 0; q < R; ++q) {
        const int idx = m0 + q - (int)j0;
        r[q] = idx >= 0 ? rap[rap_index(idx)] : 0.0;
      }
      const int steps = (cnt + 1) & ~1;
      for (int j = 0; j < steps; j += 2) {
        const double2 nn = wh::ck_as<const double2>(nz + j)[0];
        const int inew = m0 - (int)j0 - j - 2;  // even: (ra[inew], ra[inew+1]) is an aligned pair
        double2 fresh = make_double2(0.0, 0.0);
        if (inew >= 0) fresh = wh::ck_as<const double2>(rap + rap_index(inew))[0];
  #pragma unroll
        for (int q = 0; q < R; ++q) acc[q] = fma(nn.x, r[q], acc[q]);
  #pragma unroll
        for (int q = R - 1; q > 0; --q) r[q] = r[q - 1];
        r[0] = fresh.y;  // ra[m0 - g - 1]
  #pragma unroll
        for (int q = 0; q < R; ++q) acc[q] = fma(nn.y, r[q], acc[q]);
  #pragma unroll
        for (int q = R - 1; q > 0; --q) r[q] = r[q - 1];
        r[0] = fresh.x;  // ra[m0 - g - 2]
      }
    }
  }

  if constexpr (resp_conv8<N>()) {
    // the two halves' partial sums meet: half 0 parks its eight outputs in the aperiodic chain's buffer, half 1 in the
    // padded response's (both free now), and every thread collects the four outputs the overlap-add expects of it
    constexpr int HT = FT / 2;
    const int half_id = WH_TID / HT, t8 = WH_TID - half_id * HT;
    wh::sync<FT>();  // every thread is done reading rap
    const wh::ckp<double> park = half_id == 0 ? zrA : rap;
#pragma unroll
    for (int q = 0; q < 8; q += 2) wh::ck_as<double2>(park + (t8 * 8 + q))[0] = make_double2(acc8[q], acc8[q + 1]);
    wh::sync<FT>();
#pragma unroll
    for (int q = 0; q < R; q += 2) {
      const double2 a = wh::ck_as<const double2>(zrA + (m0 + q))[0], b2 = wh::ck_as<const double2>(rap + (m0 + q))[0];
      acc[q] = a.x + b2.x;
      acc[q + 1] = a.y + b2.y;
    }
  }
  RSTAGE_MARK(3)
  // ---- DC removal of the periodic response (synthesis.py:72-73) ------------------------------------
  double dc_total = 0.0;
  const double gain = sqrt((double)(noise_size > 1 ? noise_size : 1));
  if (voiced) {
    double part = 0.0;
    for (int n = WH_TID; n < N; n += FT) part += zrP[n] / N;
    dc_total = wh::block_sum<FT>(part, scratch);
  }

  // ---- overlap-add with the reference's clipped fancy-index semantics (Q8), through the run's ring ----------
  const int64_t s1 = pidx - N / 2 + 1;  // 1-based index of this pulse's first tap
  if (rs.any) {
    const int64_t e = s1 < rs.win_start + N ? s1 : rs.win_start + N;  // the samples the window leaves behind
    ring_flush<N>(ring, rs.win_start, e, row, rs.row_start, m.ny);
    // (pulses more than N samples apart — f0 below fs / N: the samples between the two windows belong to the row too)
    for (int64_t tgt = rs.win_start + N + WH_TID; tgt < (s1 < m.ny ? s1 : m.ny); tgt += FT) row[1 + (tgt - rs.row_start)] = 0.0;
    wh::sync<FT>();
  } else {
    rs.row_start = s1 < 1 ? 1 : s1;
  }
  rs.any = true;
  rs.win_start = s1;
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const int mm = m0 + q;
    const int64_t tgt = s1 + mm;
    double v = acc[q];
    // (the eight-output form reads the weight where it uses it: four values held across the convolution were registers it lacked)
    if (voiced) v += (zrP[(mm + N / 2) & (N - 1)] / N + (R <= 4 && !resp_conv8<N>() ? dcw[R <= 4 ? q : 0] : dc_base[mm]) * -dc_total) * gain;
    if (tgt < 1) continue;                    // clipped to 1 and overwritten by the in-range tap
    if (tgt < m.ny) ring[(int)(tgt & (N - 1))] += v;    // this thread is the only writer of its R slots
    else if (mm == N - 1) rs.last += v;                 // last duplicate wins on the high side: the last sample's share
  }
  RSTAGE_MARK(4)
}

// Pulses per workgroup: 6 up to N = 1024, 8 beyond (measured with the pulse record prefetch in place: 4 / 5 / 6 / 7 / 8 /
// 12 / 16 pulses 3.414 / 3.416 / 3.417 / 3.449 / 3.46 / 3.51 / 3.59 ms at config 2; at 48 kHz, N = 2048, 5 pulses 42.5
// against 41.6 ms for 8: the flush of the longer ring is what a short run does not amortise).
constexpr int resp_run(int n) { return WH_RESP_RUN > 0 ? WH_RESP_RUN : (n <= 1024 ? 6 : 8); }

// First run of every utterance: runs never straddle utterances, so that what a run sums does not depend on the
// utterance's position in the batch.
__global__ void pulse_run_base_kernel(const int32_t* __restrict__ p_count, int n_utt, int run, int64_t* __restrict__ base) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int64_t at = 0;
    for (int u = 0; u < n_utt; ++u) {
      base[u] = at;
      at += (p_count[u] + run - 1) / run;
    }
    base[n_utt] = at;
  }
}

// Where every run's row begins in its utterance's region: row r holds 1 + (end_r - start_r) doubles (RunState), the rows
// lie one behind the other.  One workgroup per utterance scans its runs in chunks of 256.  A row that does not fit the
// region gets offset -1 (its run is skipped, its samples are not gathered) and raises WH_FLAG_PULSE_OVERFLOW.
template <int N>
__global__ __launch_bounds__(256) void pulse_rows_kernel(const SynUtt* __restrict__ meta, const int64_t* __restrict__ p_idx,
                                                         const int32_t* __restrict__ p_count, int64_t runs_cap,
                                                         const int64_t* __restrict__ row_base, int64_t* __restrict__ row_off,
                                                         int32_t* __restrict__ flags) {
  constexpr int RUN = resp_run(N);
  __shared__ int wsum[4];
  const SynUtt m = meta[blockIdx.x];
  const int count = p_count[blockIdx.x];
  const int64_t* pi = p_idx + m.p_off;
  const int n_runs = (count + RUN - 1) / RUN;
  int64_t* out = row_off + (int64_t)blockIdx.x * runs_cap;
  int64_t carry = 0;
  bool over = false;
  const int64_t room = row_base[blockIdx.x + 1] - row_base[blockIdx.x];
  for (int base = 0; base < n_runs; base += 256) {
    const int r = base + threadIdx.x;
    int len = 0;
    if (r < n_runs) {
      const int kf = r * RUN;
      const int kl = (kf + RUN < count ? kf + RUN : count) - 1;
      int64_t start = pi[kf] - N / 2 + 1, end = pi[kl] + N / 2 + 1;
      start = start < 1 ? 1 : start;
      end = end < m.ny ? end : m.ny;
      len = 1 + (int)(end > start ? end - start : 0);
    }
    int total;
    const int excl = block_excl_scan_256(len, wsum, &total);
    if (r < n_runs) {
      const int64_t at = carry + excl;
      const bool fits = at + len <= room;
      out[r] = fits ? at : -1;
      over = over || !fits;
    }
    carry += total;
    __syncthreads();  // wsum is reused by the next chunk
  }
  if (over) atomicOr(flags + WH_FLAG_PULSE_OVERFLOW, 1);
}

// One workgroup per RUN of resp_run(N) consecutive pulses of one utterance (runs numbered utterance by utterance, in
// time order).  The grid is sized from the host's pulse capacity; the runs that exist (device-side pulse counts) are
// dealt to the XCDs in contiguous ranges, so that the spectrogram / aperiodicity rows neighbouring pulses share are
// fetched into one L2 — the surplus workgroups exit at once.
template <int N>
__global__ __launch_bounds__(ft_syn(N), 4) void response_kernel(RespArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int FT = ft_syn(N);
  constexpr int RUN = resp_run(N);
  const int64_t n_runs = A.run_base[A.n_utt];
  const int64_t run = wh::xcd_unit(blockIdx.x, n_runs);
  if (run >= n_runs) return;
  int u = 0;
  {  // the utterance of this run: last u with run_base[u] <= run (scalar loads, block-uniform)
    int lo = 0, hi = A.n_utt;  // run_base[lo] <= run < run_base[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (A.run_base[mid] <= run) lo = mid; else hi = mid;
    }
    u = lo;
  }
  const int64_t r_in_utt = run - A.run_base[u];
  const wh::ckp<double> ring = wh::ck_make(reinterpret_cast<double*>(smem) + (2 * (N + 2) + (N + N / 16 + 2) + 256 + 32), N, wh::WH_CK_LDS_OTHER);
  for (int i = threadIdx.x; i < N; i += FT) ring[i] = 0.0;
  RunState rs{false, 0, 1, 0.0};
  const int64_t gp0 = A.p_base[u] + r_in_utt * RUN;
  const int64_t gp1 = gp0 + RUN < A.p_base[u + 1] ? gp0 + RUN : A.p_base[u + 1];
  const int64_t my_off = A.row_off[(int64_t)u * A.runs_cap + r_in_utt];
  if (my_off < 0) return;  // the utterance's row region is full (flagged by pulse_rows_kernel)
  // slot i of the row at row[i]; what is left of the utterance's region behind the row's start is the range it may touch
  const wh::ckp<double> row = wh::ck_make(A.rows + A.row_base[u] + my_off, A.row_base[u + 1] - A.row_base[u] - my_off, wh::WH_CK_OUT);
  // A pulse's record travels as ONE dword per lane (lane i & 15 holds dword i) and is turned into scalars by
  // v_readlane at the top of the pulse that uses it — a pulse after its load was issued.  Fetched as a struct the
  // compiler made scalars of it (readfirstlane) right behind the load: the "prefetch" was waited for at once.
  auto fetch_rec = [&](int64_t gp) -> uint32_t {
    return reinterpret_cast<const uint32_t*>(A.p_rec + gp)[threadIdx.x & 15];
  };
  auto unpack_rec = [&](uint32_t word) -> PulseRec {
    uint32_t d[16];
#pragma unroll
    for (int i = 0; i < 13; ++i) d[i] = (uint32_t)__builtin_amdgcn_readlane((int)word, i);
    PulseRec r;
    r.pidx = (int64_t)(((uint64_t)d[1] << 32) | d[0]);
    r.rows = (int64_t)(((uint64_t)d[3] << 32) | d[2]);
    r.weight = __hiloint2double((int)d[5], (int)d[4]);
    r.shift = __hiloint2double((int)d[7], (int)d[6]);
    r.noff = (int64_t)(((uint64_t)d[9] << 32) | d[8]);
    r.u = (int32_t)d[10];
    r.noise_size = (int32_t)d[11];
    r.vuv = (int32_t)d[12];
    r.pad_[0] = r.pad_[1] = r.pad_[2] = 0;
    return r;
  };
  uint32_t cur_w = fetch_rec(gp0);
  // the DC-removal weights of this thread's R output samples are the same for every pulse: read once per run (R <= 4;
  // beyond that they would cost the registers the convolution needs)
  constexpr int R = N / FT;
  double dcw[R <= 4 ? R : 1];
  if constexpr (R <= 4) {
#pragma unroll
    for (int q = 0; q < R; ++q) dcw[q] = A.dc_base[threadIdx.x * R + q];
  } else {
    dcw[0] = 0.0;
  }
  // the row's first sample is known from the run's first pulse: max(1, its first tap)
  int64_t row_start;
  {
    const int64_t pidx0 = reinterpret_cast<const int64_t*>(A.p_rec + gp0)[0];
    row_start = pidx0 - N / 2 + 1;
    row_start = row_start < 1 ? 1 : row_start;
  }
  (void)row_start;
#pragma unroll 1
  for (int64_t gp = gp0; gp < gp1; ++gp) {
    const PulseRec cur = unpack_rec(cur_w);
    const uint32_t nxt_w = fetch_rec(gp + 1 < gp1 ? gp + 1 : gp);  // in flight under this whole pulse
    response_pulse<N>(A, cur, smem, ring, rs, row, dcw);
    cur_w = nxt_w;
  }
  wh::sync<FT>();
  if (rs.any) ring_flush<N>(ring, rs.win_start, rs.win_start + N, row, rs.row_start, A.meta[u].ny);
  if (threadIdx.x == FT - 1) row[0] = rs.last;
}

// y[t] = sum of the rows that cover t, in run order (see RunState): one thread per output sample.  The runs whose
// pulses reach a tile of 256 samples follow from the utterance's ascending pulse indices (the wave search of the
// Requiem excitation); a run's extent from its first and last pulse.  The utterance's LAST sample receives, of every
// pulse whose window reaches it or beyond, the last tap only (the reference's clipped fancy-index assignment keeps the
// last of the duplicates, Q8): the rows' slot 0.
#ifndef WH_GATHER_PER
#define WH_GATHER_PER 4
#endif
constexpr int kGatherTile = 256 * WH_GATHER_PER;  // output samples per workgroup: one pulse search for all of them
template <int N>
__global__ __launch_bounds__(256) void response_gather_kernel(const SynUtt* __restrict__ meta,
                                                              const int64_t* __restrict__ p_idx,
                                                              const int32_t* __restrict__ p_count,
                                                              const double* __restrict__ rows, const int64_t* __restrict__ row_base,
                                                              const int64_t* __restrict__ row_off, int64_t runs_cap,
                                                              double* __restrict__ y) {
  constexpr int RUN = resp_run(N);
  constexpr int PER = WH_GATHER_PER;
  const SynUtt m = meta[blockIdx.y];
  const int64_t n0 = (int64_t)blockIdx.x * kGatherTile;
  if (n0 >= m.ny) return;
  const int count = p_count[blockIdx.y];
  const int64_t* pi = p_idx + m.p_off;
  const double* ru = rows + row_base[blockIdx.y];
  const int64_t* ro = row_off + (int64_t)blockIdx.y * runs_cap;
  // pulses whose window [pidx - N/2 + 1, pidx + N/2] reaches the tile's samples n0 + 1 .. n0 + kGatherTile
  const int k0 = first_pulse_at(pi, count, n0 + 1 - N / 2);
  const int n_runs = (count + RUN - 1) / RUN;
  double sum[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) sum[q] = 0.0;
  const int64_t t0 = n0 + 1 + threadIdx.x;  // this thread's samples: t0 + 256 q (1-based)
  for (int r = k0 / RUN; r < n_runs; ++r) {
    const int kf = r * RUN;
    const int kl = (kf + RUN < count ? kf + RUN : count) - 1;
    const int64_t s1f = pi[kf] - N / 2 + 1;
    if (s1f > n0 + kGatherTile) break;
    const int64_t start = s1f < 1 ? 1 : s1f;
    int64_t end = pi[kl] + N / 2 + 1;  // one behind the last tap of the run's last pulse ...
    end = end < m.ny ? end : m.ny;     // ... and the last sample takes the rows' slot 0 only (below)
    const int64_t off = ro[r];
    if (off < 0) continue;  // (dropped: the region was full — flagged)
    const double* rr = ru + off + 1 - start;  // sample t of the run at rr[t]: slot 1 + (t - start)
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int64_t tgt = t0 + 256 * q;
      if (tgt >= start && tgt < end) sum[q] += rr[tgt];
    }
  }
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int64_t tgt = t0 + 256 * q;
    if (tgt < m.ny) y[m.y_off + tgt - 1] = sum[q];
  }
  if (n0 + kGatherTile >= m.ny) {  // the tile that holds the utterance's last sample: block-uniform
    const int k_end = first_pulse_at(pi, count, m.ny - N / 2);  // first pulse whose last tap reaches the last sample
    if (threadIdx.x == 0) {
      double last = 0.0;
      for (int r = k_end / RUN; r < n_runs; ++r) {
        const int64_t off = ro[r];
        if (off >= 0) last += ru[off];
      }
      y[m.y_off + m.ny - 1] = last;
    }
  }
}


template <int N>
int launch_resp(wh_ctx* ctx, hipStream_t st, int B, int64_t pcap_max, int64_t max_ny, const std::vector<int64_t>& h_ny, const SynUtt* d_meta, const double* tp,
                const double* spec, const double* ap, double fs, const PulseRec* p_rec, const int64_t* p_base,
                const int64_t* p_idx, const int32_t* p_count, const double* noise, uint64_t seed, double* y) {
  std::vector<double> dc(N);
  double sum = 0.0;
  for (int n = 0; n < N; ++n) {  // hanning(N+2)[1:-1] normalised (synthesis.py:57-58)
    dc[n] = 0.5 - 0.5 * cos(2.0 * M_PI * (double)(n + 1) / (double)(N + 1));
    sum += dc[n];
  }
  for (int n = 0; n < N; ++n) dc[n] /= sum;
  const double* d_dc = nullptr;
  if (int rc = wh::const_table(ctx, "dc_base:" + std::to_string(N), dc, &d_dc)) return rc;
  const size_t lds = sizeof(double) * (2 * (N + 2) + (N + N / 16 + 2) + 256 + 32 + N);  // ... + the overlap-add ring
  if (int rc = wh::allow_lds(&response_kernel<N>, lds)) return rc;
  // overlap-add rows (RunState): per utterance ceil(pcap / RUN) rows of N + 1 slots laid along the time axis.  Held in a
  // buffer of its own, not in the arena: the render reserves no workspace (the time base may live in this context's)
  const int64_t runs_cap = (pcap_max + resp_run(N) - 1) / resp_run(N);
  // doubles per output sample in an utterance's row region: 12 with the default pulse capacity (ny / 8) — the rows take
  // 6.2 where the reference places its 500 Hz unvoiced pulses at 16 kHz, 5 at a voiced 400 Hz, 12 at ~1 kHz — up to 48 with
  // the safe capacity of the retry (ny / 2: f0 up to fs / 4 at N = 1024)
  int64_t per_sample = (96 * pcap_max + max_ny - 1) / (max_ny > 0 ? max_ny : 1);
  per_sample = per_sample < 12 ? 12 : (per_sample > 48 ? 48 : per_sample);
  // every utterance's region from ITS OWN sample count (sized from the longest utterance a ragged batch held
  // max_ny per utterance, ~1 GB per 64 x 10 s whatever the other lengths were)
  std::vector<int64_t> row_base((size_t)B + 1, 0);
  for (int u = 0; u < B; ++u) row_base[u + 1] = row_base[u] + per_sample * h_ny[u] + 4 * (N + 1);
  int64_t* d_row_base = nullptr;
  if (int rc = wh::persistent_upload(ctx, st, "syn.row_base", row_base, &d_row_base)) return rc;
  void* d_rows = nullptr;
  void* d_rb = nullptr;
  void* d_ro = nullptr;
  if (int rc = wh::persistent_scratch(ctx, "syn.ola_rows", sizeof(double) * (size_t)row_base[B], &d_rows)) return rc;
  if (int rc = wh::persistent_scratch(ctx, "syn.run_base", sizeof(int64_t) * ((size_t)B + 1), &d_rb)) return rc;
  if (int rc = wh::persistent_scratch(ctx, "syn.row_off", sizeof(int64_t) * (size_t)runs_cap * B, &d_ro)) return rc;
  { wh::KernelTimer _kt(ctx, st, "pulse_run_base_kernel"); hipLaunchKernelGGL(pulse_run_base_kernel, dim3(1), dim3(64), 0, st, p_count, B, resp_run(N), reinterpret_cast<int64_t*>(d_rb)); }
  WH_LAUNCH_CHECK("pulse_run_base_kernel");
  { wh::KernelTimer _kt(ctx, st, "pulse_rows_kernel"); hipLaunchKernelGGL(pulse_rows_kernel<N>, dim3(B), dim3(256), 0, st, d_meta, p_idx, p_count, runs_cap, d_row_base, reinterpret_cast<int64_t*>(d_ro), ctx->d_flags); }
  WH_LAUNCH_CHECK("pulse_rows_kernel");
  // one workgroup per run of resp_run(N) pulse slots; runs past the real pulse count exit at once
  const int64_t grid = wh::xcd_grid(runs_cap * B);
  { wh::KernelTimer _kt(ctx, st, "response_kernel"); RespArgs ra{d_meta, tp, spec, ap, fs, p_rec, p_base, B, noise, seed, d_dc, ctx->d_twiddle, reinterpret_cast<double*>(d_rows), d_row_base, reinterpret_cast<const int64_t*>(d_rb), reinterpret_cast<const int64_t*>(d_ro), runs_cap};
  hipLaunchKernelGGL(response_kernel<N>, dim3((unsigned)grid), dim3(ft_syn(N)), lds, st, ra); }
  WH_LAUNCH_CHECK("response_kernel");
  { wh::KernelTimer _kt(ctx, st, "response_gather_kernel"); hipLaunchKernelGGL(response_gather_kernel<N>, dim3((unsigned)((max_ny + kGatherTile - 1) / kGatherTile), B), dim3(256), 0, st, d_meta, p_idx, p_count, reinterpret_cast<const double*>(d_rows), d_row_base, reinterpret_cast<const int64_t*>(d_ro), runs_cap, y); }
  WH_LAUNCH_CHECK("response_gather_kernel");
  return 0;
}


// ================================================================================================
// Requiem synthesis (world/synthesisRequiem.py:12-141): excitation = band-weighted seed noise +
// band-mixed seed pulses, then frame-wise minimum-phase filtering with overlap-add.
// ================================================================================================
struct ReqUtt {
  int64_t hop;        // int((tp[1]-tp[0])*fs), host-evaluated
  int64_t cursor[8];  // per-band start position in the circular noise seed
  int64_t row_off;    // first overlap-add row of the utterance (req_filter_kernel), in doubles
  int64_t n_runs;     // its runs of frames
};

__global__ __launch_bounds__(256) void req_linap_kernel(const double* __restrict__ band_db, int64_t count,
                                                        double* __restrict__ lin) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < count) lin[i] = pow(10.0, band_db[i] / 10);  // synthesisRequiem.py:125
}

// bracketing frames of time t (SciPy interp1d linear + extrapolate)
__device__ __forceinline__ void bracket(const double* __restrict__ tp, int64_t nf, double t, int64_t* il, int64_t* ih) {
  *ih = lerp_segment(tp, nf, t);  // guess from the grid's mean step, checked; bisection otherwise
  *il = *ih - 1;
}

// Per pulse: the gain sqrt(max(1, next index - this one)) — 0 for a pulse the reference skips (unvoiced at its sample, or
// lowest-band aperiodicity above 0.999, synthesisRequiem.py:55) — and the band weights 1 - ap_b at the pulse's sample
// (synthesisRequiem.py:57-60,66-71).  One thread per pulse: the chain of dependent look-ups (pulse index, voicing,
// bracketing frames, band rows) is paid once per pulse here, with no atomics behind it.
__global__ __launch_bounds__(256) void req_pulse_weights_kernel(const SynUtt* __restrict__ meta, const double* __restrict__ tp,
                                                                const double* __restrict__ lin, int nb,
                                                                const int64_t* __restrict__ p_idx,
                                                                const int32_t* __restrict__ p_count,
                                                                const uint8_t* __restrict__ vuv_s,
                                                                double* __restrict__ p_gain, double* __restrict__ p_w) {
  const SynUtt m = meta[blockIdx.y];
  const int count = p_count[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const int64_t pidx = p_idx[m.p_off + i];
  int64_t p = pidx - 1;
  p = p < 0 ? 0 : (p > m.ny - 1 ? m.ny - 1 : p);
  double gain = 0.0;
  if (vuv_s[m.y_off + p] != 0) {
    const double t = m.t0 + (double)p * m.dt;
    const double* tpu = tp + m.f_off;
    int64_t il, ih;
    bracket(tpu, m.nf, t, &il, &ih);
    const double dx = tpu[ih] - tpu[il];
    double w0 = 0.0;
    for (int b = 0; b < nb; ++b) {
      const double y_lo = lin[(m.f_off + il) * nb + b], y_hi = lin[(m.f_off + ih) * nb + b];
      const double w = (y_hi - y_lo) / dx * (t - tpu[il]) + y_lo;
      if (b == 0) w0 = w;
      p_w[(m.p_off + i) * nb + b] = 1 - w;
    }
    if (!(w0 > 0.999)) {
      const int64_t nxt = p_idx[m.p_off + (i + 1 < count ? i + 1 : count - 1)];
      const int64_t ns = nxt - pidx;
      gain = sqrt((double)(ns > 1 ? ns : 1));
    }
  }
  p_gain[m.p_off + i] = gain;
}

// The excitation signal (synthesisRequiem.py:27-63), one thread per output sample: the aperiodic component (band noises
// weighted by the interpolated aperiodicities) plus the periodic one GATHERED from the pulses whose 512-tap band-mixed
// seed covers the sample, in pulse order — the order in which the reference accumulates them, so the sum is the
// reference's, bit for bit, and the same from run to run.  (The scatter form, one wave per pulse adding its taps with
// atomics on top of the noise, was bound by the rate of those atomics: 5.4 + 1.2 ms for the two kernels at 1024
// utterances.)  The reference's clipped fancy-index assignment (Q8) keeps, of the taps that fall before the first or
// behind the last sample, only the LAST one written: taps before sample 1 are dropped (the in-range tap of index 1 is
// written after them), and the last sample receives the last tap of every pulse that reaches it or beyond.
__global__ __launch_bounds__(256) void req_excite_kernel(const SynUtt* __restrict__ meta, const ReqUtt* __restrict__ rq,
                                                         const double* __restrict__ tp, const double* __restrict__ lin,
                                                         int nb, const double* __restrict__ noise_seed, int64_t nlen,
                                                         const double* __restrict__ pulse_seed, int pfft,
                                                         const int64_t* __restrict__ p_idx, const int32_t* __restrict__ p_count,
                                                         const double* __restrict__ p_gain, const double* __restrict__ p_w,
                                                         double* __restrict__ exc) {
  const SynUtt m = meta[blockIdx.y];
  const int64_t n0 = (int64_t)blockIdx.x * 256;
  if (n0 >= m.ny) return;
  const int count = p_count[blockIdx.y];
  const int64_t* pi = p_idx + m.p_off;
  const double* pg = p_gain + m.p_off;
  const double* pw = p_w + m.p_off * nb;
  // pulses whose taps reach this tile: index in [first sample - pfft/2, last sample + pfft/2 - 1] (1-based)
  const int64_t lo = n0 + 1 - pfft / 2, hi = n0 + 256 + pfft / 2 - 1;
  const int k0 = first_pulse_at(pi, count, lo);
  const int k_end = first_pulse_at(pi, count, m.ny - pfft / 2);  // first pulse whose last tap reaches the last sample
  const int64_t i = n0 + threadIdx.x;
  if (i >= m.ny) return;
  const int64_t tgt = i + 1;
  double periodic = 0.0;
  // (the pulse records are read with scalar loads, the same for every thread of the tile; staging the tile's pulses in
  // LDS first — one round of coalesced loads, two barriers — is slower: 3.77 against 3.45 ms at 1024 utterances)
  if (tgt < m.ny) {
    for (int k = k0; k < count; ++k) {
      const int64_t pidx = pi[k];
      if (pidx > hi) break;
      const double gain = pg[k];
      if (gain == 0.0) continue;
      const int64_t mm = tgt - pidx + pfft / 2 - 1;
      if (mm >= 0 && mm < pfft) {
        double r = 0.0;
        for (int b = 0; b < nb; ++b) r += pulse_seed[mm * nb + b] * pw[(int64_t)k * nb + b];
        periodic += r * gain;
      }
    }
  } else {
    for (int k = k_end; k < count; ++k) {
      const double gain = pg[k];
      if (gain == 0.0) continue;
      double r = 0.0;
      for (int b = 0; b < nb; ++b) r += pulse_seed[(int64_t)(pfft - 1) * nb + b] * pw[(int64_t)k * nb + b];
      periodic += r * gain;
    }
  }
  const double t = m.t0 + (double)i * m.dt;
  const double* tpu = tp + m.f_off;
  int64_t il, ih;
  bracket(tpu, m.nf, t, &il, &ih);
  const double dx = tpu[ih] - tpu[il];
  double aperiodic = 0.0;
  const bool nlen_pow2 = (nlen & (nlen - 1)) == 0;
  for (int b = 0; b < nb; ++b) {
    const double y_lo = lin[(m.f_off + il) * nb + b], y_hi = lin[(m.f_off + ih) * nb + b];
    const double ap = (y_hi - y_lo) / dx * (t - tpu[il]) + y_lo;
    const int64_t at = rq[blockIdx.y].cursor[b] + i;  // circular read of the band's noise seed (synthesisRequiem.py:131-141)
    const int64_t pos = nlen_pow2 ? (at & (nlen - 1)) : at % nlen;  // (the default table lengths are powers of two)
    aperiodic += noise_seed[pos * nb + b] * ap;
  }
  exc[m.y_off + i] = periodic + aperiodic;  // synthesisRequiem.py:62
}

// The Hanning window of the Requiem frames, hanning(2 hop + 1)[1:-1] (synthesisRequiem.py:84-86): the same for every frame of
// every utterance with that hop.  Evaluated on the device with req_filter_kernel's own expression (bitwise what the kernel
// computes in place), cached per context and window length.
__global__ void req_hann_kernel(double* __restrict__ w, int wlen) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < wlen) w[j] = 0.5 - 0.5 * cospi(2.0 * (double)(j + 1) / (double)(wlen + 1));
}

// frames per run of req_filter_kernel: 4 up to N = 1024 — measured at config 4 (filter + gather) with the run's sums in
// LDS: 1 frame 1.40 + 0.22 ms, 4 frames 1.57 + 0.08, 8 frames 1.70 + 0.06, 16 frames 1.96 + 0.05; one frame per row
// beyond (no benchmark config decodes Requiem there).  At the north-star size (1024 x 10 s, round 6): 1 frame 23.4 + 3.2 ms
// (nine workgroups per CU instead of six: -7 % for +50 % of the waves — the kernel is not waiting for occupancy), 2 frames
// 25.3 + 2.0, 4 frames 25.1 + 1.2, 8 frames 27.0 + 0.9
#ifndef WH_REQ_RUNF
#define WH_REQ_RUNF 4
#endif
constexpr int req_runf(int n) { return n <= 1024 ? WH_REQ_RUNF : 1; }

// Frame-wise minimum-phase filtering of the excitation with overlap-add (synthesisRequiem.py:74-101), WITHOUT atomics:
// a workgroup takes a run of RUNF consecutive frames of one utterance, adds their responses — in frame order — into an
// LDS accumulator that spans the run ((RUNF - 1) hop + N samples), and writes it as the run's ROW; req_gather_kernel
// then adds, per output sample, the two or three rows that cover it, in run order.  The same sum from launch to launch
// and wherever the utterance sits in a batch (runs are numbered per utterance); the reference adds frame after frame
// into y — runs of frames first is another association of that sum.  Row r of an utterance: W = (RUNF - 1) hop + N + 1
// doubles at row_off + r W; slot 0 = the run's share of the utterance's LAST sample (Q8: of the taps clipped onto it
// only the last one written survives — the last tap of every frame whose response reaches it or beyond), slot 1 + j =
// the sum at the 1-based sample a_r + j, a_r = r RUNF hop + 1.  RUNF = 1 (long transforms, long hops): the row is the
// frame's own response, written straight from the transform buffer.  Rows instead of atomics take the 1.07 GB of
// read-modify-write traffic per 64 utterances down to a 0.33 GB row write + as much read by the gather.
#ifndef WH_REQ_MINW
#define WH_REQ_MINW 1
#endif
template <int N, int RUNF>
__global__ __launch_bounds__(ft_syn(N), (RUNF > 1 && N <= 1024 ? WH_REQ_MINW : 1)) void req_filter_kernel(const SynUtt* __restrict__ meta, const ReqUtt* __restrict__ rq,
                                                        const double* __restrict__ spectrogram,
                                                        const double* __restrict__ exc,
                                                        const double2* __restrict__ tw_base_arg, double* rows,
                                                        const double* __restrict__ hann) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int FT = ft_syn(N);
  constexpr int K = N / 2 + 1;
  const double2* tw_raw = tw_base_arg;
  const SynUtt m = meta[blockIdx.y];
  const ReqUtt q = rq[blockIdx.y];
  // (wh::ckp<T> is T* in every shipped build; the bounds build checks each access against the range named here)
  const wh::ckp<double> zr = wh::ck_make(reinterpret_cast<double*>(smem), N + 2, wh::WH_CK_LDS_MAIN);  // minimum-phase half spectrum (N/2+1 complex)
  const wh::ckp<double2> zb = wh::ck_as<double2>(zr);
  const wh::ckp<double> sr = wh::ck_make(reinterpret_cast<double*>(smem) + (N + 2), N + 2, wh::WH_CK_LDS_AUX);  // windowed excitation frame / its half spectrum
  const wh::ckp<double2> sb = wh::ck_as<double2>(sr);
  // RUNF > 1: the run's sums, (RUNF - 1) hop + N doubles
  const wh::ckp<double> acc = wh::ck_make(reinterpret_cast<double*>(smem) + 2 * (N + 2), RUNF > 1 ? (RUNF - 1) * q.hop + N : 0, wh::WH_CK_LDS_OTHER);
  if ((int64_t)blockIdx.x >= q.n_runs) return;
  const int64_t hop = q.hop;
  int64_t wlen = 2 * hop - 1;
  const int64_t i0 = (int64_t)blockIdx.x * RUNF + 2;  // frames 2 .. F-2  (synthesisRequiem.py:83)
  const int64_t i1 = i0 + RUNF - 1 < m.nf - 2 ? i0 + RUNF - 1 : m.nf - 2;
  const int64_t a_r = (i0 - 2) * hop + 1;  // 1-based sample of the run's first tap (= the first frame's origin)
  const int64_t W = (RUNF - 1) * hop + N + 1;
  const wh::ckp<double> row = wh::ck_make(rows + q.row_off + (int64_t)blockIdx.x * W, W, wh::WH_CK_OUT);
  const int span = (int)(W - 1);
  if (RUNF > 1) {
    for (int j = threadIdx.x; j < span; j += FT) acc[j] = 0.0;  // (ordered before the first add by the chain's barriers)
  }
  double last = 0.0;  // (thread FT-1: tap N-1 of every frame whose response reaches the utterance's last sample)
  const wh::ckp<const double> eu = wh::ck_make(exc + m.y_off, m.ny, wh::WH_CK_WAVEFORM);
#pragma unroll 1
  for (int64_t i = i0; i <= i1; ++i) {
    const int64_t origin = (i - 1) * hop - (hop - 1);  // 1-based
    // per frame: neither the twiddles nor the window values of one frame are parked in registers for the next (both are
    // the same for every frame, and hoisted out of this loop they cost a wave per SIMD)
    asm volatile("" : "+s"(tw_raw));
    const wh::ckp<const double2> tw_base = wh::ck_make(tw_raw, 2 * WH_MAX_TWIDDLE, wh::WH_CK_TWIDDLE);
    {
      int hop_s = __builtin_amdgcn_readfirstlane((int)hop);  // (uniform by construction; said so for the constraint)
      asm volatile("" : "+s"(hop_s));
      wlen = 2 * (int64_t)hop_s - 1;
    }
    for (int j = WH_TID; j < N; j += FT) {
      double v = 0.0;
      if (j < wlen) {
        int64_t g = origin + j;
        g = g > m.ny ? m.ny : g;
        g = g < 1 ? 1 : g;
        // hanning(wlen+2)[1:-1] — from the launch's table when every utterance has this hop (req_hann_kernel: the same
        // expression, evaluated once instead of per frame: a cospi and a divide per sample were ~5 % of the kernel's
        // instructions), else in place
        const double wv = hann ? hann[j] : 0.5 - 0.5 * cospi(2.0 * (double)(j + 1) / (double)(wlen + 1));
        v = eu[g - 1] * wv;
      }
      sr[j] = v;
    }
    const wh::ckp<const double> sp = wh::ck_make(spectrogram + (m.f_off + (i - 1)) * K, K, wh::WH_CK_IN);
    for (int k = WH_TID; k < K; k += FT) {  // log|S| / 2, Hermitian-mirrored: the input of the chain's first transform
      const double lw = log_call(fabs(sp[k])) / 2;  // (two bins per call: measured, no gain here — 24.0 ms either way)
      zr[k] = lw;
      if (k > 0 && k < N / 2) zr[N - k] = lw;
    }
    wh::sync<FT>();
    wh::rfft_lds<N, FT>(sb, tw_base);
    // minimum-phase spectrum x excitation spectrum (both Hermitian, so is the product), straight into the inverse
    // transform: the fused chain of the pulse responses with the product applied to the register-held bin pairs
    min_phase_response<N, FT>(zb, tw_base, 0.0, [&](int k, double2 e) { return wh::cmul(e, sb[k]); });
    // The run's sums live in LDS and go to the row ONCE, at the end of the run.  (Kept in the row itself — read, add,
    // write back per frame — the kernel is 3 % faster, 1.52 against 1.57 ms at config 4: the accumulator's 10 KB cost two
    // of its eight workgroups per CU; but every frame's 8 KB then travel to HBM and the kernel moves 2.2 GB per 64
    // utterances where this form moves ~1 GB.  Register-held sums spill: 90 VGPRs.)
    const int shift = (int)(origin - a_r);  // (i - i0) * hop: where this frame's tap 0 falls in the run
    for (int mm = WH_TID; mm < N; mm += FT) {
      const double v = origin + mm < m.ny ? zr[mm] / N : 0.0;  // (origin + mm >= 1 always)
      if (RUNF > 1) acc[shift + mm] += v;  // one writer per slot and frame; frames are separated by barriers
      else row[1 + mm] = v;
    }
    if (WH_TID == FT - 1 && origin + (N - 1) >= m.ny) last += zr[N - 1] / N;
    if (RUNF > 1) wh::sync<FT>();  // zr is free for the next frame, this frame's adds are visible to its successor
  }
  if (RUNF > 1) {
    for (int j = threadIdx.x; j < span; j += FT) row[1 + j] = acc[j];  // (zeros behind a short last run's frames)
  }
  if (threadIdx.x == FT - 1) row[0] = last;
}

// y[t] = sum of the rows of req_filter_kernel that cover t, in run order; the last sample: the rows' slot 0.
template <int N, int RUNF>
__global__ __launch_bounds__(256) void req_gather_kernel(const SynUtt* __restrict__ meta, const ReqUtt* __restrict__ rq,
                                                         const double* __restrict__ rows, double* __restrict__ y) {
  const SynUtt m = meta[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m.ny) return;
  const ReqUtt q = rq[blockIdx.y];
  const int64_t adv = RUNF * q.hop;               // samples from one run's first tap to the next run's
  const int64_t W = (RUNF - 1) * q.hop + N + 1;
  const double* ru = rows + q.row_off;
  const int64_t tgt = i + 1;
  double sum = 0.0;
  if (tgt < m.ny) {
    // run r covers the samples a_r .. a_r + W - 2, a_r = r adv + 1
    int64_t r_hi = (tgt - 1) / adv;
    r_hi = r_hi > q.n_runs - 1 ? q.n_runs - 1 : r_hi;
    int64_t r_lo = tgt - (W - 1) <= 0 ? 0 : (tgt - (W - 1) - 1) / adv + 1;  // first r with a_r + W - 2 >= tgt
    for (int64_t r = r_lo; r <= r_hi; ++r) sum += ru[r * W + 1 + (tgt - (r * adv + 1))];  // (rows are written in full)
  } else {
    // frames whose last tap reaches the last sample live in the runs from (ny - N) / adv - 1 on; the others hold 0 there
    int64_t r_lo = (m.ny - N) / adv - 1;
    r_lo = r_lo < 0 ? 0 : r_lo;
    for (int64_t r = r_lo; r < q.n_runs; ++r) sum += ru[r * W];
  }
  y[m.y_off + i] = sum;
}

// frames per run of req_filter_kernel: 8 while the accumulator fits beside the transform buffers at full occupancy
// (N <= 1024: 16 KB + 12.4 KB at a hop of 80), else the frame's own row

template <int N>
int launch_req_filter(wh_ctx* ctx, hipStream_t st, int B, int64_t max_nf, int64_t max_ny, int64_t max_hop, bool runs,
                      const SynUtt* d_meta, const ReqUtt* d_rq, const double* spec, const double* exc, double* rows,
                      double* y, int64_t uniform_hop) {
  constexpr int RUNF = req_runf(N);
  const double* d_hann = nullptr;
#ifndef WH_REQ_HANN_TAB
#define WH_REQ_HANN_TAB 1
#endif
  if (WH_REQ_HANN_TAB && uniform_hop > 0 && 2 * uniform_hop - 1 <= N) {
    const int wlen = (int)(2 * uniform_hop - 1);
    const std::string key = "req.hann:" + std::to_string(wlen);
    auto it = ctx->tables.find(key);
    if (it == ctx->tables.end()) {
      double* d = nullptr;
      WH_CHECK(hipMalloc((void**)&d, sizeof(double) * (size_t)wlen));
      hipLaunchKernelGGL(req_hann_kernel, dim3((unsigned)((wlen + 255) / 256)), dim3(256), 0, st, d, wlen);
      WH_LAUNCH_CHECK("req_hann_kernel");
      ctx->tables[key] = d;
      ctx->table_bytes += sizeof(double) * (size_t)wlen;
      d_hann = d;
    } else {
      d_hann = it->second;
    }
  }
  const size_t lds = sizeof(double2) * 2 * (N / 2 + 1) + 64;  // the chain's buffer and the excitation frame's
  if (max_nf >= 4) {
    wh::KernelTimer _kt(ctx, st, "req_filter_kernel");
    if (runs && RUNF > 1) {
      const size_t lds_run = lds + sizeof(double) * (size_t)((RUNF - 1) * max_hop + N);  // + the run's sums
      if (int rc = wh::allow_lds(&req_filter_kernel<N, RUNF>, lds_run)) return rc;
      hipLaunchKernelGGL((req_filter_kernel<N, RUNF>), dim3((unsigned)((max_nf - 3 + RUNF - 1) / RUNF), B), dim3(ft_syn(N)), lds_run, st, d_meta, d_rq, spec, exc, ctx->d_twiddle, rows, d_hann);
    } else {
      if (int rc = wh::allow_lds(&req_filter_kernel<N, 1>, lds)) return rc;
      hipLaunchKernelGGL((req_filter_kernel<N, 1>), dim3((unsigned)(max_nf - 3), B), dim3(ft_syn(N)), lds, st, d_meta, d_rq, spec, exc, ctx->d_twiddle, rows, d_hann);
    }
  }
  WH_LAUNCH_CHECK("req_filter_kernel");
  {
    wh::KernelTimer _kt(ctx, st, "req_gather_kernel");
    if (runs && RUNF > 1) hipLaunchKernelGGL((req_gather_kernel<N, RUNF>), dim3((unsigned)((max_ny + 255) / 256), B), dim3(256), 0, st, d_meta, d_rq, rows, y);
    else hipLaunchKernelGGL((req_gather_kernel<N, 1>), dim3((unsigned)((max_ny + 255) / 256), B), dim3(256), 0, st, d_meta, d_rq, rows, y);
  }
  WH_LAUNCH_CHECK("req_gather_kernel");
  return 0;
}

}  // namespace

namespace {
int fill_syn_meta(const char* who, const wh_batch* b, const int64_t* h_y_off, const double* h_t0, const double* h_dt,
                  int64_t pulse_cap, const double* noise, const int64_t* h_noise_off, std::vector<SynUtt>& meta,
                  int64_t* max_ny) {
  const int B = b->n_utt;
  meta.resize(B);
  *max_ny = 0;
  for (int u = 0; u < B; ++u) {
    SynUtt& m = meta[u];
    m.f_off = b->h_frame_off[u];
    m.nf = b->h_frame_off[u + 1] - b->h_frame_off[u];
    if (m.nf < 2) return wh::fail_msg(who, "an utterance has fewer than 2 frames");
    m.y_off = h_y_off[u];
    m.ny = h_y_off[u + 1] - h_y_off[u];
    m.p_off = (int64_t)u * pulse_cap;
    m.pcap = pulse_cap;
    m.noise_off = noise ? h_noise_off[u] : 0;
    m.noise_len = noise ? h_noise_off[u + 1] - h_noise_off[u] : -1;
    m.t0 = h_t0[u];
    m.dt = h_dt[u];
    *max_ny = std::max(*max_ny, m.ny);
  }
  return 0;
}
}  // namespace

// Time base of synthesis(): everything that depends on tp / f0 / vuv alone (synthesis.py:118-140, 144-152) — phase
// increments, the exact cumulative phase, pulse positions and fractional shifts, noise offsets, per-pulse frame pairs.
// The results stay in ctx's workspace (ctx->timebase records where) until another call lays the workspace out again.
extern "C" int wh_synthesis_timebase(wh_ctx* ctx, void* stream, const wh_batch* b, const double* tp, const double* f0,
                                     const double* vuv, double fs, const int64_t* h_y_off, const double* h_t0,
                                     const double* h_dt, int64_t pulse_cap, double f0_low_limit) {
  if (!ctx || !b || !tp || !f0 || !vuv || !h_y_off || !h_t0 || !h_dt)
    return wh::fail_msg("wh_synthesis_timebase", "null argument");
  WH_ENTER(ctx);
  if (pulse_cap < 1) return wh::fail_msg("wh_synthesis_timebase", "pulse_cap must be >= 1");
  hipStream_t st = (hipStream_t)stream;
  const int B = b->n_utt;
  std::vector<SynUtt> meta;
  int64_t max_ny = 0;
  if (int rc = fill_syn_meta("wh_synthesis_timebase", b, h_y_off, h_t0, h_dt, pulse_cap, nullptr, nullptr, meta, &max_ny)) return rc;
  const int64_t ny_tot = h_y_off[B];
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t off = 0;
  const size_t o_phase = off; off += al(sizeof(double) * ny_tot);
  const size_t o_vuv = off; off += al((size_t)ny_tot);
  const size_t o_pt = off; off += al(sizeof(double) * B * pulse_cap);
  const size_t o_pi = off; off += al(sizeof(int64_t) * B * pulse_cap);
  const size_t o_ps = off; off += al(sizeof(double) * B * pulse_cap);
  const size_t o_pn = off; off += al(sizeof(int64_t) * B * pulse_cap);
  const size_t o_pc = off; off += al(sizeof(int32_t) * B);
  const size_t o_px = off; off += pulse_scratch_bytes(B, max_ny);
  const size_t o_pb = off; off += al(sizeof(int64_t) * (B + 1));
  const size_t o_rec = off; off += al(sizeof(PulseRec) * B * pulse_cap);
  if (int rc = wh::ws_reserve(ctx, off)) return rc;
  ctx->timebase.valid = false;
  char* ws = reinterpret_cast<char*>(ctx->ws);
  SynUtt* d_meta = nullptr;
  double* d_phase = reinterpret_cast<double*>(ws + o_phase);
  uint8_t* d_vuv = reinterpret_cast<uint8_t*>(ws + o_vuv);
  int64_t* d_pb = reinterpret_cast<int64_t*>(ws + o_pb);
  double* d_pt = reinterpret_cast<double*>(ws + o_pt);
  int64_t* d_pi = reinterpret_cast<int64_t*>(ws + o_pi);
  double* d_ps = reinterpret_cast<double*>(ws + o_ps);
  int64_t* d_pn = reinterpret_cast<int64_t*>(ws + o_pn);
  int32_t* d_pc = reinterpret_cast<int32_t*>(ws + o_pc);
  if (int rc = wh::persistent_upload(ctx, st, "syn.tbmeta", meta, &d_meta)) return rc;
  { wh::KernelTimer _kt(ctx, st, "prep_kernel"); hipLaunchKernelGGL(prep_kernel, dim3((unsigned)((max_ny + 255) / 256), B), dim3(256), 0, st, d_meta, tp, f0, vuv, fs,
                     f0_low_limit, d_phase, d_vuv); }
  WH_LAUNCH_CHECK("prep_kernel");
  if (int rc = exact_cumsum_segments(ctx, st, d_phase, h_y_off, B)) return rc;
  if (int rc = launch_pulses(ctx, st, B, max_ny, d_meta, d_phase, fs, d_pt, d_pi, d_ps, d_pn, d_pc, ws + o_px)) return rc;
  { wh::KernelTimer _kt(ctx, st, "pulse_base_kernel"); hipLaunchKernelGGL(pulse_base_kernel, dim3(1), dim3(64), 0, st, d_pc, B, d_pb); }
  WH_LAUNCH_CHECK("pulse_base_kernel");
  { wh::KernelTimer _kt(ctx, st, "pulse_frames_kernel"); hipLaunchKernelGGL(pulse_frames_kernel, dim3((unsigned)((pulse_cap + 255) / 256), B), dim3(256), 0, st, d_meta, tp, d_pt, d_pi, d_ps, d_pn, d_vuv, d_pc, d_pb,
                     reinterpret_cast<PulseRec*>(ws + o_rec)); }
  WH_LAUNCH_CHECK("pulse_frames_kernel");
  wh_ctx::TimeBase& t = ctx->timebase;
  t.valid = true;
  t.n_utt = B;
  t.pulse_cap = pulse_cap;
  t.ny_tot = ny_tot;
  t.frames = b->total_frames;
  t.o_vuv = o_vuv; t.o_pt = o_pt; t.o_pi = o_pi; t.o_ps = o_ps; t.o_pn = o_pn; t.o_pc = o_pc; t.o_pb = o_pb;
  t.o_rec = o_rec;
  return 0;
}

// The spectral part: one response per pulse of the time base held by `timebase_ctx` (this context or another one of the
// same device), overlap-added into y.  Stream order (or an event the caller waits on) must put it behind that time base.
extern "C" int wh_synthesis_render(wh_ctx* ctx, void* stream, const wh_batch* b, const wh_ctx* timebase_ctx,
                                   const double* tp, const double* spectrogram, const double* aperiodicity, double fs,
                                   int fft_size, const int64_t* h_y_off, const double* h_t0, const double* h_dt,
                                   int64_t pulse_cap, const double* noise, const int64_t* h_noise_off, uint64_t seed,
                                   double* y, int32_t* pulse_count_out) {
  if (!ctx || !b || !timebase_ctx || !tp || !spectrogram || !aperiodicity || !h_y_off || !h_t0 || !h_dt || !y)
    return wh::fail_msg("wh_synthesis_render", "null argument");
  WH_ENTER(ctx);
  if (noise && !h_noise_off) return wh::fail_msg("wh_synthesis_render", "noise given without h_noise_off");
  hipStream_t st = (hipStream_t)stream;
  const int B = b->n_utt;
  const wh_ctx::TimeBase& t = timebase_ctx->timebase;
  if (!t.valid || t.n_utt != B || t.pulse_cap != pulse_cap || t.ny_tot != h_y_off[B] || t.frames != b->total_frames ||
      timebase_ctx->device != ctx->device)
    return wh::fail_msg("wh_synthesis_render", "no matching time base in timebase_ctx (run wh_synthesis_timebase on it "
                                               "with the same batch, lengths and pulse_cap, and nothing else since)");
  std::vector<SynUtt> meta;
  int64_t max_ny = 0;
  if (int rc = fill_syn_meta("wh_synthesis_render", b, h_y_off, h_t0, h_dt, pulse_cap, noise, h_noise_off, meta, &max_ny)) return rc;
  const char* ws = reinterpret_cast<const char*>(timebase_ctx->ws);
  SynUtt* d_meta = nullptr;
  if (int rc = wh::persistent_upload(ctx, st, "syn.meta", meta, &d_meta)) return rc;
  const int64_t* d_pb = reinterpret_cast<const int64_t*>(ws + t.o_pb);
  const int64_t* d_pi = reinterpret_cast<const int64_t*>(ws + t.o_pi);
  const PulseRec* d_rec = reinterpret_cast<const PulseRec*>(ws + t.o_rec);
  const int32_t* d_pc = reinterpret_cast<const int32_t*>(ws + t.o_pc);
  if (noise) {  // (the time base knows nothing about the noise stream: the cover test belongs to the call that is given one)
    wh::KernelTimer _kt(ctx, st, "noise_cover_kernel");
    hipLaunchKernelGGL(noise_cover_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, d_meta, d_rec, d_pb, B, ctx->d_flags);
  }
  WH_LAUNCH_CHECK("noise_cover_kernel");
  std::vector<int64_t> h_ny((size_t)B);
  for (int u = 0; u < B; ++u) h_ny[u] = meta[u].ny;
  int rc;
  switch (fft_size) {
    case 512: rc = launch_resp<512>(ctx, st, B, pulse_cap, max_ny, h_ny, d_meta, tp, spectrogram, aperiodicity, fs, d_rec, d_pb, d_pi, d_pc, noise, seed, y); break;
    case 1024: rc = launch_resp<1024>(ctx, st, B, pulse_cap, max_ny, h_ny, d_meta, tp, spectrogram, aperiodicity, fs, d_rec, d_pb, d_pi, d_pc, noise, seed, y); break;
    case 2048: rc = launch_resp<2048>(ctx, st, B, pulse_cap, max_ny, h_ny, d_meta, tp, spectrogram, aperiodicity, fs, d_rec, d_pb, d_pi, d_pc, noise, seed, y); break;
    case 4096: rc = launch_resp<4096>(ctx, st, B, pulse_cap, max_ny, h_ny, d_meta, tp, spectrogram, aperiodicity, fs, d_rec, d_pb, d_pi, d_pc, noise, seed, y); break;
    default: return wh::fail_msg("wh_synthesis_render", "fft_size must be a power of two in [512, 4096]");
  }
  if (rc) return rc;
  if (pulse_count_out) WH_CHECK(hipMemcpyAsync(pulse_count_out, d_pc, sizeof(int32_t) * B, hipMemcpyDeviceToDevice, st));
  return 0;
}

extern "C" int wh_synthesis(wh_ctx* ctx, void* stream, const wh_batch* b, const double* tp, const double* f0,
                            const double* vuv, const double* spectrogram, const double* aperiodicity, double fs,
                            int fft_size, const int64_t* h_y_off, const double* h_t0, const double* h_dt,
                            int64_t pulse_cap, const double* noise, const int64_t* h_noise_off, uint64_t seed, double* y,
                            int32_t* pulse_count_out) {
  if (!ctx || !b || !tp || !f0 || !vuv || !spectrogram || !aperiodicity || !h_y_off || !h_t0 || !h_dt || !y)
    return wh::fail_msg("wh_synthesis", "null argument");
  if (noise && !h_noise_off) return wh::fail_msg("wh_synthesis", "noise given without h_noise_off");
  if (fft_size != 512 && fft_size != 1024 && fft_size != 2048 && fft_size != 4096)
    return wh::fail_msg("wh_synthesis", "fft_size must be a power of two in [512, 4096]");
  if (int rc = wh_synthesis_timebase(ctx, stream, b, tp, f0, vuv, fs, h_y_off, h_t0, h_dt, pulse_cap, 0.0)) return rc;
  return wh_synthesis_render(ctx, stream, b, ctx, tp, spectrogram, aperiodicity, fs, fft_size, h_y_off, h_t0, h_dt, pulse_cap,
                             noise, h_noise_off, seed, y, pulse_count_out);
}

// Samples [q0, q0 + n) of the normal stream that the device-noise decode (noise == NULL) reads for utterance `utt`
// under `seed`: pulse i of that utterance consumes samples noff_i .. noff_i + max(3, noise_size_i) of it, exactly as it
// would consume a host-supplied stream.  Feeding the dump back as `noise` therefore reproduces the seeded decode.
extern "C" int wh_philox_normals(wh_ctx* ctx, void* stream, uint64_t seed, int utt, int64_t q0, int64_t n, double* out) {
  if (!ctx || !out || utt < 0 || q0 < 0 || n < 0) return wh::fail_msg("wh_philox_normals", "bad argument");
  WH_ENTER(ctx);
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  { wh::KernelTimer _kt(ctx, st, "philox_dump_kernel"); hipLaunchKernelGGL(philox_dump_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, seed, (int32_t)utt, q0, n, out); }
  WH_LAUNCH_CHECK("philox_dump_kernel");
  return 0;
}

// ---- peak normalisation of decode(): y /= max|y| where it exceeds 1 (world/main.py:209-212), per utterance ----
// Non-negative doubles order like their bit patterns, so the maximum is an integer atomicMax.
__global__ __launch_bounds__(256) void peak_max_kernel(const double* __restrict__ y, const int64_t* __restrict__ off,
                                                       unsigned long long* __restrict__ peak_bits) {
  __shared__ unsigned long long wmax[4];
  const int u = blockIdx.y;
  const int64_t s = off[u], n = off[u + 1] - s;
  unsigned long long m = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(y[s + i]));
    m = b > m ? b : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long v = (unsigned long long)__shfl_xor((long long)m, o, 64);
    m = v > m ? v : m;
  }
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) m = wmax[w] > m ? wmax[w] : m;
    atomicMax(peak_bits + u, m);
  }
}
__global__ __launch_bounds__(256) void peak_scale_kernel(double* __restrict__ y, const int64_t* __restrict__ off,
                                                         const unsigned long long* __restrict__ peak_bits) {
  const int u = blockIdx.y;
  const double peak = __longlong_as_double((long long)peak_bits[u]);
  if (!(peak > 1.0)) return;  // world/main.py:210: only when the maximum exceeds 1 (a NaN peak leaves y alone)
  const int64_t s = off[u], n = off[u + 1] - s;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[s + i] = y[s + i] / peak;
}

extern "C" int wh_peak_normalise(wh_ctx* ctx, void* stream, double* y, const int64_t* h_y_off, int n_utt) {
  if (!ctx || !y || !h_y_off || n_utt < 0) return wh::fail_msg("wh_peak_normalise", "bad argument");
  WH_ENTER(ctx);
  if (n_utt == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  std::vector<int64_t> off(h_y_off, h_y_off + n_utt + 1);
  int64_t* d_off = nullptr;
  if (int rc = wh::persistent_upload(ctx, st, "peak.off", off, &d_off)) return rc;
  int64_t max_n = 0;
  for (int u = 0; u < n_utt; ++u) max_n = std::max(max_n, off[u + 1] - off[u]);
  if (int rc = wh::ws_reserve(ctx, sizeof(unsigned long long) * (size_t)n_utt)) return rc;
  unsigned long long* d_peak = reinterpret_cast<unsigned long long*>(ctx->ws);
  WH_CHECK(hipMemsetAsync(d_peak, 0, sizeof(unsigned long long) * (size_t)n_utt, st));
  const unsigned gx = (unsigned)std::min<int64_t>(64, (max_n + 4 * 256 - 1) / (4 * 256) + 1);
  { wh::KernelTimer _kt(ctx, st, "peak_max_kernel"); hipLaunchKernelGGL(peak_max_kernel, dim3(gx, n_utt), dim3(256), 0, st, y, d_off, d_peak); }
  WH_LAUNCH_CHECK("peak_max_kernel");
  { wh::KernelTimer _kt(ctx, st, "peak_scale_kernel"); hipLaunchKernelGGL(peak_scale_kernel, dim3(gx, n_utt), dim3(256), 0, st, y, d_off, d_peak); }
  WH_LAUNCH_CHECK("peak_scale_kernel");
  return 0;
}

#ifdef WH_RESP_STAGE_TIMER
extern "C" int wh_debug_resp_stages(unsigned long long* out8, int reset) {
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_resp_stage), sizeof(unsigned long long) * 8);
  if (reset) {
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_resp_stage), z, sizeof(z));
  }
  return 0;
}
#endif

// In-place exact sequential cumulative sum of n_seg independent segments of NON-NEGATIVE doubles
// (h_off[n_seg + 1] element offsets into d_data) — the routine behind the phase accumulator, exposed so that its
// bit-for-bit agreement with np.cumsum can be tested directly.
extern "C" int wh_cumsum_exact(wh_ctx* ctx, void* stream, double* d_data, const int64_t* h_off, int n_seg) {
  if (!ctx || !d_data || !h_off || n_seg < 0) return wh::fail_msg("wh_cumsum_exact", "bad argument");
  WH_ENTER(ctx);
  if (n_seg == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  return exact_cumsum_segments(ctx, st, d_data, h_off, n_seg);
}

// Pulse bookkeeping only (no responses): per-utterance pulse count and total noise draws
// sum(max(3, noise_size)) — lets a host draw EXACTLY the reference's number of randn samples.
extern "C" int wh_synthesis_plan(wh_ctx* ctx, void* stream, const wh_batch* b, const double* tp, const double* f0,
                                 const double* vuv, double fs, const int64_t* h_y_off, const double* h_t0,
                                 const double* h_dt, int64_t pulse_cap, int32_t* h_pulse_count,
                                 int64_t* h_noise_total) {
  if (!ctx || !b || !tp || !f0 || !vuv || !h_y_off || !h_t0 || !h_dt || !h_pulse_count || !h_noise_total)
    return wh::fail_msg("wh_synthesis_plan", "null argument");
  WH_ENTER(ctx);
  hipStream_t st = (hipStream_t)stream;
  const int B = b->n_utt;
  std::vector<SynUtt> meta(B);
  int64_t max_ny = 0;
  for (int u = 0; u < B; ++u) {
    SynUtt& m = meta[u];
    m.f_off = b->h_frame_off[u];
    m.nf = b->h_frame_off[u + 1] - b->h_frame_off[u];
    if (m.nf < 2) return wh::fail_msg("wh_synthesis_plan", "an utterance has fewer than 2 frames");
    m.y_off = h_y_off[u];
    m.ny = h_y_off[u + 1] - h_y_off[u];
    m.p_off = (int64_t)u * pulse_cap;
    m.pcap = pulse_cap;
    m.noise_off = 0;
    m.noise_len = -1;
    m.t0 = h_t0[u];
    m.dt = h_dt[u];
    max_ny = std::max(max_ny, m.ny);
  }
  const int64_t ny_tot = h_y_off[B];
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t off = 0;
  const size_t o_phase = off; off += al(sizeof(double) * ny_tot);
  const size_t o_vuv = off; off += al((size_t)ny_tot);
  const size_t o_pt = off; off += al(sizeof(double) * B * pulse_cap);
  const size_t o_pi = off; off += al(sizeof(int64_t) * B * pulse_cap);
  const size_t o_ps = off; off += al(sizeof(double) * B * pulse_cap);
  const size_t o_pn = off; off += al(sizeof(int64_t) * B * pulse_cap);
  const size_t o_pc = off; off += al(sizeof(int32_t) * B);
  const size_t o_px = off; off += pulse_scratch_bytes(B, max_ny);
  if (int rc = wh::ws_reserve(ctx, off)) return rc;
  char* ws = reinterpret_cast<char*>(ctx->ws);
  SynUtt* d_meta = nullptr;
  double* d_phase = reinterpret_cast<double*>(ws + o_phase);
  uint8_t* d_vuv = reinterpret_cast<uint8_t*>(ws + o_vuv);
  int64_t* d_pi = reinterpret_cast<int64_t*>(ws + o_pi);
  int64_t* d_pn = reinterpret_cast<int64_t*>(ws + o_pn);
  int32_t* d_pc = reinterpret_cast<int32_t*>(ws + o_pc);
  if (int rc = wh::persistent_upload(ctx, st, "syn.meta", meta, &d_meta)) return rc;
  { wh::KernelTimer _kt(ctx, st, "prep_kernel"); hipLaunchKernelGGL(prep_kernel, dim3((unsigned)((max_ny + 255) / 256), B), dim3(256), 0, st, d_meta, tp, f0, vuv, fs,
                     0.0, d_phase, d_vuv); }
  WH_LAUNCH_CHECK("prep_kernel");
  ctx->timebase.valid = false;  // (this call lays the workspace out its own way)
  if (int rc = exact_cumsum_segments(ctx, st, d_phase, h_y_off, B)) return rc;
  if (int rc = launch_pulses(ctx, st, B, max_ny, d_meta, d_phase, fs, reinterpret_cast<double*>(ws + o_pt), d_pi,
                             reinterpret_cast<double*>(ws + o_ps), d_pn, d_pc, ws + o_px)) return rc;
  WH_CHECK(hipMemcpyAsync(h_pulse_count, d_pc, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
  WH_CHECK(hipStreamSynchronize(st));
  // total draws = noff[last] + max(3, 0)
  for (int u = 0; u < B; ++u) {
    int64_t total = 0;
    if (h_pulse_count[u] > 0) {
      int64_t last_off = 0;
      WH_CHECK(hipMemcpy(&last_off, d_pn + (int64_t)u * pulse_cap + h_pulse_count[u] - 1, sizeof(int64_t),
                         hipMemcpyDeviceToHost));
      total = last_off + 3;
    }
    h_noise_total[u] = total;
  }
  return 0;
}


extern "C" int wh_synthesis_requiem(wh_ctx* ctx, void* stream, const wh_batch* b, const double* tp, const double* f0,
                                    const double* vuv, const double* spectrogram, const double* band_aperiodicity,
                                    double fs, int fft_size, const int64_t* h_y_off, const double* h_t0, const double* h_dt,
                                    const int64_t* h_hop, int64_t pulse_cap, const double* pulse_seed, int pulse_fft,
                                    const double* noise_seed, int64_t noise_len, int n_bands, const int64_t* h_cursor,
                                    double* y) {
  if (!ctx || !b || !tp || !f0 || !vuv || !spectrogram || !band_aperiodicity || !h_y_off || !h_t0 || !h_dt || !h_hop ||
      !pulse_seed || !noise_seed || !h_cursor || !y)
    return wh::fail_msg("wh_synthesis_requiem", "null argument");
  WH_ENTER(ctx);
  if (n_bands < 1 || n_bands > 8) return wh::fail_msg("wh_synthesis_requiem", "n_bands must be in [1, 8]");
  if (pulse_cap < 1 || noise_len < 1) return wh::fail_msg("wh_synthesis_requiem", "bad pulse_cap / noise_len");
  hipStream_t st = (hipStream_t)stream;
  const int B = b->n_utt;
  std::vector<SynUtt> meta(B);
  std::vector<ReqUtt> rq(B);
  int64_t max_ny = 0, max_nf = 0, max_hop = 0, uniform_hop = 0;  // (uniform_hop: the hop every utterance has, or 0)
  for (int u = 0; u < B; ++u) {
    SynUtt& m = meta[u];
    m.f_off = b->h_frame_off[u];
    m.nf = b->h_frame_off[u + 1] - b->h_frame_off[u];
    if (m.nf < 2) return wh::fail_msg("wh_synthesis_requiem", "an utterance has fewer than 2 frames");
    m.y_off = h_y_off[u];
    m.ny = h_y_off[u + 1] - h_y_off[u];
    m.p_off = (int64_t)u * pulse_cap;
    m.pcap = pulse_cap;
    m.noise_off = 0;
    m.noise_len = -1;
    m.t0 = h_t0[u];
    m.dt = h_dt[u];
    rq[u].hop = h_hop[u];
    if (rq[u].hop < 1) return wh::fail_msg("wh_synthesis_requiem", "frame hop below one sample");
    max_hop = std::max(max_hop, rq[u].hop);
    if (u == 0) uniform_hop = rq[u].hop;
    else if (rq[u].hop != uniform_hop) uniform_hop = 0;
    for (int k = 0; k < 8; ++k) rq[u].cursor[k] = k < n_bands ? ((h_cursor[(int64_t)u * n_bands + k] % noise_len) + noise_len) % noise_len : 0;
    max_ny = std::max(max_ny, m.ny);
    max_nf = std::max(max_nf, m.nf);
  }
  const int64_t ny_tot = h_y_off[B];
  const int64_t F = b->total_frames;
  // overlap-add rows of req_filter_kernel: runs of frames (one row per frame beyond N = 1024 and for long hops)
  int runf = 1;
  switch (fft_size) {
    case 512: runf = req_runf(512); break;
    case 1024: runf = req_runf(1024); break;
    case 2048: runf = req_runf(2048); break;
    case 4096: runf = req_runf(4096); break;
    default: return wh::fail_msg("wh_synthesis_requiem", "fft_size must be a power of two in [512, 4096]");
  }
  // (the run's sums are an LDS accumulator of (RUNF - 1) hop + N doubles: at most 2 N, i.e. 16 KB at N = 1024)
  const bool runs = runf > 1 && (runf - 1) * max_hop <= (int64_t)fft_size;
  if (!runs) runf = 1;
  int64_t rows_tot = 0;
  for (int u = 0; u < B; ++u) {
    const int64_t frames = meta[u].nf >= 4 ? meta[u].nf - 3 : 0;  // frames 2 .. F-2
    rq[u].n_runs = (frames + runf - 1) / runf;
    rq[u].row_off = rows_tot;
    rows_tot += rq[u].n_runs * ((runf - 1) * rq[u].hop + fft_size + 1);
  }
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t off = 0;
  const size_t o_phase = off; off += al(sizeof(double) * ny_tot);
  const size_t o_vuv = off; off += al((size_t)ny_tot);
  const size_t o_pt = off; off += al(sizeof(double) * B * pulse_cap);
  const size_t o_pi = off; off += al(sizeof(int64_t) * B * pulse_cap);
  const size_t o_ps = off; off += al(sizeof(double) * B * pulse_cap);
  const size_t o_pn = off; off += al(sizeof(int64_t) * B * pulse_cap);
  const size_t o_pc = off; off += al(sizeof(int32_t) * B);
  const size_t o_px = off; off += pulse_scratch_bytes(B, max_ny);
  const size_t o_lin = off; off += al(sizeof(double) * F * n_bands);
  const size_t o_exc = off; off += al(sizeof(double) * ny_tot);
  const size_t o_pw = off; off += al(sizeof(double) * B * pulse_cap * n_bands);  // band weights per pulse (the gains reuse o_pt)
  const size_t o_rows = off; off += al(sizeof(double) * (size_t)(rows_tot + 8));
  if (int rc = wh::ws_reserve(ctx, off)) return rc;
  char* ws = reinterpret_cast<char*>(ctx->ws);
  SynUtt* d_meta = nullptr;
  ReqUtt* d_rq = nullptr;
  double* d_phase = reinterpret_cast<double*>(ws + o_phase);
  uint8_t* d_vuv = reinterpret_cast<uint8_t*>(ws + o_vuv);
  double* d_pt = reinterpret_cast<double*>(ws + o_pt);
  int64_t* d_pi = reinterpret_cast<int64_t*>(ws + o_pi);
  double* d_ps = reinterpret_cast<double*>(ws + o_ps);
  int64_t* d_pn = reinterpret_cast<int64_t*>(ws + o_pn);
  int32_t* d_pc = reinterpret_cast<int32_t*>(ws + o_pc);
  double* d_lin = reinterpret_cast<double*>(ws + o_lin);
  double* d_exc = reinterpret_cast<double*>(ws + o_exc);
  double* d_pw = reinterpret_cast<double*>(ws + o_pw);
  if (int rc = wh::persistent_upload(ctx, st, "syn.meta", meta, &d_meta)) return rc;
  if (int rc = wh::persistent_upload(ctx, st, "syn.req", rq, &d_rq)) return rc;
  double* d_rows = reinterpret_cast<double*>(ws + o_rows);
  { wh::KernelTimer _kt(ctx, st, "prep_kernel"); hipLaunchKernelGGL(prep_kernel, dim3((unsigned)((max_ny + 255) / 256), B), dim3(256), 0, st, d_meta, tp, f0, vuv, fs, 0.0, d_phase, d_vuv); }
  WH_LAUNCH_CHECK("prep_kernel");
  if (int rc = exact_cumsum_segments(ctx, st, d_phase, h_y_off, B)) return rc;
  if (int rc = launch_pulses(ctx, st, B, max_ny, d_meta, d_phase, fs, d_pt, d_pi, d_ps, d_pn, d_pc, ws + o_px)) return rc;
  { wh::KernelTimer _kt(ctx, st, "req_linap_kernel"); hipLaunchKernelGGL(req_linap_kernel, dim3((unsigned)((F * n_bands + 255) / 256)), dim3(256), 0, st, band_aperiodicity, F * n_bands, d_lin); }
  WH_LAUNCH_CHECK("req_linap_kernel");
  { wh::KernelTimer _kt(ctx, st, "req_pulse_weights_kernel"); hipLaunchKernelGGL(req_pulse_weights_kernel, dim3((unsigned)((pulse_cap + 255) / 256), B), dim3(256), 0, st, d_meta, tp, d_lin, n_bands, d_pi, d_pc, d_vuv, d_pt, d_pw); }
  WH_LAUNCH_CHECK("req_pulse_weights_kernel");
  { wh::KernelTimer _kt(ctx, st, "req_excite_kernel"); hipLaunchKernelGGL(req_excite_kernel, dim3((unsigned)((max_ny + 255) / 256), B), dim3(256), 0, st, d_meta, d_rq, tp, d_lin, n_bands, noise_seed, noise_len, pulse_seed, pulse_fft, d_pi, d_pc, d_pt, d_pw, d_exc); }
  WH_LAUNCH_CHECK("req_excite_kernel");
  switch (fft_size) {
    case 512: return launch_req_filter<512>(ctx, st, B, max_nf, max_ny, max_hop, runs, d_meta, d_rq, spectrogram, d_exc, d_rows, y, uniform_hop);
    case 1024: return launch_req_filter<1024>(ctx, st, B, max_nf, max_ny, max_hop, runs, d_meta, d_rq, spectrogram, d_exc, d_rows, y, uniform_hop);
    case 2048: return launch_req_filter<2048>(ctx, st, B, max_nf, max_ny, max_hop, runs, d_meta, d_rq, spectrogram, d_exc, d_rows, y, uniform_hop);
    case 4096: return launch_req_filter<4096>(ctx, st, B, max_nf, max_ny, max_hop, runs, d_meta, d_rq, spectrogram, d_exc, d_rows, y, uniform_hop);
    default: return wh::fail_msg("wh_synthesis_requiem", "fft_size must be a power of two in [512, 4096]");
  }
}
