// D4C / D4C-Requiem band aperiodicity.
//   love_train_kernel : VUV gate, one workgroup per frame, one FFT          (world/d4c.py:68-88)
//   d4c_kernel        : per gated frame, everything in LDS: two Blackman frames packed as
//                       x + j*n*x into ONE complex FFT each (spectrum and time-weighted spectrum
//                       are separated by Hermitian symmetry), Hann frame FFT, three sliding-window
//                       smoothings, then per 3 kHz band: Nuttall-windowed group delay → FFT →
//                       rank-select of the smallest-energy bins → energy ratio      (world/d4c.py:114-209)
// Replaces d4c() (world/d4c.py:10-64) and d4cRequiem() (world/d4cRequiem.py:9-44).
#include <map>

#include <hip/hip_runtime.h>
// The thread index as the FFT / reduction helpers of wh_device.h see it: an opaque read.  d4c_kernel runs four
// transforms and four windows per frame through the same helpers; with the plain threadIdx.x the compiler recognises
// the per-thread LDS addresses (eight swizzled store addresses and eight load addresses per radix-8 pass), twiddle
// offsets and index-to-double conversions as common subexpressions of all of them, computes them once and parks them
// in registers for the whole kernel (193 VGPRs wanted where four workgroups per CU allow 128).  Re-deriving them per
// use costs a few integer instructions.
// Round-6 experiment (tools/build_variants.py): WH_D4C_PAIR frames per workgroup, one after the other;
// WH_D4C_HOIST=1 lifts the three fences (plain thread index, fresh_table, stage_fence) so that the compiler may keep the
// per-thread addresses, twiddles and index conversions it wants in registers and share them across the frames of a
// workgroup and all their transforms; WH_D4C_PAIR_MINW sets the register budget (2: 256 VGPRs).  Shipped: 1 / 0.
#ifndef WH_D4C_PAIR
#define WH_D4C_PAIR 1
#endif
#ifndef WH_D4C_HOIST
#define WH_D4C_HOIST 0
#endif
#ifndef WH_D4C_PAIR_MINW
#define WH_D4C_PAIR_MINW 2
#endif
__device__ __forceinline__ int wh_opaque_tid() {
  int t = threadIdx.x;
#if !WH_D4C_HOIST
  asm volatile("" : "+v"(t));
#endif
  return t;
}
#define WH_TID wh_opaque_tid()
#include "wh_host.h"
#include "wh_spectral.h"

#ifndef WH_D4C_ABLATE
#define WH_D4C_ABLATE 0
#endif
// FFT radix caps of d4c_kernel by transform length.  At 128 VGPRs (four workgroups per CU) the radix-8 plan fits
// N <= 1024 without spilling; at N = 2048 / 4096 it spills ~23 registers — still 2 % faster, but the spills are HBM
// traffic (1.86 GB per launch where the kernel's compulsory bytes are 0.61 GB), so those lengths keep radix 4.
#ifndef WH_D4C_KEEP_W
#define WH_D4C_KEEP_W 4  // rows of a register-fed window whose window values survive from the first walk to the second
#endif
#ifndef WH_D4C_GATHER_UNCOND
#define WH_D4C_GATHER_UNCOND 1
#endif
#ifndef WH_D4C_REGFED
#define WH_D4C_REGFED 1  // windows keep their samples in registers and feed the first (radix-8) FFT pass directly
#endif
#ifndef WH_D4C_RMAXR
#define WH_D4C_RMAXR 0  // 0: by length (below); 4 / 8: forced
#endif
#ifndef WH_D4C_MAXR
#define WH_D4C_MAXR 0
#endif
constexpr int d4c_maxr(int n) { return WH_D4C_MAXR ? WH_D4C_MAXR : ((n <= 1024 || WH_D4C_REGFED) ? 8 : 4); }
constexpr int d4c_rmaxr(int n) { return WH_D4C_RMAXR ? WH_D4C_RMAXR : ((n <= 1024 || WH_D4C_REGFED) ? 8 : 4); }
#ifndef WH_LOVE_REGWIN
#define WH_LOVE_REGWIN 1
#endif
#ifndef WH_LOVE_MAXR
#define WH_LOVE_MAXR 8
#endif
#ifndef WH_D4C_MINBLK
#define WH_D4C_MINBLK 4
#endif
#ifndef WH_D4C_WIN_UNROLL
#define WH_D4C_WIN_UNROLL 4
#endif
#ifndef WH_D4C_CENT_LOOP
#define WH_D4C_CENT_LOOP 0
#endif
// -DWH_D4C_STAGE_TIMER: thread 0 of every workgroup adds the shader-clock cycles between stage boundaries to
// g_d4c_stage[] (read with wh_debug_d4c_stages, tools/d4c_stage_timer.py) — the per-stage latencies quoted in DESIGN.md.
#ifdef WH_D4C_STAGE_TIMER
__device__ unsigned long long g_d4c_stage[16];
// (the running time stamp of a workgroup lives in global memory so that device functions can mark stages too)
__device__ unsigned long long g_d4c_t0[1 << 20];
#define STAGE_TIMER_BEGIN { if (threadIdx.x == 0) g_d4c_t0[blockIdx.x & ((1 << 20) - 1)] = __builtin_readcyclecounter(); }
#define STAGE_MARK(i) { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long _t = __builtin_readcyclecounter(); atomicAdd(&g_d4c_stage[i], _t - g_d4c_t0[blockIdx.x & ((1 << 20) - 1)]); g_d4c_t0[blockIdx.x & ((1 << 20) - 1)] = _t; } }
#else
#define STAGE_TIMER_BEGIN
#define STAGE_MARK(i)
#endif
namespace {

// The kernel runs four transforms through the same twiddle table.  Left alone, the compiler recognises the repeated
// read-only loads and address arithmetic, computes them once and keeps them in registers across the whole kernel
// (203 VGPRs); passing the table pointer through an empty asm before each transform makes every instance re-derive
// what it needs from L1/L2-resident data.
__device__ __forceinline__ const double2* fresh_table(const double2* p) {
#if !WH_D4C_HOIST
  asm volatile("" : "+s"(p));
#endif
  return p;
}
#if WH_BOUNDS
__device__ __forceinline__ wh::ckp<const double2> fresh_table(wh::ckp<const double2> p) {
#if !WH_D4C_HOIST
  asm volatile("" : "+s"(p.p));
#endif
  return p;
}
#endif
// Stage fence for a per-frame scalar: everything a stage derives from the returned value (window phase, sample
// addresses, rotation constants ...) can only be computed after this point, i.e. the compiler cannot start the next
// stage's loads and transcendental set-up underneath the current stage's transform (which it does otherwise, and
// pays for with ~60 VGPRs of values parked across the FFT).
__device__ __forceinline__ double stage_fence(double v) {
#if !WH_D4C_HOIST
  asm volatile("" : "+v"(v));
#endif
  return v;
}

#ifndef WH_FT_D4C
#define WH_FT_D4C 256
#endif
// Threads cooperating on one frame: 256 up to N = 2048; 512 at N = 4096 (48 kHz), where the 96 KB of LDS per frame
// leave one workgroup per CU and the thread count is the only occupancy there is.
// At N = 1024 (D4C-Requiem at 16 kHz) 128 threads make N = 8 * FT, the shape of the register-fed windows.
#ifndef WH_FT_D4C_1024
#define WH_FT_D4C_1024 (WH_D4C_REGFED ? WH_FT_D4C / 2 : WH_FT_D4C)
#endif
#ifndef WH_FT_D4C_4096
#define WH_FT_D4C_4096 (2 * WH_FT_D4C)  // (1024 threads, staged windows: 62.9 ms at config 5 — 50.8 compiled for 8 waves
                                        // per SIMD with 164 spilled registers — against 31.8: occupancy is what this
                                        // instance lacks, but LDS (66 KB) and registers (111) both stop it at 4 waves)
#endif
constexpr int ft_of(int n) { return n >= 8192 ? 2 * WH_FT_D4C : n >= 4096 ? WH_FT_D4C_4096 : (n == 1024 ? WH_FT_D4C_1024 : WH_FT_D4C); }
// Waves per SIMD the register allocation must leave room for (HIP's second __launch_bounds__ argument is
// MIN_WAVES_PER_EU).  LDS per frame is the 2N-double transform buffer (33 KB at N = 2048: 4 workgroups of 4 waves
// per CU, 66 KB at N = 4096: 2 workgroups of 8 waves), i.e. 4 waves per SIMD either way -> 128 VGPRs.
#ifndef WH_D4C_MINBLK4096
#define WH_D4C_MINBLK4096 4
#endif
#ifndef WH_D4C_MINBLK1024
// N <= 1024 (D4C-Requiem at 16 kHz).  Staged form: 96 VGPRs, five 4-wave workgroups per CU (17 KB of LDS each).  Register-
// fed form (two waves per frame): the radix-8 butterflies need the 128-register budget; eight workgroups per CU.
#define WH_D4C_MINBLK1024 (WH_D4C_REGFED ? 4 : 5)
#endif
// (N = 8192, 96 kHz material: 131 KB of LDS per frame -> one 512-thread workgroup per CU, 256 registers per thread)
constexpr int minblk_of(int n) { return n >= 8192 ? 1 : n >= 4096 ? WH_D4C_MINBLK4096 : (n == 1024 ? WH_D4C_MINBLK1024 : (n < 1024 ? 5 : WH_D4C_MINBLK)); }

// Windowed, DC-removed pitch-synchronous frame (world/d4c.py:92-110).  emit(j, value) is called for every sample
// j = tid + q*FT < N (zero beyond the window; rows longer than N are cropped like np.fft.fft(x, n), Q7) — the callers
// store straight into the transform buffer, so no per-thread output array exists.  ENERGY: the values are divided by
// the frame's norm sqrt(sum(wave^2)) over the FULL window (d4c.py:147).  BLACKMAN selects window type 2, else Hann.
//
// Two walks, one reduction, no per-thread arrays: the first walk accumulates the sums, the second fetches the
// samples again (L1/L2 hits) and emits the DC-removed values.  Keeping x*w and w in registers between the walks (the
// first version) made this routine the kernel's register peak (~95 VGPRs on its own).  The window is re-derived
// cheaply in both walks because cos(pi*f0*t_j) advances from j to j + FT by a fixed rotation (one sincospi per
// thread and 4 flops per sample instead of one cospi per sample).  The energy of the DC-removed frame comes out of
// the same block reduction as the two means:
//   sum (xw - w*dc)^2 = sum (xw)^2 - 2*dc*sum (xw*w) + dc^2 * sum w^2
// (three more partial sums, no second pass over the data and no second pair of barriers).  The expansion loses
// log10((DC/AC)^2) digits to cancellation — nothing for speech-like input (DC << AC), and still 1e-10 relative for a
// DC offset 1000x the signal.
// Per-frame set-up of one analysis window, evaluated ONCE per workgroup by a single lane (win_setup) and read back by
// every thread through LDS broadcasts: window length, clamped sample range, rotation constants and the start phase are
// the same for all threads, yet as straight-line code each of the four waves spent ~280 instructions per window on
// them (five FP64 divides, two sincospi, the 64-bit clamps) — a fifth of this kernel's instruction stream.
// The per-thread start phase is base * E[tid]: E = exp(i*pi*delta*tid) costs one sincospi per thread and is shared by
// the windows that have the same f0 and length (the Hann frame and the two centroid frames).
constexpr int kWinTab = 16;  // doubles per window in the table
struct WinSetup {
  int hwl, L, rlo, rhi;
  long long centre;
  double rot_s, rot_c, base_s, base_c, delta, inv_span, phase, cf;
};
__device__ __forceinline__ void win_setup(wh::ckp<double> tab, long long xn, double fs, double cf, double pos, double half_length,
                                          int ft) {
  const int hwl = (int)(half_length * fs / cf + 0.5);
  const long long centre = wh::frame_centre(pos, fs);
  const double phase = (pos * fs - (double)(long long)(pos * fs + 0.5)) / fs;
  // per-frame constants are inverted once and multiplied in: an FP64 divide is ~12 instructions with a long
  // dependency chain, and the per-sample ones were a fifth of this kernel's instruction count (results move by an ulp)
  const double inv_span = 1.0 / fs / half_length;
  // sample index relative to the centre, clamped to the utterance (d4c.py:98): x[centre - 1 + rel]
  const long long rel_min = 1 - centre, rel_max = xn - centre;
  const int rlo = (int)(rel_min < -(1 << 30) ? -(1 << 30) : (rel_min > (1 << 30) ? (1 << 30) : rel_min));
  const int rhi = (int)(rel_max > (1 << 30) ? (1 << 30) : (rel_max < -(1 << 30) ? -(1 << 30) : rel_max));
  double rot_s, rot_c, base_s, base_c;
  sincospi((double)ft * inv_span * cf, &rot_s, &rot_c);                 // rotation by FT samples
  sincospi(((double)(0 - hwl) * inv_span + phase) * cf, &base_s, &base_c);  // phase of sample 0
  tab[0] = (double)hwl;
  tab[1] = (double)(2 * hwl + 1);
  tab[2] = (double)rlo;
  tab[3] = (double)rhi;
  tab[4] = (double)centre;  // |centre| < 2^53
  tab[5] = rot_s;
  tab[6] = rot_c;
  tab[7] = base_s;
  tab[8] = base_c;
  tab[9] = inv_span * cf;
  tab[10] = inv_span;
  tab[11] = phase;
  tab[12] = cf;
}
__device__ __forceinline__ WinSetup win_load(wh::ckp<const double> tab) {
  WinSetup w;
  w.hwl = (int)tab[0];
  w.L = (int)tab[1];
  w.rlo = (int)tab[2];
  w.rhi = (int)tab[3];
  w.centre = (long long)tab[4];
  w.rot_s = tab[5];
  w.rot_c = tab[6];
  w.base_s = tab[7];
  w.base_c = tab[8];
  w.delta = tab[9];
  w.inv_span = tab[10];
  w.phase = tab[11];
  w.cf = tab[12];
  return w;
}
// E[tid] = exp(i*pi*delta*tid) as (sin, cos)
__device__ __forceinline__ double2 win_thread_phase(double delta) {
  double s, c;
  sincospi(delta * (double)threadIdx.x, &s, &c);
  return make_double2(s, c);
}

// slot / STRIDE: sample j is parked at slot[j * STRIDE] (LDS) between the gather and the second walk — the place emit()
// overwrites with the final value, so the park costs no extra memory.
template <bool BLACKMAN, int N, bool ENERGY, int STRIDE, int FT_ = 0, class Emit>
__device__ __forceinline__ void d4c_window(wh::ckp<const double> WH_RESTRICT xu, wh::ckp<const double> tab, double2 e_tid,
                                           wh::ckp<double> scratch, wh::ckp<double> slot, Emit emit) {
  constexpr int FT = FT_ ? FT_ : ft_of(N);
  constexpr int Q = N / FT;
  const WinSetup ws = win_load(tab);
  const int hwl = ws.hwl, L = ws.L, rlo = ws.rlo, rhi = ws.rhi;
  const double inv_span = ws.inv_span, phase = ws.phase, cf = ws.cf;
  auto shape = [](double c1) -> double {
    return BLACKMAN ? (0.08 * (2 * c1 * c1 - 1) + 0.5 * c1 + 0.42) : (0.5 * c1 + 0.5);  // cos(2a) = 2cos^2(a)-1
  };
  auto win = [&](int j) -> double { return shape(cospi(((double)(j - hwl) * inv_span + phase) * cf)); };
  const wh::ckp<const double> xb = xu + (ws.centre - 1);  // (re-derived from laundered bits before the second walk)
  auto sample = [&](int j) -> double {
    int rel = j - hwl;
    rel = rel < rlo ? rlo : rel;
    rel = rel > rhi ? rhi : rel;
    return xb[rel];
  };
  const double rot_s = ws.rot_s, rot_c = ws.rot_c;
  // phase of this thread's first sample: base * E[tid]
  const double c0 = ws.base_c * e_tid.y - ws.base_s * e_tid.x;
  const double s0 = ws.base_s * e_tid.y + ws.base_c * e_tid.x;
  double s_sw = 0.0, s_w = 0.0, s_swsw = 0.0, s_sww = 0.0, s_ww = 0.0;
  // The gather is ONE round: all Q loads of the thread are issued together and parked in LDS as they arrive (each
  // thread only ever touches its own slots, so no barrier is involved).  Both walks then read LDS.  Gathering inside
  // the walks — in chunks of four, twice — put four dependent global-memory round trips (~3.5 k cycles each on the
  // loaded chip) into every window: 15 k of a window's 20 k cycles; carrying the samples in registers across the
  // reduction instead was the register peak of the kernel.
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int j = threadIdx.x + q * FT;
    slot[j * STRIDE] = sample(j);  // clamped: always a valid address
  }
  {
    double c = c0, sn = s0;
#pragma unroll 2
    for (int q = 0; q < Q; ++q) {
      const int j = threadIdx.x + q * FT;
      if (j < L) {
        const double w = shape(c);
        const double sw = slot[j * STRIDE] * w;
        s_sw += sw;
        s_w += w;
        if (ENERGY) {
          s_swsw += sw * sw;
          s_sww += sw * w;
          s_ww += w * w;
        }
      }
      const double cn = c * rot_c - sn * rot_s;
      sn = sn * rot_c + c * rot_s;
      c = cn;
    }
    for (int j = N + threadIdx.x; j < L; j += FT) {  // rows longer than N: cropped, but they count in the sums
      const double w = win(j);
      const double sw = sample(j) * w;
      s_sw += sw;
      s_w += w;
      if (ENERGY) {
        s_swsw += sw * sw;
        s_sww += sw * w;
        s_ww += w * w;
      }
    }
  }
  STAGE_MARK(10)
  if (ENERGY) wh::block_sum5<FT>(s_sw, s_w, s_swsw, s_sww, s_ww, scratch);
  else wh::block_sum2<FT>(s_sw, s_w, scratch);
  STAGE_MARK(11)
  const double dc = s_sw / s_w;  // = mean(x w) / mean(w): the two divisions by L cancel (two FP64 divides less per window)
  const double inv_nrm = ENERGY ? 1.0 / sqrt((s_swsw - 2.0 * dc * s_sww) + dc * dc * s_ww) : 1.0;
  {
    double c = c0, sn = s0;
#pragma unroll 2
    for (int q = 0; q < Q; ++q) {
      const int j = threadIdx.x + q * FT;
      double val = 0.0;
      if (j < L) {
        const double w = shape(c);
        val = slot[j * STRIDE] * w - w * dc;
        if (ENERGY) val *= inv_nrm;
      }
      emit(j, val);
      const double cn = c * rot_c - sn * rot_s;
      sn = sn * rot_c + c * rot_s;
      c = cn;
    }
  }
}

// The register-fed form (N = 8 * FT: the lengths D4C runs at from 16 kHz up).  A thread's Q = N / FT samples
// j = tid + q*FT are exactly the operands of its radix-8 butterfly in the FIRST pass of the transform that follows, so
// the windowed frame never exists in LDS: one round of global loads into registers, walk 1 (the sums) and walk 2 (the
// DC-removed, normalised values) over those registers, and out[q] goes straight into wh::fft_lds_from_regs.  Against
// d4c_window this removes, per frame and window, the parking store (2048 x 8 B), both walks' LDS reads, the emit of
// 2048 complex values and the first pass's read of them — stores are what an FFT pass costs on this LDS (~80 B/clk per
// CU, MI355X_MICROARCH.md) — and the walks stop at the window's end: rows q >= ceil(L / FT) are zeros (a window spans
// 4 pitch periods, ~640 of the 2048 samples at 100 Hz), uniformly for the workgroup.
template <bool BLACKMAN, int N, bool ENERGY, int FT_ = 0>
__device__ __forceinline__ void d4c_window_regs(wh::ckp<const double> WH_RESTRICT xu, wh::ckp<const double> tab, double2 e_tid,
                                                wh::ckp<double> scratch, double (&out)[N / (FT_ ? FT_ : ft_of(N))]) {
  constexpr int FT = FT_ ? FT_ : ft_of(N);
  constexpr int Q = N / FT;
  const WinSetup ws = win_load(tab);
  const int hwl = ws.hwl, L = ws.L, rlo = ws.rlo, rhi = ws.rhi;
  const double inv_span = ws.inv_span, phase = ws.phase, cf = ws.cf;
  auto shape = [](double c1) -> double {
    return BLACKMAN ? (0.08 * (2 * c1 * c1 - 1) + 0.5 * c1 + 0.42) : (0.5 * c1 + 0.5);  // cos(2a) = 2cos^2(a)-1
  };
  auto win = [&](int j) -> double { return shape(cospi(((double)(j - hwl) * inv_span + phase) * cf)); };
  const wh::ckp<const double> xb = xu + (ws.centre - 1);
  auto sample = [&](int j) -> double {
    int rel = j - hwl;
    rel = rel < rlo ? rlo : rel;
    rel = rel > rhi ? rhi : rel;
    return xb[rel];
  };
  const int nq = L >= N ? Q : (L + FT - 1) / FT;  // rows that hold window samples (workgroup-uniform)
  // All Q loads are issued unconditionally (the index is clamped: always a valid address; rows past the window read its
  // last sample, one line for the whole wave) and the rows past nq zeroed by a select.  Written as `if (q < nq) out[q] =
  // sample(..)` the compiler made a chain of conditional blocks, each WAITING for its load before the next block's and
  // copying the whole array between them: nq dependent global round trips and ~25 register moves per row.
#if WH_D4C_GATHER_UNCOND
#pragma unroll
  for (int q = 0; q < Q; ++q) out[q] = sample(threadIdx.x + q * FT);
#pragma unroll
  for (int q = 0; q < Q; ++q) out[q] = q < nq ? out[q] : 0.0;
#else
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    out[q] = 0.0;
    if (q < nq) out[q] = sample(threadIdx.x + q * FT);
  }
#endif
  const double rot_s = ws.rot_s, rot_c = ws.rot_c;
  const double c0 = ws.base_c * e_tid.y - ws.base_s * e_tid.x;  // phase of this thread's first sample: base * E[tid]
  const double s0 = ws.base_s * e_tid.y + ws.base_c * e_tid.x;
  double s_sw = 0.0, s_w = 0.0, s_swsw = 0.0, s_sww = 0.0, s_ww = 0.0;
  // The window values of the first KEEPQ rows are kept for the second walk (rows beyond that — windows longer than
  // KEEPQ * FT samples, f0 below ~62 Hz at N = 2048 — re-derive theirs by the rotation, as every row used to).
  constexpr int KEEPQ = N >= 4096 ? 0 : (WH_D4C_KEEP_W < Q ? WH_D4C_KEEP_W : Q);  // (N = 4096: 26 ... 276 spilled registers)
  double wk[KEEPQ > 0 ? KEEPQ : 1];
  {
    double c = c0, sn = s0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (q < nq) {
        const int j = threadIdx.x + q * FT;
        double w = 0.0;
        if (j < L) {
          w = shape(c);
          const double sw = out[q] * w;
          s_sw += sw;
          s_w += w;
          if (ENERGY) {
            s_swsw += sw * sw;
            s_sww += sw * w;
            s_ww += w * w;
          }
        }
        if (q < KEEPQ) wk[q] = w;
        const double cn = c * rot_c - sn * rot_s;
        sn = sn * rot_c + c * rot_s;
        c = cn;
      }
    }
    for (int j = N + threadIdx.x; j < L; j += FT) {  // rows longer than N: cropped, but they count in the sums
      const double w = win(j);
      const double sw = sample(j) * w;
      s_sw += sw;
      s_w += w;
      if (ENERGY) {
        s_swsw += sw * sw;
        s_sww += sw * w;
        s_ww += w * w;
      }
    }
  }
  STAGE_MARK(10)
  if (ENERGY) wh::block_sum5<FT>(s_sw, s_w, s_swsw, s_sww, s_ww, scratch);
  else wh::block_sum2<FT>(s_sw, s_w, scratch);
  STAGE_MARK(11)
  const double dc = s_sw / s_w;  // = mean(x w) / mean(w): the two divisions by L cancel (two FP64 divides less per window)
  const double inv_nrm = ENERGY ? 1.0 / sqrt((s_swsw - 2.0 * dc * s_sww) + dc * dc * s_ww) : 1.0;
  {
    double c = c0, sn = s0;
    if (KEEPQ < Q && nq > KEEPQ) {  // phase of row KEEPQ for the rows that re-derive their window value
#pragma unroll
      for (int q = 0; q < KEEPQ; ++q) {
        const double cn = c * rot_c - sn * rot_s;
        sn = sn * rot_c + c * rot_s;
        c = cn;
      }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (q < nq) {
        const int j = threadIdx.x + q * FT;
        double val = 0.0;
        if (q < KEEPQ) {
          const double w = wk[q];  // 0 past the window's end
          val = out[q] * w - w * dc;
          if (ENERGY) val *= inv_nrm;
          if (!(j < L)) val = 0.0;
        } else {
          if (j < L) {
            const double w = shape(c);
            val = out[q] * w - w * dc;
            if (ENERGY) val *= inv_nrm;
          }
          const double cn = c * rot_c - sn * rot_s;
          sn = sn * rot_c + c * rot_s;
          c = cn;
        }
        out[q] = val;
      }
    }
  }
}
// Quantities of a launch that depend on the sampling rate and the transform length alone, evaluated once on the host
// with the reference's expressions (d4c.py:78-80, 197-199) instead of by every wave of every frame (FP64 divides and
// ceil / floor: ~60 VALU instructions per frame that no lane needs to repeat).
struct D4cLaunchConst {
  int b0, b1, b2;      // love-train band edges: ceil(100 | 4000 | 7900 / (fs / N)) + 1
  int boundary;        // int(N / wlen * 8 + 0.5)
  int centre[8];       // per band: floor(interval * (b + 1) / (fs / N))
};
inline D4cLaunchConst d4c_launch_const(double fs, int n, int wlen, int interval, int nap) {
  D4cLaunchConst c;
  c.b0 = (int)(ceil(100.0 / (fs / n)) + 1);
  c.b1 = (int)(ceil(4000.0 / (fs / n)) + 1);
  c.b2 = (int)(ceil(7900.0 / (fs / n)) + 1);
  c.boundary = (int)((double)n / wlen * 8 + 0.5);
  for (int b = 0; b < 8; ++b) c.centre[b] = b < nap ? (int)floor((double)interval * (b + 1) / (fs / n)) : 0;
  return c;
}

template <int N>
constexpr bool d4c_regfed() { return WH_D4C_REGFED && N == 8 * ft_of(N); }

// Threads per frame of the stand-alone gate kernel: its one transform is real (N/2 complex points), so N/16 threads are
// one radix-8 butterfly each — half of d4c_kernel's count.
#ifndef WH_FT_LOVE_DIV
#define WH_FT_LOVE_DIV 2
#endif
constexpr int ft_love(int n) { return ft_of(n) / WH_FT_LOVE_DIV < 64 ? 64 : ft_of(n) / WH_FT_LOVE_DIV; }
template <int NLT>
__global__ __launch_bounds__(ft_love(NLT)) void love_train_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int32_t* __restrict__ frame_utt,
    const double* __restrict__ tp, double* __restrict__ f0_io, const double* __restrict__ vuv, double fs,
    double threshold, const double2* __restrict__ tw_base, int32_t* __restrict__ gate, long long n_frames) {
  constexpr int FT = ft_love(NLT);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // (wh::ckp<T> is T* in every shipped build; the bounds build checks each access against the range named here)
  const wh::ckp<double> lds_all = wh::ck_make(reinterpret_cast<double*>(smem), NLT + 2 + 48 + kWinTab, wh::WH_CK_LDS_OTHER);
  const wh::ckp<double> zr = wh::ck_sub(lds_all, 0, NLT + 2, wh::WH_CK_LDS_MAIN);  // the NLT real samples, then the half spectrum (NLT + 2 doubles)
  const wh::ckp<double2> zb = wh::ck_as<double2>(zr);                               // NLT/2+1 complex after the real FFT
  const wh::ckp<double> scratch = wh::ck_sub(lds_all, NLT + 2, 48, wh::WH_CK_LDS_SCRATCH);
  const wh::ckp<double> wtab = wh::ck_sub(lds_all, NLT + 2 + 48, kWinTab, wh::WH_CK_LDS_AUX);  // window set-up table (kWinTab doubles)
  const wh::ckp<const double2> tw = wh::ck_make(tw_base, 2 * WH_MAX_TWIDDLE, wh::WH_CK_TWIDDLE);
  const int64_t f = wh::xcd_unit(blockIdx.x, n_frames);
  if (f >= n_frames) return;
  double f0 = f0_io[f];
  if (vuv[f] == 0.0) f0 = 0.0;  // d4c.py:32 — written back (Q6)
  if (threadIdx.x == 0) f0_io[f] = f0;
  if (f0 == 0.0) {
    if (threadIdx.x == 0) gate[f] = 0;
    return;
  }
  const int u = frame_utt[f];
  const long long xn = x_off[u + 1] - x_off[u];
  const wh::ckp<const double> xu = wh::ck_make(x + x_off[u], xn, wh::WH_CK_WAVEFORM);
  const double cf = fmax(f0, 40.0);
  if (threadIdx.x == 0) win_setup(wtab, xn, fs, cf, tp[f], 1.5, FT);
  wh::sync<FT>();
#if WH_LOVE_REGWIN
  {
    // the frame's samples stay in registers from the gather through both walks (round 6): parked in LDS between them, the
    // window cost a store and two reads per sample on an LDS pipe this kernel keeps busy all the time
    double v[NLT / FT];
    d4c_window_regs<true, NLT, false, FT>(xu, wtab, win_thread_phase(wtab[9]), scratch, v);
#pragma unroll
    for (int q = 0; q < NLT / FT; ++q) zr[WH_TID + q * FT] = v[q];
  }
#else
  d4c_window<true, NLT, false, 1, FT>(xu, wtab, win_thread_phase(wtab[9]), scratch, zr, [&](int j, double val) { zr[j] = val; });
#endif
  wh::sync<FT>();
  wh::rfft_lds<NLT, FT, FT, WH_LOVE_MAXR>(zb, tw);  // (66 VGPRs here: the radix-8 plan fits, unlike in d4c_kernel)
  const int b0 = (int)(ceil(100.0 / (fs / NLT)) + 1);
  const int b1 = (int)(ceil(4000.0 / (fs / NLT)) + 1);
  const int b2 = (int)(ceil(7900.0 / (fs / NLT)) + 1);
  double s1 = 0.0, s2 = 0.0;
  {
    // the voicing decision s1 / s2 > 0.85 (d4c.py:86): powers, sums and ratio stay unfused, like NumPy's (ADVICE r5 — the
    // translation unit is compiled with contraction for its transforms; what feeds a discrete decision is not)
#pragma clang fp contract(off)
    for (int k = b0 + threadIdx.x; k < b2 && k < NLT; k += FT) {
      const double2 z = zb[k <= NLT / 2 ? k : NLT - k];  // |X[k]|^2 is even about NLT/2
      const double p = z.x * z.x + z.y * z.y;
      s2 += p;
      if (k < b1) s1 += p;
    }
  }
  wh::block_sum2<FT>(s1, s2, scratch);
  if (threadIdx.x == 0) gate[f] = (s1 / s2 > threshold) ? 1 : 0;
}

// Sum of the m smallest of K non-negative values and their total, without sorting and without LDS atomics.
// The reference sorts the K powers and prefix-sums them (world/d4c.py:206-208); only the VALUES of the m smallest
// enter the sum, and m = K - (boundary + 1) is close to K, so the kernel finds the few LARGE values to leave out:
//   (1) per wave, the maximum IEEE exponent (shuffles) and the population counts of the 8 exponents at and below it
//       by ballot (scalar unit); counts AND the wave's maximum go through LDS together, so one hop yields the block
//       maximum and the block's counts; the window slides further down in the rare case that the K - m largest span
//       more than 8 octaves;
//   (2) the one exponent bin that holds the threshold is compacted into a list at offsets derived from the same
//       ballots (no atomic counter) and its members are ranked against each other (~11 on speech); equal values are
//       interchangeable in a sum, so ties need no index rule.
// Each thread then adds its own kept elements in a fixed order -> deterministic sums.
// x[q], q < PER: the thread's share of the K values (bit q of `valid` set where the slot is used — any assignment of the
// values to threads will do).  work: >= 80 ints + K doubles of free
// LDS; scratch: 32 doubles.  Four barrier phases.
#ifndef WH_D4C_SEL_PACKED
#define WH_D4C_SEL_PACKED 1
#endif
// 64-lane sum of a 32-bit integer on the VALU (the DPP ladder of wh::wave_sum), result uniform
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
  v += (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  v += (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  v += (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
  v += (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true);  // row_mirror
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// mine[e] = number of slots in the wave whose digit (0 .. WIN-1, or < 0: not counted) is e.  Packed form: every lane
// counts its own slots into 16-bit fields (two digit values per register) and the registers are summed over the wave
// by DPP — VALU only; the ballot form is a v_cmp, an s_bcnt1 and an s_add per (digit value, slot), each SALU instruction
// waiting for the VALU-written mask.
template <int WIN, int PER, class Digit>
__device__ __forceinline__ void wave_digit_counts(Digit digit, int (&mine)[WIN]) {
#if WH_D4C_SEL_PACKED
  static_assert(WIN % 2 == 0, "two digit values per register");
  unsigned c[WIN / 2];
#pragma unroll
  for (int j = 0; j < WIN / 2; ++j) c[j] = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int d = digit(q);
#pragma unroll
    for (int j = 0; j < WIN / 2; ++j) c[j] += (d >> 1) == j ? (1u << (16 * (d & 1))) : 0u;
  }
#pragma unroll
  for (int j = 0; j < WIN / 2; ++j) {
    const unsigned ssum = wave_sum_u32(c[j]);
    mine[2 * j] = (int)(ssum & 0xFFFF);
    mine[2 * j + 1] = (int)(ssum >> 16);
  }
#else
#pragma unroll
  for (int e = 0; e < WIN; ++e) {
    int cc = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) cc += __popcll(__ballot(digit(q) == e));
    mine[e] = cc;
  }
#endif
}

template <int K, int FT, int PER>
__device__ __forceinline__ void sum_smallest(const double (&x)[PER], unsigned valid, int m, wh::ckp<double> work, wh::ckp<double> scratch,
                                             double* s_small, double* s_total) {
  constexpr int NW = FT / 64;
  // exponents per round.  On speech the K - m (~22) largest bins lie within 4 octaves of the maximum on average, 7 at
  // most (measured on the oracle's spectra): one round of 8 almost always; the loop below slides on otherwise.
  // Long spectra (K = 2049 at 48 kHz: 65 bins dropped, spread over more octaves, hundreds of values in the threshold bin)
  // take 16 exponents per round and 4 mantissa bits per refinement level: the same number of ballots in half as many
  // rounds, i.e. half as many barrier pairs and count exchanges.
#ifndef WH_D4C_SEL_DB_LONG
#define WH_D4C_SEL_DB_LONG 2
#endif
#ifndef WH_D4C_SEL_DB
#define WH_D4C_SEL_DB 2
#endif
  constexpr int DB = K > 1100 ? WH_D4C_SEL_DB_LONG : WH_D4C_SEL_DB;   // mantissa bits per refinement level
  constexpr int WIN = 1 << DB;           // exponents per round = values of a mantissa digit
  const wh::ckp<int> cnts = wh::ck_as<int>(work);               // [NW][WIN + 1]: counts per exponent, then the wave's top
  const wh::ckp<double> list = work + (NW * (WIN + 1) + (NW * (WIN + 1) & 1)) / 2;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int key[PER];
  int kmax = 0;
  double t = 0.0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const bool in = (valid >> q) & 1u;  // slot q of this thread holds one of the K values
    key[q] = in ? (int)((__double_as_longlong(x[q]) >> 52) & 0x7FF) : -1;
    if (in) t += x[q];
    kmax = key[q] > kmax ? key[q] : kmax;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int u = __shfl_xor(kmax, o, 64);
    kmax = u > kmax ? u : kmax;
  }
  const int drop = K - m;  // how many of the largest values are left out (>= 1)
  // Round 0 counts below the WAVE's own maximum and publishes that maximum next to the counts: one LDS hop gives every
  // thread the block maximum and all counts (a wave whose maximum is lower has nothing above its own window, and its
  // window reaches at least as far down as the block's).  Further rounds (rare) count below the common `top`.
  int top = kmax;          // exponent at the top of this wave's current window
  int above = 0;           // elements with an exponent above the window
  int tbin = -1, wave_before = 0, in_bin = 0;
  bool first = true;
  while (true) {
    int mine[WIN];
    wave_digit_counts<WIN, PER>([&](int q) {  // digit e = exponent top - e; padding slots (key -1) and the rest: none
      const int d = top - key[q];
      return (key[q] >= 0 && d >= 0 && d < WIN) ? d : -1;
    }, mine);
    wh::sync<FT>();  // the work area is free (previous round's counts have been read by everyone)
    if (lane <= WIN) {
      int c = top;
#pragma unroll
      for (int e = 0; e < WIN; ++e) c = lane == e ? mine[e] : c;
      cnts[w * (WIN + 1) + lane] = c;
    }
    wh::sync<FT>();
    int gtop = top;
    if (first) {
#pragma unroll
      for (int i = 0; i < NW; ++i) gtop = cnts[i * (WIN + 1) + WIN] > gtop ? cnts[i * (WIN + 1) + WIN] : gtop;
    }
    // lane e sums the waves' counts of offset e (and what the waves in front of this one hold of it); the eight results
    // come back through readlane as uniform values — 8 LDS reads in 8 lanes instead of 64 in every lane
    int tot_l = 0, bef_l = 0;
    if (lane < WIN) {
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        // wave i counted exponent gtop - e at its own offset e - (gtop - top_i)
        const int sh = first ? gtop - cnts[i * (WIN + 1) + WIN] : 0;
        const int c = lane - sh >= 0 ? cnts[i * (WIN + 1) + (lane - sh)] : 0;
        bef_l += i < w ? c : 0;
        tot_l += c;
      }
    }
    int run = above;
#pragma unroll
    for (int e = 0; e < WIN; ++e) {  // e: offset below the BLOCK's top
      const int tot = __builtin_amdgcn_readlane(tot_l, e), before = __builtin_amdgcn_readlane(bef_l, e);
      if (tbin < 0 && run + tot >= drop) {
        tbin = gtop - e;
        above = run;
        wave_before = before;
        in_bin = tot;
      }
      run += tot;
    }
    if (tbin >= 0 || gtop - WIN < 0) break;
    above = run;
    top = gtop - WIN;  // every wave continues below the common window
    first = false;
  }
  // tbin < 0 cannot happen (every element has an exponent in [0, kmax]); guard anyway: drop nothing more
  int need = tbin >= 0 ? drop - above : 0;  // members of the threshold bin that belong to the large set
  double a = 0.0;
#pragma unroll
  for (int q = 0; q < PER; ++q)
    if (key[q] >= 0 && key[q] < tbin) a += x[q];  // everything below the threshold bin is kept
  // The members of the threshold bin are ranked against each other below, in_bin^2 / FT comparisons: fine for the ~11
  // members a 2048-point band spectrum leaves there (22 bins dropped of 1025), not for the hundreds of a 4096-point one
  // (65 of 2049: 41 % of the whole kernel at 48 kHz).  While the bin holds more than 32 values it is split by the next
  // three mantissa bits — the same ballot counts, eight digits, one LDS hop — and only the digit that holds the
  // threshold stays a candidate: larger digits are dropped whole, smaller ones kept whole.
  unsigned cand = 0;  // bit q: slot q is a member of the current threshold set
#pragma unroll
  for (int q = 0; q < PER; ++q) cand |= (key[q] == tbin ? 1u : 0u) << q;
  int shift = 52;
  while (in_bin > 32 && shift >= DB) {  // (uniform)
    shift -= DB;
    int dig[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) dig[q] = ((cand >> q) & 1u) ? (int)((__double_as_longlong(x[q]) >> shift) & (WIN - 1)) : -1;
    int mine[WIN];
    wave_digit_counts<WIN, PER>([&](int q) { return dig[q] >= 0 ? WIN - 1 - dig[q] : -1; }, mine);  // e counts down from the largest digit
    wh::sync<FT>();
    if (lane < WIN) {
      int c = 0;
#pragma unroll
      for (int e = 0; e < WIN; ++e) c = lane == e ? mine[e] : c;
      cnts[w * (WIN + 1) + lane] = c;
    }
    wh::sync<FT>();
    int tot_l = 0, bef_l = 0;
    if (lane < WIN) {
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const int c = cnts[i * (WIN + 1) + lane];
        bef_l += i < w ? c : 0;
        tot_l += c;
      }
    }
    int run = 0, td = -1, bef = 0, tot_d = 0, run_at = 0;
#pragma unroll
    for (int e = 0; e < WIN; ++e) {
      const int tot = __builtin_amdgcn_readlane(tot_l, e), before = __builtin_amdgcn_readlane(bef_l, e);
      if (td < 0 && run + tot >= need) {
        td = WIN - 1 - e;
        run_at = run;
        bef = before;
        tot_d = tot;
      }
      run += tot;
    }
    // (td >= 0 always: the set holds at least `need` members)
    need -= run_at;
    wave_before = bef;
    in_bin = tot_d;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      if (dig[q] >= 0 && dig[q] < td) a += x[q];  // below the threshold digit: kept
      if (dig[q] != td) cand &= ~(1u << q);
    }
  }
  // compaction of the threshold set: wave offset from the per-wave counts, lane offset from the ballots
  {
    int pos = wave_before;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const bool mem = (cand >> q) & 1u;
      const unsigned long long mk = __ballot(mem);
      if (mem) list[pos + __popcll(mk & ((1ull << lane) - 1ull))] = x[q];
      pos += __popcll(mk);
    }
  }
  wh::sync<FT>();
  // the threshold bin: list entry i is kept unless it is one of the `need` largest (ties: list order).  One entry
  // per thread, so the ranking costs in_bin LDS reads per thread whatever the distribution of the bin over threads.
  for (int i = threadIdx.x; i < in_bin; i += FT) {
    const double v = list[i];
    int ahead = 0;
    for (int j = 0; j < in_bin; ++j) {
      const double o = list[j];
      ahead += (o > v || (o == v && j < i)) ? 1 : 0;
    }
    if (ahead >= need) a += v;
  }
  wh::block_sum2<FT>(a, t, scratch);
  *s_small = a;
  *s_total = t;
}

// ---- thread-owned runs of bins ---------------------------------------------------------------------------------
// From the first spectrum to the band stage the K = N/2+1 per-bin quantities of a frame (smoothed power, group-delay
// centroid and what the smoothings make of them) live in REGISTERS: thread t owns the bins [t*KR, (t+1)*KR).  LDS then
// holds nothing but the 2N-double transform buffer (32 KB at N = 2048 -> 4 workgroups per CU instead of 3 with the
// two 8 KB per-bin arrays of the first version), and the smoothings (BandWindow, wh_spectral.h) read and write the
// same runs.
template <int N>
struct Runs {
  static constexpr int FT = ft_of(N);
  static constexpr int K = N / 2 + 1;
  static constexpr int KR = (K + FT - 1) / FT;
};

// Mirror-add of the bins below f0 (wh::low_band_replica, d4c.py:213-220) for a run-resident array: the owners of
// the bins below `reach` publish them to tmp (LDS, >= 2*nlow doubles), the interpolated replica is evaluated by a
// thread-strided loop (a handful of bins; kept out of the unrolled per-run code, whose five copies of the divides and
// searches cost ~30 VGPRs of spills) and the owners add it to their registers.
template <int N>
__device__ __forceinline__ void low_band_replica_runs(double (&p)[Runs<N>::KR], wh::ckp<double> tmp, double fs, double f0,
                                                      double reach) {
  constexpr int FT = Runs<N>::FT, K = Runs<N>::K, KR = Runs<N>::KR;
  const int k0 = threadIdx.x * KR;
  int nlow = (int)(reach / fs * N) + 2;  // count of bins with k/N*fs < reach (monotone in k)
  if (nlow > K) nlow = K;                // (the reference indexes the half spectrum: bins beyond it do not exist)
  while (nlow > 0 && !(((double)(nlow - 1) / N * fs) < reach)) --nlow;
  const wh::ckp<double> add = tmp + ((nlow + 1) & ~1);
  // The bins below `reach` (1.2 f0 <= 960 Hz: a few dozen) all belong to the first lanes of wave 0.  When they fit one
  // wave — always, at the supported rates — that wave does the whole correction with wave-level ordering and the other
  // waves only meet it at the closing barrier: one barrier instead of three, and three waves skip the code.
  const bool one_wave = nlow <= 64 * KR;
  if (!one_wave || threadIdx.x < 64) {
#pragma unroll
    for (int r = 0; r < KR; ++r)
      if (k0 + r < nlow) tmp[k0 + r] = p[r];
    if (one_wave) wh::sync<64>(); else wh::sync<FT>();
#pragma unroll 1
    for (int kk = threadIdx.x; kk < nlow; kk += (one_wave ? 64 : FT)) {
      const double fk = (double)kk / N * fs;
      double inc = 0.0;
      if (nlow >= 2 && fk < f0) {
        // ascending nodes a_m = f0 - f_{nlow-1-m}; hi = clamp(#nodes < fk, 1, nlow-1).  The node predicate
        // a_m < fk is monotone in m, so the count is its boundary: estimated in closed form, then settled with
        // the exact floating-point predicate (the estimate is within one of the truth).
        auto below = [&](int mm) { return (f0 - ((double)(nlow - 1 - mm) / N * fs)) < fk; };
        int cnt = (int)ceil((double)(nlow - 1) - (f0 - fk) / fs * N);
        cnt = cnt < 0 ? 0 : (cnt > nlow ? nlow : cnt);
        while (cnt > 0 && !below(cnt - 1)) --cnt;
        while (cnt < nlow && below(cnt)) ++cnt;
        const int hi = cnt < 1 ? 1 : (cnt > nlow - 1 ? nlow - 1 : cnt);
        const int lo = hi - 1;
        const double a_lo = f0 - ((double)(nlow - 1 - lo) / N * fs);
        const double a_hi = f0 - ((double)(nlow - 1 - hi) / N * fs);
        const double y_lo = tmp[nlow - 1 - lo];
        const double y_hi = tmp[nlow - 1 - hi];
        const double slope = (y_hi - y_lo) / (a_hi - a_lo);
        inc = slope * (fk - a_lo) + y_lo;
      }
      add[kk] = inc;
    }
    if (one_wave) wh::sync<64>(); else wh::sync<FT>();
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      const int kk = k0 + r;
      if (kk < nlow && nlow >= 2 && ((double)kk / N * fs) < f0) p[r] = add[kk] + p[r];
    }
  }
  wh::sync<FT>();  // tmp is reused by the caller
}

// v[0..N) = Hermitian mirror of the run-resident half spectrum times fs/N (wh::fill_mirrored for runs).
template <int N>
__device__ __forceinline__ void fill_mirrored_runs(const double (&p)[Runs<N>::KR], wh::ckp<double> v, double fs) {
  constexpr int FT = Runs<N>::FT, K = Runs<N>::K, KR = Runs<N>::KR;
  const int k0 = threadIdx.x * KR;
  const double df = fs / N;
#pragma unroll
  for (int r = 0; r < KR; ++r) {
    const int k = k0 + r;
    if (k < K) {
      const double val = p[r] * df;
      v[k] = val;
      if (k > 0 && k < N / 2) v[N - k] = val;
    }
  }
  wh::sync<FT>();
}

// Group-delay centroid of one Blackman frame, added into the run-resident cent (d4c.py:146-153).
// x and n*x (two real sequences) share ONE complex FFT: z = x + i*n*x, separated afterwards by symmetry.
template <int N>
__device__ __forceinline__ void add_centroid(wh::ckp<const double> xu, wh::ckp<const double> wtab, double2 e_tid,
                                             wh::ckp<double2> buf, double (&cent)[Runs<N>::KR], bool first,
                                             wh::ckp<const double2> tw_base, wh::ckp<double> scratch) {
  constexpr int FT = Runs<N>::FT, K = Runs<N>::K, KR = Runs<N>::KR;
  // z[j] = x[j] + i*(j+1)*x[j] (n is 1-based), normalised frame
  if constexpr (d4c_regfed<N>()) {
    double val[N / FT];
    d4c_window_regs<true, N, true>(xu, wtab, e_tid, scratch, val);
    double2 zin[N / FT];
#pragma unroll
    for (int q = 0; q < N / FT; ++q) zin[q] = make_double2(val[q], val[q] * (double)(WH_TID + q * FT + 1));
    wh::fft_lds_from_regs<N, false, FT, 8>(zin, buf, fresh_table(tw_base) + N);
  } else {
    // written straight into the transform buffer
    d4c_window<true, N, true, 2>(xu, wtab, e_tid, scratch, wh::ck_as<double>(buf),
                                 [&](int j, double val) { buf[j] = make_double2(val, val * (double)(j + 1)); });
    wh::sync<FT>();
    wh::fft_lds<N, false, FT, FT, d4c_maxr(N)>(buf, fresh_table(tw_base) + N);
  }
  const int k0 = threadIdx.x * KR;
#pragma unroll
  for (int r = 0; r < KR; ++r) {
    const int k = k0 + r;
    if (k < K) {
      const double2 a = buf[k];
      const double2 b = buf[(N - k) & (N - 1)];
      // S = (Z[k]+conj(Z[N-k]))/2 ; T = (Z[k]-conj(Z[N-k]))/(2i)
      const double sr = 0.5 * (a.x + b.x), si = 0.5 * (a.y - b.y);
      const double tr = 0.5 * (a.y + b.y), ti = -0.5 * (a.x - b.x);
      const double c = tr * sr + si * ti;  // -Im(W)Re(S)+Im(S)Re(W) with W = -i*T
      cent[r] = first ? c : cent[r] + c;
    }
  }
  wh::sync<FT>();  // buf is free again
}

// FUSED (love-train FFT size == D4C FFT size, the case at 16/22.05/44.1/48 kHz for d4c()): the VUV gate's
// Blackman frame and the Hann frame of the smoothed power spectrum are two real sequences → ONE complex FFT,
// separated by Hermitian symmetry; the separate love_train_kernel launch and one transform disappear.
template <int N, bool FUSED>
__device__ __forceinline__ void d4c_frame(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int32_t* __restrict__ frame_utt,
    const double* __restrict__ tp, double* __restrict__ f0_io, const double* __restrict__ vuv,
    const int32_t* __restrict__ gate, double threshold, double fs, int nap, int interval,
    const double* __restrict__ window, int wlen, const double2* __restrict__ tw_base,
    int k_spec,                       // >0: dense amplitude output [F][k_spec]; 0: Requiem band output [F][nap+2]
    double* __restrict__ out, double* __restrict__ coarse_dbg, long long n_frames, const D4cLaunchConst& lc, int64_t f) {
  constexpr int FT = Runs<N>::FT, K = Runs<N>::K, KR = Runs<N>::KR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // (wh::ckp<T> is T* in every shipped build; the bounds build checks each access against the range named here)
  const wh::ckp<double> lds_all = wh::ck_make(reinterpret_cast<double*>(smem), 2 * N + 40 + 8 + 4 * kWinTab, wh::WH_CK_LDS_OTHER);
  const wh::ckp<double> zr = wh::ck_sub(lds_all, 0, 2 * N, wh::WH_CK_LDS_MAIN);  // 2N doubles: real buffers, mirrored spectra, scratch
  const wh::ckp<double2> buf = wh::ck_as<double2>(zr);   // the same as N complex (centroid FFT) / N/2+1 complex (real FFTs)
  const wh::ckp<double> scratch = wh::ck_sub(lds_all, 2 * N, 40, wh::WH_CK_LDS_SCRATCH);  // 40 (block_sum5 at 8 waves)
  const wh::ckp<double> band = wh::ck_sub(lds_all, 2 * N + 40, 8, wh::WH_CK_LDS_AUX);     // nap (<= 8)
  const wh::ckp<double> wtab = wh::ck_sub(lds_all, 2 * N + 48, 4 * kWinTab, wh::WH_CK_LDS_AUX);  // 4 windows x kWinTab: gate, power, centroid +, centroid -
  constexpr int KPAD = (K + 1) & ~1;
  const wh::ckp<double> td = zr + (2 * N - KPAD);        // band stage: the shaped group delay, above the real-FFT buffer
  const wh::ckp<const double2> tw = wh::ck_make(tw_base, 2 * WH_MAX_TWIDDLE, wh::WH_CK_TWIDDLE);  // the table of size n at offset n
  const wh::ckp<const double> win_tab = wh::ck_make(window, wlen, wh::WH_CK_TABLE);

  STAGE_TIMER_BEGIN
  if (f >= n_frames) return;
  const int u = frame_utt[f];
  const long long xn = x_off[u + 1] - x_off[u];
  const wh::ckp<const double> xu = wh::ck_make(x + x_off[u], xn, wh::WH_CK_WAVEFORM);
  const double pos = tp[f];
  double f0v = f0_io[f];
  bool voiced;
  if (FUSED) {
    if (vuv[f] == 0.0) f0v = 0.0;  // d4c.py:32 — written back (Q6)
    if (threadIdx.x == 0) f0_io[f] = f0v;
    voiced = f0v != 0.0;
  } else {
    voiced = gate[f] != 0;
  }
  const double cf = fmax(47.0, f0v);
  const int k0 = threadIdx.x * KR;
  double pw[KR], cent[KR];  // run-resident per-bin arrays
#pragma unroll
  for (int r = 0; r < KR; ++r) pw[r] = cent[r] = 0.0;
  // the four windows of the frame are set up by four lanes at once (win_setup); the three that share f0 and length
  // also share the per-thread phase factor e_frame
  if (voiced && threadIdx.x < 4) {
    const int wdx = threadIdx.x;
    const double wcf = wdx == 0 ? fmax(f0v, 40.0) : cf;
    const double wpos = wdx == 2 ? pos + 1 / cf / 4 : (wdx == 3 ? pos - 1 / cf / 4 : pos);
    win_setup(wtab + wdx * kWinTab, xn, fs, wcf, wpos, wdx == 0 ? 1.5 : 2.0, FT);
  }
  wh::sync<FT>();
  double2 e_frame = make_double2(0.0, 1.0);
  if (voiced) e_frame = win_thread_phase(wtab[kWinTab + 9]);
  if (FUSED && voiced) {
    // love-train frame (Blackman, 3*T0, f0 floored at 40 Hz) and smoothed-power frame (Hann, 4*T0) in one FFT
    if constexpr (d4c_regfed<N>()) {
      double va[N / FT], vb[N / FT];
      d4c_window_regs<true, N, false>(xu, wtab, win_thread_phase(wtab[9]), scratch, va);
      d4c_window_regs<false, N, false>(xu, wtab + kWinTab, e_frame, scratch, vb);
      STAGE_MARK(7)
      double2 zin[N / FT];
#pragma unroll
      for (int q = 0; q < N / FT; ++q) zin[q] = make_double2(va[q], vb[q]);
      wh::fft_lds_from_regs<N, false, FT, 8>(zin, buf, fresh_table(tw) + N);
    } else {
      d4c_window<true, N, false, 2>(xu, wtab, win_thread_phase(wtab[9]), scratch, zr, [&](int j, double val) { zr[2 * j] = val; });
      d4c_window<false, N, false, 2>(xu, wtab + kWinTab, e_frame, scratch, zr + 1, [&](int j, double val) { zr[2 * j + 1] = val; });
      STAGE_MARK(7)
      wh::sync<FT>();
      wh::fft_lds<N, false, FT, FT, d4c_maxr(N)>(buf, fresh_table(tw) + N);
    }
    STAGE_MARK(8)
    const int b0 = lc.b0, b1 = lc.b1, b2 = lc.b2;
    double s1 = 0.0, s2 = 0.0;
    {
    // (the love-train powers feed the voicing decision below: unfused, see love_train_kernel)
#pragma clang fp contract(off)
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      const int k = k0 + r;
      if (k < K) {
        const double2 a = buf[k], b = buf[(N - k) & (N - 1)];
        const double ar = 0.5 * (a.x + b.x), ai = 0.5 * (a.y - b.y);   // love-train spectrum A[k]
        const double br = 0.5 * (a.y + b.y), bi = -0.5 * (a.x - b.x);  // Hann-frame spectrum B[k]
        pw[r] = br * br + bi * bi;
        const double pa = ar * ar + ai * ai;  // |A[k]|^2 = |A[N-k]|^2: bins k and N-k of the full power spectrum
        if (k >= b0 && k < b2) {
          s2 += pa;
          if (k < b1) s1 += pa;
        }
        const int km = N - k;  // mirrored bin (only reached when 7.9 kHz lies above fs/2)
        if (k > 0 && k < N / 2 && km >= b0 && km < b2) {
          s2 += pa;
          if (km < b1) s1 += pa;
        }
      }
    }
    }
    wh::block_sum2<FT>(s1, s2, scratch);  // (its barriers also free buf for the next stage)
    voiced = s1 / s2 > threshold;  // d4c.py:86
  }
  if (!voiced) {
    if (k_spec > 0) {
      const wh::ckp<double> o = wh::ck_make(out + f * (int64_t)k_spec, k_spec, wh::WH_CK_OUT);
      for (int k = threadIdx.x; k < k_spec; k += FT) o[k] = 1 - 0.000000000001;
      if (coarse_dbg) for (int b = threadIdx.x; b < nap; b += FT) coarse_dbg[f * nap + b] = 0.0;
    } else {
      const wh::ckp<double> o = wh::ck_make(out + f * (int64_t)(nap + 2), nap + 2, wh::WH_CK_OUT);
      for (int b = threadIdx.x; b < nap + 2; b += FT) o[b] = -0.000000000001;
    }
    return;
  }

#if WH_D4C_ABLATE == 1
  if (threadIdx.x == 0) out[f * (int64_t)(k_spec > 0 ? k_spec : nap + 2)] = pw[0];
  return;
#endif
  STAGE_MARK(0)
  // ---- static centroid from two frames at +-T0/4 (d4c.py:132-142) ---------------------------------------
#if WH_D4C_CENT_LOOP
  // ONE copy of the centroid frame's code, run twice (cent starts at 0: 0 + c == c): the kernel's instruction stream
  // shrinks by a fifth (66 KB -> 52 KB), below the 64 KB instruction cache that two CUs share
#pragma unroll 1
  for (int c = 0; c < 2; ++c) {
    add_centroid<N>(xu, wtab + (2 + c) * kWinTab, e_frame, buf, cent, false, tw, scratch);
    STAGE_MARK(1 + c)
  }
#else
  add_centroid<N>(xu, wtab + 2 * kWinTab, e_frame, buf, cent, true, tw, scratch);
  STAGE_MARK(1)
  add_centroid<N>(xu, wtab + 3 * kWinTab, e_frame, buf, cent, false, tw, scratch);
  STAGE_MARK(2)
#endif
  low_band_replica_runs<N>(cent, zr, fs, cf, 1.2 * cf);
  STAGE_MARK(3)

#if WH_D4C_ABLATE == 2
  if (threadIdx.x == 0) out[f * (int64_t)(k_spec > 0 ? k_spec : nap + 2)] = cent[0] + pw[3];
  return;
#endif
  // ---- smoothed power spectrum (d4c.py:157-161) ----------------------------------------------
  if (!FUSED) {
    d4c_window<false, N, false, 1>(xu, wtab + kWinTab, e_frame, scratch, zr, [&](int j, double val) { zr[j] = val; });
    wh::sync<FT>();
    wh::rfft_lds<N, FT, FT, d4c_rmaxr(N)>(buf, fresh_table(tw));
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      if (k0 + r < K) {
        const double2 z = buf[k0 + r];
        pw[r] = z.x * z.x + z.y * z.y;
      }
    }
    wh::sync<FT>();
  }
  const wh::ckp<double> cum = zr;  // the FFT buffer is idle during the smoothing steps (low-band scratch, then the mirrored spectra)
  const double inv_cf = 1.0 / cf;
  low_band_replica_runs<N>(pw, cum, fs, cf, 1.2 * cf);
  // the three rectangular smoothings as sliding windowed sums (wh_spectral.h: BandWindow) over the owned runs
  double bandv[KR];
  wh::BandWindow bw;
  fill_mirrored_runs<N>(pw, cum, fs);
  bw.init(cum, N, fs, cf / 2);
  bw.run<KR>(k0, K, bandv);
#pragma unroll
  for (int r = 0; r < KR; ++r) cent[r] = cent[r] / (bandv[r] * inv_cf);  // T_g = centroid / smoothed power (d4c.py:169; no zero guard, Q14)
  wh::sync<FT>();
  // ---- group-delay shaping (d4c.py:165-174) --------------------------------------------------
  fill_mirrored_runs<N>(cent, cum, fs);
  {
    const double w2 = cf / 2;
    const double inv_w2 = 1.0 / w2;
    bw.init(cum, N, fs, w2 / 2);
    bw.run<KR>(k0, K, bandv);
#pragma unroll
    for (int r = 0; r < KR; ++r) pw[r] = bandv[r] * inv_w2;  // T_gs
  }
  wh::sync<FT>();
  fill_mirrored_runs<N>(pw, cum, fs);
  bw.init(cum, N, fs, cf / 2);
  bw.run<KR>(k0, K, bandv);
  wh::sync<FT>();  // everyone is done with the mirrored spectrum before td (inside the same buffer) is written
#pragma unroll
  for (int r = 0; r < KR; ++r)
    if (k0 + r < K) td[k0 + r] = pw[r] - bandv[r] * inv_cf;  // T_D = T_gs - T_gb
  wh::sync<FT>();

#if WH_D4C_ABLATE == 3
  if (threadIdx.x == 0) out[f * (int64_t)(k_spec > 0 ? k_spec : nap + 2)] = td[threadIdx.x];
  return;
#endif
  STAGE_MARK(4)
  // ---- band-wise aperiodicity (d4c.py:192-209) -----------------------------------------------
  const int boundary = lc.boundary;
  const int half = wlen / 2;
  // The band window (Nuttall, wlen <= N / 2 + 1 taps: 2 * floor(interval / (fs / N)) + 1) is the same for every band: a
  // thread's taps j = tid + q FT are fetched ONCE, together, in front of the band loop.  Read inside the fill loop they
  // were one dependent global round trip per tap and band — the fill was 24 % of the N = 4096 instance's latency (five
  // bands at 48 kHz: 62.7 k of 258.7 k cycles per frame, tools/d4c_stage_timer.py 48000), more than the bands' transforms.
  constexpr int WQ = (N / 2 + 1 + FT - 1) / FT;
  double wv[WQ];
#pragma unroll
  for (int q = 0; q < WQ; ++q) {
    const int j = threadIdx.x + q * FT;
    wv[q] = win_tab[j < wlen ? j : 0];  // (clamped, not skipped: no branch per load)
  }
  for (int b = 0; b < nap; ++b) {
    const int centre = lc.centre[b];
#pragma unroll
    for (int q = 0; q < N / FT; ++q) {
      const int j = threadIdx.x + q * FT;
      double val = 0.0;
      if (j < wlen) {
        int idx = centre - half + j;          // index into the mirrored full group delay
        idx = idx < 0 ? -idx : idx;
        idx = idx > N / 2 ? N - idx : idx;
        val = td[idx] * (q < WQ ? wv[q < WQ ? q : 0] : win_tab[j]);
      }
      zr[j] = val;
    }
    wh::sync<FT>();
    STAGE_MARK(12)
    // real transform of the windowed segment = half-size complex transform of its sample pairs + post-pass; only
    // |X[k]|^2 is needed, so the post-pass goes from the pair of bins (k, N/2 - k) straight to the two powers in the
    // thread's registers (the selection below takes its values in any distribution over the threads)
    constexpr int MB = N / 2;
    constexpr int PJ = (MB / 2 + 1 + FT - 1) / FT;  // pair jobs per thread
    double px[2 * PJ];
    unsigned pvalid = 0;
    {
      const wh::ckp<const double2> twb = fresh_table(tw);
      wh::fft_lds<MB, false, FT, FT, d4c_rmaxr(N)>(buf, twb + MB);
      const wh::ckp<const double2> WH_RESTRICT wpost = twb + N;
#pragma unroll
      for (int i = 0; i < PJ; ++i) {
        const int k = threadIdx.x + i * FT;
        px[2 * i] = px[2 * i + 1] = 0.0;
        if (k <= MB / 2) {
          const double2 a = buf[k], b = buf[MB - k];
          if (k == 0) {
            px[0] = (a.x + a.y) * (a.x + a.y);
            px[1] = (a.x - a.y) * (a.x - a.y);
            pvalid |= 3u;
          } else {
            const double er = 0.5 * (a.x + b.x), ei = 0.5 * (a.y - b.y);
            const double dr = 0.5 * (a.x - b.x), di = 0.5 * (a.y + b.y);
            const double2 wk = wh::ldg2(wpost + k);
            const double tr = fma(wk.x, di, wk.y * dr);
            const double ti = fma(wk.y, di, -(wk.x * dr));
            const double xr = er + tr, xi = ei + ti, yr = er - tr, yi = ti - ei;
            px[2 * i] = xr * xr + xi * xi;
            pvalid |= 1u << (2 * i);
            if (k != MB - k) {  // (the middle bin pairs with itself)
              px[2 * i + 1] = yr * yr + yi * yi;
              pvalid |= 2u << (2 * i);
            }
          }
        }
      }
    }
    STAGE_MARK(13)
    wh::sync<FT>();  // the spectrum has been read: the lower part of the buffer becomes the selection's work area
    STAGE_MARK(9)
    double s_small, s_total;
#if WH_D4C_ABLATE == 5  // (timing experiment: no rank selection)
    s_small = px[0];
    s_total = px[1] + 1.0 + (double)pvalid;
#else
    sum_smallest<K, FT, 2 * PJ>(px, pvalid, N / 2 - boundary, zr, scratch, &s_small, &s_total);
#endif
    if (threadIdx.x == 0) band[b] = -10 * log10(s_small / s_total);
    wh::sync<FT>();
  }

#if WH_D4C_ABLATE == 4
  if (threadIdx.x == 0) out[f * (int64_t)(k_spec > 0 ? k_spec : nap + 2)] = band[0];
  return;
#endif
  STAGE_MARK(5)
  // ---- outputs (d4c.py:56-59 / d4cRequiem.py:40) ---------------------------------------------
  const double tilt = (cf - 100) * 2 / 100;
  if (k_spec > 0) {
    if (coarse_dbg) for (int b = threadIdx.x; b < nap; b += FT) coarse_dbg[f * nap + b] = -fmax(0.0, band[b] - tilt);
    const wh::ckp<double> o = wh::ck_make(out + f * (int64_t)k_spec, k_spec, wh::WH_CK_OUT);
    const int nn = nap + 2;  // nodes: 0, interval, ..., interval*nap, fs/2
    // k * fs / (2 (K-1)): the divisor is a power of two for every FFT size, so k * (fs / divisor) is the same double
    // (both roundings are of the exact quotient) without a divide per bin
    const int qden = 2 * (k_spec - 1);
    const bool qexact = (qden & (qden - 1)) == 0;
    const double qstep = fs / (double)qden;
    for (int k = threadIdx.x; k < k_spec; k += FT) {
      // unfused like the reference's interpolation (d4c.py:60-62) whatever the translation unit's setting: with a 0 dB
      // band a fused slope * dx + y_lo can land an ulp ABOVE 0 dB, i.e. an aperiodicity above 1
#pragma clang fp contract(off)
      const double q = qexact ? (double)k * qstep : (double)k * fs / (double)qden;
      int cnt = 0;  // searchsorted-left over the coarse axis
      for (int m = 0; m < nn; ++m) {
        const double am = m <= nap ? (double)(m * interval) : fs / 2;
        cnt += (am < q) ? 1 : 0;
      }
      const int hi = cnt < 1 ? 1 : (cnt > nn - 1 ? nn - 1 : cnt);
      const int lo = hi - 1;
      const double a_lo = lo <= nap ? (double)(lo * interval) : fs / 2;
      const double a_hi = hi <= nap ? (double)(hi * interval) : fs / 2;
      const double y_lo = lo == 0 ? -60.0 : -fmax(0.0, band[lo - 1] - tilt);
      const double y_hi = hi == nn - 1 ? -0.000000000001 : -fmax(0.0, band[hi - 1] - tilt);
      const double slope = (y_hi - y_lo) / (a_hi - a_lo);
      const double db = slope * (q - a_lo) + y_lo;
      o[k] = exp(db * (M_LN10 / 20));  // 10^(db/20)
    }
  } else {
    const wh::ckp<double> o = wh::ck_make(out + f * (int64_t)(nap + 2), nap + 2, wh::WH_CK_OUT);
    for (int b = threadIdx.x; b < nap + 2; b += FT)
      o[b] = b == 0 ? -60.0 : (b == nap + 1 ? -0.000000000001 : -fmax(0.0, band[b - 1] - tilt));
  }
  STAGE_MARK(6)
}

template <int N, bool FUSED>
__global__ __launch_bounds__(ft_of(N), WH_D4C_PAIR > 1 ? WH_D4C_PAIR_MINW : minblk_of(N)) void d4c_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int32_t* __restrict__ frame_utt,
    const double* __restrict__ tp, double* __restrict__ f0_io, const double* __restrict__ vuv,
    const int32_t* __restrict__ gate, double threshold, double fs, int nap, int interval,
    const double* __restrict__ window, int wlen, const double2* __restrict__ tw_base, int k_spec,
    double* __restrict__ out, double* __restrict__ coarse_dbg, long long n_frames, D4cLaunchConst lc) {
#if WH_D4C_PAIR > 1
  // (experiment) WH_D4C_PAIR neighbouring frames per workgroup, one after the other
  const long long n_units = (n_frames + WH_D4C_PAIR - 1) / WH_D4C_PAIR;
  const long long unit = wh::xcd_unit(blockIdx.x, n_units);
  if (unit >= n_units) return;
#if defined(WH_D4C_PAIR_ROLLED) && WH_D4C_PAIR_ROLLED
#pragma unroll 1  // one copy of the frame's code (two copies are 130 KB: twice the instruction cache two CUs share)
#else
#pragma unroll
#endif
  for (int rep = 0; rep < WH_D4C_PAIR; ++rep) {
    d4c_frame<N, FUSED>(x, x_off, frame_utt, tp, f0_io, vuv, gate, threshold, fs, nap, interval, window, wlen, tw_base, k_spec,
                        out, coarse_dbg, n_frames, lc, unit * WH_D4C_PAIR + rep);
    __syncthreads();
  }
#else
  d4c_frame<N, FUSED>(x, x_off, frame_utt, tp, f0_io, vuv, gate, threshold, fs, nap, interval, window, wlen, tw_base, k_spec, out,
                      coarse_dbg, n_frames, lc, wh::xcd_unit(blockIdx.x, n_frames));
#endif
}

int pow2_at_least(double v) { return (int)llround(pow(2.0, ceil(log2(v)))); }

// Nuttall window of (possibly float-valued) length n, world/d4c.py:237-245.
std::vector<double> nuttall(int n) {
  std::vector<double> w(n);
  for (int i = 0; i < n; ++i) {
    const double t = (double)i * 2 * M_PI / (double)(n - 1);
    w[i] = 0.355768 * cos(0 * t) + -0.487396 * cos(t) + 0.144232 * cos(2 * t) + -0.012604 * cos(3 * t);
  }
  return w;
}

template <int NLT>
int launch_lt(wh_ctx* ctx, hipStream_t st, const wh_batch* b, const double* x, const double* tp, double* f0,
              const double* vuv, double fs, double thr, int32_t* gate) {
  const size_t lds = sizeof(double) * (NLT + 2 + 48 + kWinTab);  // 17 KB at 2048: the 66 VGPRs, not LDS, set the occupancy
  if (int rc = wh::allow_lds(&love_train_kernel<NLT>, lds)) return rc;
  { wh::KernelTimer _kt(ctx, st, "love_train_kernel"); hipLaunchKernelGGL(love_train_kernel<NLT>, dim3((unsigned)wh::xcd_grid(b->total_frames)), dim3(ft_love(NLT)), lds, st, x, b->d_x_off,
                     b->d_frame_utt, tp, f0, vuv, fs, thr, ctx->d_twiddle, gate, (long long)b->total_frames); }
  WH_LAUNCH_CHECK("love_train_kernel");
  return 0;
}

template <int N, bool FUSED>
int launch_main(wh_ctx* ctx, hipStream_t st, const wh_batch* b, const double* x, const double* tp, double* f0,
                const double* vuv, const int32_t* gate, double thr, double fs, int nap, int interval, const double* win,
                int wlen, int k_spec, double* out, double* coarse) {
  const size_t lds = sizeof(double) * (2 * N + 40 + 8 + 4 * kWinTab);
  if (int rc = wh::allow_lds(&d4c_kernel<N, FUSED>, lds)) return rc;
  { wh::KernelTimer _kt(ctx, st, "d4c_kernel"); hipLaunchKernelGGL((d4c_kernel<N, FUSED>), dim3((unsigned)wh::xcd_grid((b->total_frames + WH_D4C_PAIR - 1) / WH_D4C_PAIR)), dim3(ft_of(N)), lds, st, x, b->d_x_off,
                     b->d_frame_utt, tp, f0, vuv, gate, thr, fs, nap, interval, win, wlen, ctx->d_twiddle, k_spec, out,
                     coarse, (long long)b->total_frames, d4c_launch_const(fs, N, wlen, interval, nap)); }
  WH_LAUNCH_CHECK("d4c_kernel");
  return 0;
}

int d4c_common(wh_ctx* ctx, void* stream, const wh_batch* b, const double* x, const double* tp, double* f0,
               const double* vuv, double fs, double threshold, int nfft, int interval, int k_spec, double* out,
               double* coarse) {
  hipStream_t st = (hipStream_t)stream;
  if (b->total_frames == 0) return 0;
  const int nap = (int)floor(fmin(15000.0, fs / 2 - interval) / interval);
  if (nap <= 0) return wh::fail_msg("wh_d4c", "sampling rate too low: no aperiodicity band (reference asserts, d4c.py:35)");
  if (nap > 8) return wh::fail_msg("wh_d4c", "more than 8 aperiodicity bands unsupported");
  const int nlt = pow2_at_least(3 * fs / 40.0 + 1);
  const int wlen = (int)(floor(interval / (fs / nfft)) * 2 + 1);
  // workspace: gate[F]; the band window is a cached constant table
  if (int rc = wh::ws_reserve(ctx, (size_t)b->total_frames * sizeof(int32_t))) return rc;
  int32_t* gate = reinterpret_cast<int32_t*>(ctx->ws);
  const double* d_win = nullptr;
  if (int rc = wh::const_table(ctx, "nuttall:" + std::to_string(wlen), nuttall(wlen), &d_win)) return rc;
  if (nlt == nfft) {  // fused love-train + D4C
    switch (nfft) {
      case 512: return launch_main<512, true>(ctx, st, b, x, tp, f0, vuv, gate, threshold, fs, nap, interval, d_win, wlen, k_spec, out, coarse);
      case 1024: return launch_main<1024, true>(ctx, st, b, x, tp, f0, vuv, gate, threshold, fs, nap, interval, d_win, wlen, k_spec, out, coarse);
      case 2048: return launch_main<2048, true>(ctx, st, b, x, tp, f0, vuv, gate, threshold, fs, nap, interval, d_win, wlen, k_spec, out, coarse);
      case 4096: return launch_main<4096, true>(ctx, st, b, x, tp, f0, vuv, gate, threshold, fs, nap, interval, d_win, wlen, k_spec, out, coarse);
      case 8192: return launch_main<8192, true>(ctx, st, b, x, tp, f0, vuv, gate, threshold, fs, nap, interval, d_win, wlen, k_spec, out, coarse);
      default: return wh::fail_msg("wh_d4c", "D4C FFT size outside [512, 8192]");
    }
  }
  int rc;
  switch (nlt) {
    case 512: rc = launch_lt<512>(ctx, st, b, x, tp, f0, vuv, fs, threshold, gate); break;
    case 1024: rc = launch_lt<1024>(ctx, st, b, x, tp, f0, vuv, fs, threshold, gate); break;
    case 2048: rc = launch_lt<2048>(ctx, st, b, x, tp, f0, vuv, fs, threshold, gate); break;
    case 4096: rc = launch_lt<4096>(ctx, st, b, x, tp, f0, vuv, fs, threshold, gate); break;
    case 8192: rc = launch_lt<8192>(ctx, st, b, x, tp, f0, vuv, fs, threshold, gate); break;
    default: return wh::fail_msg("wh_d4c", "love-train FFT size outside [512, 8192] (fs must be <= ~109 kHz)");
  }
  if (rc) return rc;
  switch (nfft) {
    case 512: return launch_main<512, false>(ctx, st, b, x, tp, f0, vuv, gate, threshold, fs, nap, interval, d_win, wlen, k_spec, out, coarse);
    case 1024: return launch_main<1024, false>(ctx, st, b, x, tp, f0, vuv, gate, threshold, fs, nap, interval, d_win, wlen, k_spec, out, coarse);
    case 2048: return launch_main<2048, false>(ctx, st, b, x, tp, f0, vuv, gate, threshold, fs, nap, interval, d_win, wlen, k_spec, out, coarse);
    case 4096: return launch_main<4096, false>(ctx, st, b, x, tp, f0, vuv, gate, threshold, fs, nap, interval, d_win, wlen, k_spec, out, coarse);
    case 8192: return launch_main<8192, false>(ctx, st, b, x, tp, f0, vuv, gate, threshold, fs, nap, interval, d_win, wlen, k_spec, out, coarse);
    default: return wh::fail_msg("wh_d4c", "D4C FFT size outside [512, 8192]");
  }
}

}  // namespace

#ifdef WH_D4C_STAGE_TIMER
extern "C" int wh_debug_d4c_stages(unsigned long long* out16, int reset) {
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_d4c_stage), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_d4c_stage), z, sizeof(z));
  }
  return 0;
}
#endif

extern "C" int wh_d4c_bands(double fs, int requiem) {
  const int interval = (!requiem && fs < 16000) ? 2000 : 3000;
  return (int)floor(fmin(15000.0, fs / 2 - interval) / interval);
}

extern "C" int wh_d4c(wh_ctx* ctx, void* stream, const wh_batch* b, const double* x, const double* tp, double* f0,
                      const double* vuv, double fs, double threshold, int fft_size_for_spectrum, double* aperiodicity,
                      double* coarse_ap) {
  if (!ctx || !b || !x || !tp || !f0 || !vuv || !aperiodicity) return wh::fail_msg("wh_d4c", "null argument");
  WH_ENTER(ctx);
  if (fft_size_for_spectrum < 2) return wh::fail_msg("wh_d4c", "fft_size_for_spectrum must be >= 2");
  const int nfft = pow2_at_least(4 * fs / 47.0 + 1);
  const int interval = fs < 16000 ? 2000 : 3000;
  return d4c_common(ctx, stream, b, x, tp, f0, vuv, fs, threshold, nfft, interval, fft_size_for_spectrum / 2 + 1,
                    aperiodicity, coarse_ap);
}

extern "C" int wh_d4c_requiem(wh_ctx* ctx, void* stream, const wh_batch* b, const double* x, const double* tp,
                              double* f0, const double* vuv, double fs, double threshold, int fft_size,
                              double* band_aperiodicity) {
  if (!ctx || !b || !x || !tp || !f0 || !vuv || !band_aperiodicity) return wh::fail_msg("wh_d4c_requiem", "null argument");
  WH_ENTER(ctx);
  const int nfft = fft_size > 0 ? fft_size : pow2_at_least(3 * fs / 47.0 + 1);
  return d4c_common(ctx, stream, b, x, tp, f0, vuv, fs, threshold, nfft, 3000, 0, band_aperiodicity, nullptr);
}
