// Device-side building blocks shared by every WORLD kernel (gfx950 / CDNA4, wave64).
//
//  * block_sum*            : workgroup reductions built from 64-lane wave shuffles plus one LDS hop across
//    the waves; xcd_unit: workgroup id -> unit mapping that keeps neighbouring units on one XCD (one L2).
//  * fft_lds<N>            : in-place complex FP64 Stockham FFT on an LDS-resident buffer.  Every pass pulls
//    its radix-8 / 4 / 2 operands into registers, barriers, and writes the auto-sorted outputs back into the
//    same buffer, so no ping-pong copy is needed and a 4096-point transform fits 64 KiB of the CU's 160 KiB
//    LDS; intermediate layouts are XOR-swizzled against store bank conflicts; fft_lds_from_regs feeds the
//    first pass from registers; rfft_lds / irfft_lds do real transforms through half-size complex ones.
//
// All arithmetic is FP64: the reference is float64 end to end and the F0 stages take discrete
// decisions on it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WH_BLOCK 256
#define WH_WAVE 64
// Thread index as the FFT / reduction helpers see it.  A translation unit whose kernel LOOPS over units may define
// WH_TID as an opaque read of threadIdx.x (see wh_synthesis.hip): the helpers' per-thread addresses then stop being
// loop invariants that the compiler hoists and keeps in registers across the whole loop body.
#ifndef WH_TID
#define WH_TID threadIdx.x
#endif

namespace wh {

// Loads that are known to address global memory, said so to the compiler.  Kernel-argument pointers are inferred as
// global on their own, but not once they have been laundered through an empty asm (fresh_table, stage fences), selected
// at run time, or read out of a by-value argument struct: loads through those compile to flat_load, which ticks BOTH
// the vector-memory and the LDS counter — every s_waitcnt lgkmcnt(0) in front of an LDS read then also waits for the
// twiddle fetch that was issued early precisely to overlap with it.
__device__ __forceinline__ double2 ldg2(const double2* p) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  const v2d v = *(const v2d __attribute__((address_space(1)))*)p;
  return make_double2(v.x, v.y);
}
__device__ __forceinline__ double ldg(const double* p) { return *(const double __attribute__((address_space(1)))*)p; }
#if WH_BOUNDS
template <class T> struct ckp;
__device__ __forceinline__ double2 ldg2(const ckp<const double2>& p);
__device__ __forceinline__ double ldg(const ckp<const double>& p);
#endif
// The same for stores (a pointer read out of a job record is generic to the compiler: flat_store), plain and non-temporal.
__device__ __forceinline__ void stg(double* p, double v) { *(double __attribute__((address_space(1)))*)p = v; }
__device__ __forceinline__ void stg_nt(double* p, double v) {
  __builtin_nontemporal_store(v, (double __attribute__((address_space(1)))*)p);
}

// ------------------------------------------------------------------------------------------
// Checked pointers: the bounds build (-DWH_BOUNDS=1, tools/build_variants.py; never shipped)
// ------------------------------------------------------------------------------------------
// This image cannot run device AddressSanitizer (no instrumented ROCm runtime, XNACK off), and round 5's attempt left an
// abort of the instrumented cheaptrick_kernel that could not be read.  The deterministic replacement: wh::ckp<T> is T*
// in every normal build — the alias, so the shipped code is the code without it, instruction for instruction — and in
// the bounds build a pointer that carries the element range it may touch.  Every [] / * through it is compared with that
// range; the first access outside is recorded (buffer tag, element index, range size) in a device word that
// wh_take_flags reads: WH_FLAG_OOB with the record behind wh_bounds_last(), and the access itself is redirected to the
// range's first element, so the kernel runs on instead of faulting.  The kernels that are covered name their buffers with
// ck_make / ck_sub / ck_as; helpers take ckp<T> parameters; a raw pointer handed to such a helper converts implicitly
// to an UNCHECKED ckp (kernels not yet converted keep compiling).
#ifndef WH_BOUNDS
#define WH_BOUNDS 0
#endif
enum {  // buffer tags of the record
  WH_CK_LDS_MAIN = 1, WH_CK_LDS_AUX = 2, WH_CK_LDS_SCRATCH = 3, WH_CK_TWIDDLE = 4, WH_CK_WAVEFORM = 5, WH_CK_OUT = 6,
  WH_CK_TABLE = 7, WH_CK_LDS_OTHER = 8, WH_CK_IN = 9
};
#if WH_BOUNDS
#define WH_RESTRICT
// first out-of-range access by this translation unit's kernels since the last read: count, tag, index, size
static __device__ unsigned long long g_oob[4];
__device__ __forceinline__ void oob_report(int tag, long long index, long long size) {
  if (atomicAdd(&g_oob[0], 1ull) == 0ull) {
    g_oob[1] = (unsigned long long)tag;
    g_oob[2] = (unsigned long long)index;
    g_oob[3] = (unsigned long long)size;
  }
}
template <class T>
struct ckp {
  T* p = nullptr;
  T* base = nullptr;  // element 0 of the range
  long long n = -1;   // elements in the range; < 0: unchecked
  int tag = 0;
  __host__ __device__ ckp() {}
  __host__ __device__ ckp(T* q) : p(q), base(q), n(-1), tag(0) {}  // a raw pointer: unchecked
  __host__ __device__ ckp(T* q, T* b, long long nn, int tg) : p(q), base(b), n(nn), tag(tg) {}
  template <class U, class = decltype(static_cast<T*>(static_cast<U*>(nullptr)))>
  __host__ __device__ ckp(const ckp<U>& o) : p(o.p), base(o.base), n(o.n), tag(o.tag) {}  // T* -> const T*
  __device__ __forceinline__ T* at(long long i) const {
    if (n >= 0) {
      const long long k = (p - base) + i;
      if (k < 0 || k >= n) {
        oob_report(tag, k, n);
        return base;
      }
    }
    return p + i;
  }
  __device__ __forceinline__ T& operator[](long long i) const { return *at(i); }
  __device__ __forceinline__ T& operator*() const { return *at(0); }
  __host__ __device__ ckp operator+(long long k) const { return ckp(p + k, base, n, tag); }
  __host__ __device__ ckp operator-(long long k) const { return ckp(p - k, base, n, tag); }
  __host__ __device__ explicit operator bool() const { return p != nullptr; }
};
template <class T>
__host__ __device__ __forceinline__ ckp<T> ck_make(T* q, long long n, int tag) { return ckp<T>(q, q, n, tag); }
// elements [off, off + n) of p's range as a range of its own
template <class T>
__host__ __device__ __forceinline__ ckp<T> ck_sub(ckp<T> p, long long off, long long n, int tag) {
  return ckp<T>(p.p + off, p.p + off, n, tag);
}
// the same bytes seen as U (the range is re-expressed in elements of U)
template <class U, class T>
__host__ __device__ __forceinline__ ckp<U> ck_as(ckp<T> p) {
  U* b = reinterpret_cast<U*>(p.base);
  const long long n = p.n < 0 ? -1 : (long long)((p.n * (long long)sizeof(T)) / (long long)sizeof(U));
  return ckp<U>(reinterpret_cast<U*>(p.p), b, n, p.tag);
}
template <class T>
__host__ __device__ __forceinline__ T* ck_raw(ckp<T> p) { return p.p; }
// host side: this translation unit's record, read and cleared (registered with wh_api.hip at load time)
int bounds_register(int (*reader)(unsigned long long*));
static int bounds_reader_tu(unsigned long long* out4) {
  const unsigned long long z[4] = {0ull, 0ull, 0ull, 0ull};
  if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_oob), sizeof z) != hipSuccess) return 1;
  if (out4[0] && hipMemcpyToSymbol(HIP_SYMBOL(g_oob), z, sizeof z) != hipSuccess) return 1;
  return 0;
}
static const int bounds_registered_tu = bounds_register(&bounds_reader_tu);
__device__ __forceinline__ double2 ldg2(const ckp<const double2>& p) { return ldg2(static_cast<const double2*>(p.at(0))); }
__device__ __forceinline__ double ldg(const ckp<const double>& p) { return ldg(static_cast<const double*>(p.at(0))); }
#else
#define WH_RESTRICT __restrict__
template <class T>
using ckp = T*;
template <class T>
__host__ __device__ __forceinline__ T* ck_make(T* q, long long, int) { return q; }
template <class T>
__host__ __device__ __forceinline__ T* ck_sub(T* p, long long off, long long, int) { return p + off; }
template <class U, class T>
__host__ __device__ __forceinline__ U* ck_as(T* p) { return reinterpret_cast<U*>(p); }
template <class T>
__host__ __device__ __forceinline__ T* ck_raw(T* p) { return p; }
#endif

// Workgroup-wide synchronisation for NT cooperating threads.  A single-wave group (NT == 64) needs no
// hardware barrier: its lanes run in lockstep, so a compiler-level wavefront fence is enough to order the
// LDS traffic.  This is what lets the per-frame kernels run as one wave per frame with zero s_barrier.
template <int NT>
__device__ __forceinline__ void sync() {
  if constexpr (NT <= WH_WAVE) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

// Workgroups are dealt to the 8 XCDs of the MI355X round-robin by linear id, and every XCD has its own L2.  The
// per-frame / per-pulse kernels read overlapping neighbourhoods (a 5 ms hop against 20-40 ms analysis windows; the
// two spectrogram rows a pulse interpolates are its neighbour's too), so unit u = consecutive ids would put every
// neighbourhood into all eight L2s.  This mapping gives XCD x the contiguous range [x*ceil(n/8), (x+1)*ceil(n/8)).
// Returns n (out of range) for the padding ids of the last rows.
#ifndef WH_XCD_REMAP
#define WH_XCD_REMAP 1
#endif
__device__ __forceinline__ long long xcd_unit(long long id, long long n) {
#if WH_XCD_REMAP
  constexpr int kXcds = 8;
  const long long per = (n + kXcds - 1) / kXcds;
  const long long u = (id % kXcds) * per + id / kXcds;
  return (id / kXcds < per && u < n) ? u : n;
#else
  return id < n ? id : n;
#endif
}
__host__ __device__ __forceinline__ long long xcd_grid(long long n) { return ((n + 7) / 8) * 8; }

// Barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter, so a global load
// issued before it (the FFT passes fetch their twiddles ahead of the barrier) would have to land before any wave
// may pass; with the fences restricted to the local address space the load stays in flight across the barrier.
// Only valid where the threads hand each other LDS contents and nothing else.
template <int NT>
__device__ __forceinline__ void sync_lds() {
  if constexpr (NT <= WH_WAVE) {
    sync<NT>();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  }
}

// A lane-private serial recurrence over the samples [i0, i1) (the chunked IIR passes): the recurrence itself costs
// ~40 cycles per sample, a dependent global load per sample ~1500, and the lanes of a wave sit a whole chunk apart so
// nothing coalesces.  The samples are therefore fetched in blocks of B independent loads, one block ahead of the
// block the recurrence is consuming.  interior(i) says that fast(i) .. fast(i + B - 1) are plain loads (no edge
// extension); edge blocks go through slow().  body(i, v) runs in ascending i.
template <int B, class Interior, class Fast, class Slow, class Body>
__device__ __forceinline__ void serial_run(int64_t i0, int64_t i1, Interior interior, Fast fast, Slow slow, Body body) {
  double cur[B], nxt[B];
  auto fetch = [&](double (&v)[B], int64_t i) {
    if (i + B > i1) return;
    if (interior(i)) {
#pragma unroll
      for (int k = 0; k < B; ++k) v[k] = fast(i + k);
    } else {
#pragma unroll
      for (int k = 0; k < B; ++k) v[k] = slow(i + k);
    }
  };
  int64_t i = i0;
  fetch(cur, i);
  for (; i + B <= i1; i += B) {
    fetch(nxt, i + B);
#pragma unroll
    for (int k = 0; k < B; ++k) body(i + k, cur[k]);
#pragma unroll
    for (int k = 0; k < B; ++k) cur[k] = nxt[k];
  }
  for (; i < i1; ++i) body(i, slow(i));
}

// Sum over the 64 lanes of a wave, result in every lane.  DPP form (default): butterflies inside the rows of 16 lanes by
// quad permutes and row mirrors, then row_bcast:15 / row_bcast:31 carry the row totals up to lane 63, whose value is
// broadcast through the scalar unit — 12 v_mov_dpp + 6 v_add_f64 + 2 v_readlane, all on the VALU.  The shuffle form
// (WH_WAVE_SUM_DPP=0) is 12 ds_bpermute_b32 through the LDS crossbar with a wait in front of every add; the window
// reductions of d4c_kernel run five of these sums four times per frame.  (tools/ubench/wave_sum_check.hip)
#ifndef WH_WAVE_SUM_DPP
#define WH_WAVE_SUM_DPP 1
#endif
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_or_zero(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);  // lanes outside the mask / without a source: 0
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
// a permutation inside the rows of 16 lanes: every lane has a source, so there is no "old" value to prepare (with
// update_dpp(0, ..) each of these steps carried two v_mov_b32 v, 0 in front of its two v_mov_b32_dpp)
template <int CTRL>
__device__ __forceinline__ double dpp_row_perm(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
#if WH_WAVE_SUM_DPP
  v += dpp_row_perm<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_row_perm<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_row_perm<0x141>(v);  // row_half_mirror
  v += dpp_row_perm<0x140>(v);  // row_mirror: every lane holds its row's total
  v += dpp_or_zero<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_or_zero<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's total
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
#else
#pragma unroll
  for (int o = WH_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WH_WAVE);
  return v;
#endif
}

// Sum over the whole 256-thread block; result broadcast to every thread.
// `scratch` must hold >= 3*NT/64 doubles of LDS (24 for the largest block used, 512).  Contains two barriers.
template <int NT = WH_BLOCK>
__device__ __forceinline__ double block_sum(double v, ckp<double> scratch) {
  v = wave_sum(v);
  if constexpr (NT <= WH_WAVE) {
    sync<NT>();
    return v;
  }
  const int w = WH_TID >> 6;
  __syncthreads();
  if ((WH_TID & 63) == 0) scratch[w] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < NT / WH_WAVE; ++i) t += scratch[i];
  return t;
}

// Two sums at once (saves barriers).
template <int NT = WH_BLOCK>
__device__ __forceinline__ void block_sum2(double& a, double& b, ckp<double> scratch) {
  a = wave_sum(a);
  b = wave_sum(b);
  if constexpr (NT <= WH_WAVE) {
    sync<NT>();
    return;
  }
  const int w = WH_TID >> 6;
  __syncthreads();
  if ((WH_TID & 63) == 0) {
    scratch[w] = a;
    scratch[NT / WH_WAVE + w] = b;
  }
  __syncthreads();
  double ta = 0.0, tb = 0.0;
#pragma unroll
  for (int i = 0; i < NT / WH_WAVE; ++i) {
    ta += scratch[i];
    tb += scratch[NT / WH_WAVE + i];
  }
  a = ta;
  b = tb;
}

template <int NT = WH_BLOCK>
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, ckp<double> scratch) {
  a = wave_sum(a);
  b = wave_sum(b);
  c = wave_sum(c);
  if constexpr (NT <= WH_WAVE) {
    sync<NT>();
    return;
  }
  const int w = WH_TID >> 6;
  __syncthreads();
  if ((WH_TID & 63) == 0) {
    scratch[w] = a;
    scratch[NT / WH_WAVE + w] = b;
    scratch[2 * (NT / WH_WAVE) + w] = c;
  }
  __syncthreads();
  double ta = 0.0, tb = 0.0, tc = 0.0;
#pragma unroll
  for (int i = 0; i < NT / WH_WAVE; ++i) {
    ta += scratch[i];
    tb += scratch[NT / WH_WAVE + i];
    tc += scratch[2 * (NT / WH_WAVE) + i];
  }
  a = ta;
  b = tb;
  c = tc;
}

// Five sums at once (the D4C window: two means and the three second moments of its energy, one pair of barriers).
// `scratch` must hold >= 5*NT/64 doubles.
template <int NT = WH_BLOCK>
__device__ __forceinline__ void block_sum5(double& a, double& b, double& c, double& d, double& e, ckp<double> scratch) {
  a = wave_sum(a);
  b = wave_sum(b);
  c = wave_sum(c);
  d = wave_sum(d);
  e = wave_sum(e);
  if constexpr (NT <= WH_WAVE) {
    sync<NT>();
    return;
  }
  constexpr int NW = NT / WH_WAVE;
  const int w = WH_TID >> 6;
  __syncthreads();
  if ((WH_TID & 63) == 0) {
    scratch[w] = a;
    scratch[NW + w] = b;
    scratch[2 * NW + w] = c;
    scratch[3 * NW + w] = d;
    scratch[4 * NW + w] = e;
  }
  __syncthreads();
  double t[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < NW; ++i)
#pragma unroll
    for (int k = 0; k < 5; ++k) t[k] += scratch[k * NW + i];
  a = t[0];
  b = t[1];
  c = t[2];
  d = t[3];
  e = t[4];
}

__device__ __forceinline__ double wave_scan_incl(double v) {
  const int lane = WH_TID & 63;
#pragma unroll
  for (int o = 1; o < WH_WAVE; o <<= 1) {
    double u = __shfl_up(v, o, WH_WAVE);
    if (lane >= o) v += u;
  }
  return v;
}

// ------------------------------------------------------------------------------------------
// FFT
// ------------------------------------------------------------------------------------------
// complex multiply with fused multiply-adds (2 mul + 2 fma instead of 4 mul + 2 add)
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}
// a * conj(b): the same roundings as cmul(a, (b.x, -b.y)), the negations riding on the operands
__device__ __forceinline__ double2 cmul_conj(double2 a, double2 b) {
  return make_double2(fma(a.x, b.x, a.y * b.y), fma(-a.x, b.y, a.y * b.x));
}

// In-place Stockham passes of radix 8 (then 4 or 2 for what is left of N), NT threads on one N-point buffer.
//
// LDS cost model (MI355X_MICROARCH.md, LDS): a ds_write_b128 costs ~13 cycles per wave against 4 for a
// ds_read_b128, and stores are serviced in groups of 8 consecutive lanes over a 128-byte bank row — so the
// transforms are priced in *stores*: radix 8 needs 4 passes for 2048 points where radix 4 needs 6, and the
// scattered stores of the early passes (lane stride R complex values: every lane of a group on the same 16-byte
// slot) are made conflict-free by an XOR swizzle of the intermediate layout, element i living at
// i ^ ((i >> 3) & 7).  The swizzle is internal: the first pass reads and the last pass writes natural order.
//
// tw[i] = exp(-2*pi*i*sqrt(-1)/N), i in [0,N).  INV conjugates twiddles and butterflies.
// SNT >= NT: the barrier spans SNT threads while NT of them (thread index modulo NT) cooperate on this buffer,
// so that SNT/NT independent transforms on different buffers advance in lockstep through the same barriers.
#ifndef WH_FFT_TW_PREFETCH
#define WH_FFT_TW_PREFETCH 1
#endif
#ifndef WH_FFT_SWZ
#define WH_FFT_SWZ 1  // 0: natural intermediate layout everywhere (stores at immediate offsets, bank conflicts in the early passes)
#endif
__device__ __forceinline__ int fft_swz(int i) { return i ^ ((i >> 3) & 7); }

__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward) / +i (inverse)
template <bool INV>
__device__ __forceinline__ double2 crot(double2 a) {
  return INV ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x);
}

// R-point DFT of v[0..R) in place, natural order out.
template <int R, bool INV>
__device__ __forceinline__ void dft_small(double2 (&v)[R]) {
  if constexpr (R == 2) {
    const double2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  } else if constexpr (R == 4) {
    const double2 e0 = cadd(v[0], v[2]), e1 = csub(v[0], v[2]);
    const double2 e2 = cadd(v[1], v[3]), e3 = crot<INV>(csub(v[1], v[3]));
    v[0] = cadd(e0, e2);
    v[1] = cadd(e1, e3);
    v[2] = csub(e0, e2);
    v[3] = csub(e1, e3);
  } else {
    static_assert(R == 8, "radix");
    constexpr double h = 0.70710678118654752440;
    double2 a[4], b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a[r] = cadd(v[r], v[r + 4]);
      b[r] = csub(v[r], v[r + 4]);
    }
    // b1 *= W8, b2 *= W8^2 = -i, b3 *= W8^3   (W8 = exp(-i*pi/4); conjugated for the inverse)
    b[1] = INV ? make_double2(h * (b[1].x - b[1].y), h * (b[1].x + b[1].y))
               : make_double2(h * (b[1].x + b[1].y), h * (b[1].y - b[1].x));
    b[2] = crot<INV>(b[2]);
    b[3] = INV ? make_double2(-h * (b[3].x + b[3].y), h * (b[3].x - b[3].y))
               : make_double2(h * (b[3].y - b[3].x), -h * (b[3].x + b[3].y));
    dft_small<4, INV>(a);
    dft_small<4, INV>(b);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[2 * r] = a[r];
      v[2 * r + 1] = b[r];
    }
  }
}

// Radix of the pass that follows NS: 8 while that leaves at least half of the NT threads a butterfly (a 512-point
// transform on 256 threads is faster as 4-4-4-4-2 on 128 lanes than as 8-8-8 on 64: its passes are latency-,
// not throughput-bound), else 4, else 2.
// MAXR caps the radix for a kernel whose register budget is set elsewhere (radix-4 butterflies hold half as many
// operands): d4c_kernel at 128 VGPRs takes the radix-8 plan up to N = 1024 and radix 4 beyond (wh_d4c.hip).
template <int N, int NT, int NS, int MAXR = 8>
struct FftRadix {
  static constexpr int value = (MAXR >= 8 && NS * 8 <= N && N / 8 >= NT / 2) ? 8 : (NS * 4 <= N ? 4 : 2);
};

// Twiddle, butterfly and store of one pass for the butterflies this thread owns: v[p][r] = element
// j + r*(N/R), j = tid + p*NT.
template <int N, int NT, int R, int NS, bool INV, bool SWZ_OUT>
__device__ __forceinline__ void fft_pass_finish(ckp<double2> WH_RESTRICT s, double2 (&v)[(N / R + NT - 1) / NT][R],
                                                const double2 (&w)[(N / R + NT - 1) / NT][R]) {
  constexpr int J = N / R;
  constexpr int PER = (J + NT - 1) / NT;
  const int tid = WH_TID & (NT - 1);
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    const int j = tid + p * NT;
    if (J % NT == 0 || j < J) {
      const int k = j & (NS - 1);
      if (NS > 1) {
#pragma unroll
        for (int r = 1; r < R; ++r) v[p][r] = INV ? cmul_conj(v[p][r], w[p][r]) : cmul(v[p][r], w[p][r]);
      }
#if defined(WH_ABLATE_FIRSTPASS) && WH_ABLATE_FIRSTPASS
      if (NS != 1)  // TIMING EXPERIMENT ONLY (wrong results): the first pass's butterflies cost nothing — an upper bound of
#endif           // what pruning them for zero-padded inputs could buy
      dft_small<R, INV>(v[p]);
      const int base = (j - k) * R + k;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int o = base + r * NS;
        s[SWZ_OUT ? (NS % 64 == 0 ? fft_swz(base) + r * NS : fft_swz(o)) : o] = v[p][r];
      }
    }
  }
}

// The twiddles of one pass (global table, L1/L2 resident): fetched before the pass's barrier so that their
// latency is spent waiting for the other waves.
template <int N, int NT, int R, int NS, bool INV>
__device__ __forceinline__ void fft_pass_twiddles(ckp<const double2> WH_RESTRICT tw, double2 (&w)[(N / R + NT - 1) / NT][R]) {
  constexpr int J = N / R;
  constexpr int PER = (J + NT - 1) / NT;
  const int tid = WH_TID & (NT - 1);
  if (NS > 1) {
    constexpr int STEP = N / (NS * R);
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int j = tid + p * NT;
      if (J % NT == 0 || j < J) {
        const int k = j & (NS - 1);
#pragma unroll
        for (int r = 1; r < R; ++r) {
          // as stored: the inverse transform's conjugation is folded into the multiply (fft_pass_finish) — negating
          // here made every load's wait come right behind it, in front of the pass's LDS reads instead of under them
          w[p][r] = ldg2(tw + ((k * r * STEP) & (N - 1)));
        }
      }
    }
  }
}

template <int N, int NT, int R, int NS, bool INV, int SNT, bool SWZ_IN, bool SWZ_OUT>
__device__ __forceinline__ void fft_pass(ckp<double2> WH_RESTRICT s, ckp<const double2> WH_RESTRICT tw) {
  constexpr int J = N / R;
  constexpr int PER = (J + NT - 1) / NT;
  double2 v[PER][R], w[PER][R];
  const int tid = WH_TID & (NT - 1);
#if WH_FFT_TW_PREFETCH
  fft_pass_twiddles<N, NT, R, NS, INV>(tw, w);
#endif
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    const int j = tid + p * NT;
    if (J % NT == 0 || j < J) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        v[p][r] = s[SWZ_IN ? (J % 64 == 0 ? fft_swz(j) + r * J : fft_swz(j + r * J)) : j + r * J];
    }
  }
  sync_lds<SNT>();
#if !WH_FFT_TW_PREFETCH
  fft_pass_twiddles<N, NT, R, NS, INV>(tw, w);
#endif
  fft_pass_finish<N, NT, R, NS, INV, SWZ_OUT>(s, v, w);
  sync_lds<SNT>();
}

template <int N, int NT, int NS, bool INV, int SNT = NT, int MAXR = 8>
__device__ __forceinline__ void fft_passes(ckp<double2> s, ckp<const double2> tw) {
  if constexpr (NS < N) {
    constexpr int R = FftRadix<N, NT, NS, MAXR>::value;
    constexpr bool SWZ = WH_FFT_SWZ && N >= 64 && MAXR >= 8;  // radix-4 plans keep the natural layout (one address VGPR per pass)
    fft_pass<N, NT, R, NS, INV, SNT, SWZ && (NS > 1), SWZ && (NS * R < N)>(s, tw);
    fft_passes<N, NT, NS * R, INV, SNT, MAXR>(s, tw);
  }
}

// In-place unnormalised DFT of N complex doubles resident in LDS.  Caller guarantees that the
// buffer is fully written and visible (barrier) on entry; visible on exit.  The inverse does
// NOT divide by N.
template <int N, bool INV, int NT = WH_BLOCK, int SNT = NT, int MAXR = 8>
__device__ __forceinline__ void fft_lds(ckp<double2> s, ckp<const double2> tw) {
  fft_passes<N, NT, 1, INV, SNT, MAXR>(s, tw);
}

// The same transform with the input still in registers: x[q] = element tid + q*NT, q < N/NT (the layout a
// thread-strided producer loop leaves behind).  When N >= R*NT (R the first radix) those are exactly the operands
// of this thread's first-pass butterflies, so the input never makes the trip through LDS.  The buffer must be
// free (no other thread still reading it): a barrier is taken on entry.
template <int N, bool INV, int NT = WH_BLOCK, int MAXR = 8>
__device__ __forceinline__ void fft_lds_from_regs(const double2 (&x)[N / NT], ckp<double2> s, ckp<const double2> tw) {
  constexpr int R = FftRadix<N, NT, 1, MAXR>::value;
  static_assert(N % (R * NT) == 0 && N >= 64, "register-fed first pass needs N >= R*NT");
  constexpr int PER = N / R / NT;
  double2 v[PER][R], w[PER][R];
#pragma unroll
  for (int p = 0; p < PER; ++p)
#pragma unroll
    for (int r = 0; r < R; ++r) v[p][r] = x[p + r * PER];
  sync_lds<NT>();
  fft_pass_finish<N, NT, R, 1, INV, (WH_FFT_SWZ && R < N && MAXR >= 8)>(s, v, w);
  sync_lds<NT>();
  fft_passes<N, NT, R, INV, NT, MAXR>(s, tw);
}

// ------------------------------------------------------------------------------------------
// Real-input / real-output transforms through a half-size complex FFT
// ------------------------------------------------------------------------------------------
// Every WORLD transform is of real data or produces real data, so an N-point transform is done as an
// N/2-point complex FFT on the sample pairs (x[2j], x[2j+1]) plus one O(N) butterfly pass: half the
// butterflies and half the LDS of a complex N-point FFT.  tw_base is the context's table base: the table of
// size M (exp(-2*pi*i*k/M), k < M) lives at tw_base + M.

// Forward.  in: z[j] = (x[2j], x[2j+1]), j < N/2 (i.e. the real array itself).  out: z[k] = X[k], k = 0..N/2
// (N/2 + 1 entries).  Buffer must be visible on entry; visible on exit.
template <int N, int NT = WH_BLOCK, int SNT = NT, int MAXR = 8>
__device__ __forceinline__ void rfft_lds(ckp<double2> z, ckp<const double2> WH_RESTRICT tw_base) {
  fft_lds<N / 2, false, NT, SNT, MAXR>(z, tw_base + N / 2);
  ckp<const double2> WH_RESTRICT w = tw_base + N;
  for (int k = WH_TID & (NT - 1); k <= N / 4; k += NT) {
    if (k == 0) {
      const double2 a = z[0];
      z[0] = make_double2(a.x + a.y, 0.0);
      z[N / 2] = make_double2(a.x - a.y, 0.0);
    } else {
      const double2 a = z[k], b = z[N / 2 - k];
      const double er = 0.5 * (a.x + b.x), ei = 0.5 * (a.y - b.y);  // E = (A + conj(B))/2   (even samples)
      const double dr = 0.5 * (a.x - b.x), di = 0.5 * (a.y + b.y);  // D = (A - conj(B))/2 ; O = -i*D (odd samples)
      const double2 wk = ldg2(w + k);
      const double tr = fma(wk.x, di, wk.y * dr);   // T = W^k * O,  O = (di, -dr)
      const double ti = fma(wk.y, di, -(wk.x * dr));
      z[k] = make_double2(er + tr, ei + ti);
      z[N / 2 - k] = make_double2(er - tr, ti - ei);  // conj(E - T)
    }
  }
  sync_lds<SNT>();
}

// Inverse.  in: z[k] = X[k], k = 0..N/2: the half spectrum; the result is Re(IDFT) of its Hermitian extension
// (imaginary parts of the DC / Nyquist bins are ignored, as taking .real of a full complex IFFT would).
// out: z[j] = N * (x[2j], x[2j+1]), j < N/2 (unnormalised like fft_lds<.., true>: divide by N).
template <int N, int NT = WH_BLOCK, int SNT = NT, int MAXR = 8>
__device__ __forceinline__ void irfft_lds(ckp<double2> z, ckp<const double2> WH_RESTRICT tw_base) {
  ckp<const double2> WH_RESTRICT w = tw_base + N;
  for (int k = WH_TID & (NT - 1); k <= N / 4; k += NT) {
    double2 a = z[k], b = z[N / 2 - k];
    if (k == 0) {  // DC and Nyquist bins: only their real parts reach a real output (Re of the inverse DFT)
      a.y = 0.0;
      b.y = 0.0;
    }
    const double er = a.x + b.x, ei = a.y - b.y;  // 2E = A + conj(B)
    const double dr = a.x - b.x, di = a.y + b.y;  // 2D = A - conj(B)
    const double2 wk = ldg2(w + k);
    const double orr = fma(dr, wk.x, di * wk.y);     // 2O = 2D * conj(W^k),  conj(wk) = (wk.x, -wk.y)
    const double oi = fma(di, wk.x, -(dr * wk.y));
    z[k] = make_double2(er - oi, ei + orr);                      // Z[k]     = 2E + i*2O
    if (k != 0) z[N / 2 - k] = make_double2(er + oi, orr - ei);  // Z[N/2-k] = conj(2E) + i*conj(2O)
  }
  sync_lds<SNT>();
  fft_lds<N / 2, true, NT, SNT, MAXR>(z, tw_base + N / 2);
}

}  // namespace wh
