// Double-precision log (and exp / sincospi) for the spectral kernels (round 6).  The device library's are built for the last
// ulp over the whole range — log() is 113 instructions (double-double sums), exp() 56 and sincospi() 82, a third of them
// v_mov / s_mov pairs that materialise polynomial coefficients on every call — and one frame of a minimum-phase chain takes
// 513 logs and 514 complex exponentials, a frame of CheapTrick 513 logs and as many exponentials.
//   wh::flog — 74 instructions, <= 2 ulp (tests/test_hip_math.py: 5e-16 relative against NumPy over 1e-300 .. 1e300, near 1,
//   denormals, the ends of the domain) — is what the kernels use: CheapTrick 1.12 -> 1.04 ms and response_kernel 3.43 -> 3.33 ms
//   at config 2, req_filter 25.1 -> 24.3 and CheapTrick 17.7 -> 16.5 ms in the north-star batch.
//   wh::fexp / wh::fsincospi are accurate to the same few ulp but no shorter than the library's once the compiler has
//   materialised their coefficients (66 / 75 instructions), and with the coefficients read as a scalar table
//   (WH_MATH_SCALAR_TAB=1) every evaluation waits for the table: response 3.43 -> 3.90 ms, req_filter 25.0 -> 34.5.
//   Measured, tested (wh_math_probe), not used by any kernel.
// No loops and no branches: the ends of the domains are selects (log 0 = -inf, log of a negative = NaN, exp of +-inf = inf / 0),
// NaN goes through.
#pragma once
#include <hip/hip_runtime.h>

namespace wh {

#ifndef WH_MATH_SCALAR_TAB
#define WH_MATH_SCALAR_TAB 0  // 1: the coefficients as a __constant__ table read through the scalar unit (measured: the s_load
                              // latency in front of every evaluation made all three kernels SLOWER than the device library:
                              // response 3.43 -> 3.90 ms, req_filter 25.0 -> 34.5); 0: literals the compiler materialises
#endif
#if WH_MATH_SCALAR_TAB
__device__ __constant__ double kMathTab[44] = {
    // [0..8] atanh series 2/(2k+1), k = 1..9
    2.0 / 3, 2.0 / 5, 2.0 / 7, 2.0 / 9, 2.0 / 11, 2.0 / 13, 2.0 / 15, 2.0 / 17, 2.0 / 19,
    // [9] ln2 high part (33 bits), [10] low part, [11] log2(e)
    0x1.62e42fee00000p-1, 0x1.a39ef35793c76p-33, 0x1.71547652b82fep+0,
    // [12..25] 1/n!, n = 13 .. 0
    1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0,
    1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0,
    // [26..33] sin(pi t) = t (c7 u^7 + ... + c0), u = t^2: (-1)^k pi^(2k+1) / (2k+1)!, k = 7 .. 0
    -0x1.6fadb9f155744p-16, 0x1.e8f434d018d63p-12, -0x1.e3074fde8871fp-8, 0x1.50783487ee782p-4, -0x1.32d2cce62bd86p-1,
    0x1.466bc6775aae2p+1, -0x1.4abbce625be53p+2, 0x1.921fb54442d18p+1,
    // [34..42] cos(pi t) = d8 u^8 + ... + d0: (-1)^k pi^(2k) / (2k)!, k = 8 .. 0
    0x1.20c62c2f2d7f5p-18, -0x1.b6e24f44b128fp-14, 0x1.f9d38a3763cc3p-10, -0x1.a6d1f2a204a8cp-6, 0x1.e1f506891babbp-3,
    -0x1.55d3c7e3cbffap+0, 0x1.03c1f081b5ac4p+2, -0x1.3bd3cc9be45dep+2, 1.0,
    0.0};

// the table through a pointer the compiler cannot see through: uniform address -> s_load, values in SGPRs
__device__ __forceinline__ const double* math_tab() {
  const double* p = kMathTab;
  asm volatile("" : "+s"(p));
  return p;
}
#else
struct MathTab {
  static constexpr double v[44] = {
    // [0..8] atanh series 2/(2k+1), k = 1..9
    2.0 / 3, 2.0 / 5, 2.0 / 7, 2.0 / 9, 2.0 / 11, 2.0 / 13, 2.0 / 15, 2.0 / 17, 2.0 / 19,
    // [9] ln2 high part (33 bits), [10] low part, [11] log2(e)
    0x1.62e42fee00000p-1, 0x1.a39ef35793c76p-33, 0x1.71547652b82fep+0,
    // [12..25] 1/n!, n = 13 .. 0
    1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0,
    1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0,
    // [26..33] sin(pi t) = t (c7 u^7 + ... + c0), u = t^2: (-1)^k pi^(2k+1) / (2k+1)!, k = 7 .. 0
    -0x1.6fadb9f155744p-16, 0x1.e8f434d018d63p-12, -0x1.e3074fde8871fp-8, 0x1.50783487ee782p-4, -0x1.32d2cce62bd86p-1,
    0x1.466bc6775aae2p+1, -0x1.4abbce625be53p+2, 0x1.921fb54442d18p+1,
    // [34..42] cos(pi t) = d8 u^8 + ... + d0: (-1)^k pi^(2k) / (2k)!, k = 8 .. 0
    0x1.20c62c2f2d7f5p-18, -0x1.b6e24f44b128fp-14, 0x1.f9d38a3763cc3p-10, -0x1.a6d1f2a204a8cp-6, 0x1.e1f506891babbp-3,
    -0x1.55d3c7e3cbffap+0, 0x1.03c1f081b5ac4p+2, -0x1.3bd3cc9be45dep+2, 1.0,
    0.0};
  __device__ __forceinline__ constexpr double operator[](int i) const { return v[i]; }
};
__device__ __forceinline__ constexpr MathTab math_tab() { return MathTab(); }
#endif

// a / b through the reciprocal: v_rcp_f64, two Newton steps, the quotient and one residual correction — 8 instructions and
// <= 1 ulp where the IEEE division the compiler emits is ~15 (two v_div_scale, the reciprocal, five fused steps, v_div_fmas,
// v_div_fixup).  For divides whose operands are ordinary normal numbers and whose result feeds values compared at
// tolerances (the interval frequencies and interpolation slopes of Harvest's raw candidates, eight per frame and channel:
// hv_rawdet_kernel 5.65 -> 5.50 ms at 256 x 10 s; both forms of the raw-candidate stage use it, so they still agree bit for bit;
// the quotients of the refinement's epilogue and the crossing positions of the band walkers).  b = 0, Inf or NaN give Inf / NaN like the hardware reciprocal does.
#ifndef WH_FAST_DIV64
#define WH_FAST_DIV64 1
#endif
__device__ __forceinline__ double fdiv(double a, double b) {
#if WH_FAST_DIV64
  double y = __builtin_amdgcn_rcp(b);
  y = fma(fma(-b, y, 1.0), y, y);
  y = fma(fma(-b, y, 1.0), y, y);
  const double q = a * y;
  return fma(fma(-b, q, a), y, q);
#else
  return a / b;
#endif
}

// log(x), x > 0: x = m 2^e with m in [sqrt(1/2), sqrt(2)), log m = 2 atanh((m-1)/(m+1)) as nine terms of the series in
// r^2 <= 0.0295 (truncation 2e-17), e ln2 added in two parts.  The quotient is a reciprocal with two Newton steps and a
// residual correction.
__device__ __forceinline__ double flog(double x) {
  const auto T = math_tab();
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1) (denormals included)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool low = m < 0.70710678118654752440;
  m = low ? m + m : m;
  e = low ? e - 1 : e;
  const double f = m - 1.0, d = m + 1.0;
  double y = __builtin_amdgcn_rcp(d);
  y = fma(fma(-d, y, 1.0), y, y);
  y = fma(fma(-d, y, 1.0), y, y);
  double r = f * y;
  r = fma(fma(-d, r, f), y, r);
  const double s = r * r;
  double p = T[8];
#pragma unroll
  for (int k = 7; k >= 0; --k) p = fma(p, s, T[k]);
  const double ed = (double)e;
  // e ln2_hi is exact (33-bit constant, |e| < 2^11); the small terms are summed first
  const double small = fma(ed, T[10], (r * s) * p);
  double res = fma(ed, T[9], (r + r) + small);
  // the ends of the domain as the library has them: log(0) = -inf, log(inf) = inf, log(x < 0) = NaN (a NaN goes through by itself)
  res = x == 0.0 ? -__builtin_inf() : res;
  res = x == __builtin_inf() ? x : res;
  res = x < 0.0 ? __builtin_nan("") : res;
  return res;
}

// exp(x): x = k ln2 + r, |r| <= ln2 / 2, Taylor to r^13 (truncation 4e-18), scaled by v_ldexp
__device__ __forceinline__ double fexp(double x) {
  const auto T = math_tab();
  const double k = rint(x * T[11]);
  double r = fma(-k, T[9], x);
  r = fma(-k, T[10], r);
  double p = T[12];
#pragma unroll
  for (int i = 13; i <= 25; ++i) p = fma(p, r, T[i]);
  // (k beyond the int range only for |x| > 1e9: the clamp keeps the conversion defined, ldexp saturates to 0 / inf)
  const double kc = fmin(fmax(k, -4000.0), 4000.0);
  double res = ldexp(p, (int)kc);
  res = x > 709.79 ? __builtin_inf() : res;   // (also +inf, where r = inf - inf)
  res = x < -745.2 ? 0.0 : res;               // (also -inf)
  return res;
}

// (sin, cos)(pi x): x reduced to t in [-1/4, 1/4] around the nearest multiple of 1/2, two polynomials in t^2
// (truncation 5e-17 / 2e-18), quadrant by selects
__device__ __forceinline__ double2 fsincospi(double x) {
  const auto T = math_tab();
  const double r = fma(-2.0, rint(0.5 * x), x);  // [-1, 1]  (exact)
  const double qd = rint(r + r);                  // -2 .. 2
  const double t = fma(-0.5, qd, r);              // [-1/4, 1/4]  (exact)
  const double u = t * t;
  double ps = T[26], pc = T[34];
#pragma unroll
  for (int i = 27; i <= 33; ++i) ps = fma(ps, u, T[i]);
#pragma unroll
  for (int i = 35; i <= 42; ++i) pc = fma(pc, u, T[i]);
  const double s = t * ps, c = pc;
  const int q = (int)qd & 3;  // 0: (s, c)   1: (c, -s)   2: (-s, -c)   3: (-c, s)
  const double a = (q & 1) ? c : s, b = (q & 1) ? s : c;
  const double sn = (q & 2) ? -a : a;
  const double cs = ((q + 1) & 2) ? -b : b;
  return make_double2(sn, cs);
}

}  // namespace wh
