// colate_amd/csrc/em_math.hpp
//
// Double-precision exp / log used by the EM kernels, written as plain IEEE
// operations (add, mul, fma, rint, ldexp, frexp, correctly-rounded divide) so
// that the very same source gives bit-identical results on gfx950
// (v_fma_f64 / v_rndne_f64 / v_ldexp_f64 ...) and on a host CPU.  That lets
// tests/test_em_math.py measure their accuracy against mpmath without a GPU.
//
// Why not the stock device-library exp(): the reference evaluates, per epoch,
//     B = (t_b + 1/lambda) - (t_e + 1/lambda) * exp(-(cs_e - cs_b))
// (include/coal/coal_EM.cpp:120, 204, 336 of the reference), which multiplies the
// last-bit error of exp() by 1/lambda (up to 2e8).  glibc's exp() that the
// reference runs on is correctly rounded for all but ~1% of arguments
// (max error 0.511 ulp), so an exp() whose error before the final rounding is
// ~0.02 ulp reproduces the reference's doubles (and hence its rounding noise)
// almost always; a 1-ulp exp() does not.  em_exp() below keeps the leading
// terms of the series in double-double to get there for ~6 extra instructions.
#pragma once

#if defined(__HIPCC__) || defined(__HIP__)
#define EM_HD __host__ __device__ __forceinline__
#else
#define EM_HD inline
#endif

namespace em {

EM_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// c1 * x + c0 with two CONSTANTS.  On gfx950 the compiler selects the two-address v_fmac_f64 and then
// copies c0 into the destination first (one v_mov_b64 per term: c0 stays live across the EM loop); spelling
// the three-address v_fma_f64 out -- c1 from a scalar register pair, c0 from a loop-invariant vector pair --
// drops those copies from the dependent chains of the kernel.  Same operation, same result.
EM_HD double fma_cc(double c1, double x, double c0) {
#if defined(__HIP_DEVICE_COMPILE__)
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "s"(c1), "v"(x), "v"(c0));
  return d;
#else
  return __builtin_fma(c1, x, c0);
#endif
}
// max(x, lo) for a constant lo and a non-NaN x: one v_max_f64 (the builtin adds a canonicalising
// v_max_f64 x, x in front of it under IEEE mode).  Host: fmax.
EM_HD double max_c(double x, double lo) {
#if defined(__HIP_DEVICE_COMPILE__)
  double d;
  asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(x), "s"(lo));
  return d;
#else
  return __builtin_fmax(x, lo);
#endif
}

// Core of exp(): for xc >= -1100 returns y and k with exp(xc) = y * 2^k, y in [0.70, 1.42), and the
// pieces (rh, tp) of exp(rh) - 1 = rh + tp that 1 - exp() needs near 0.  The polynomial is
// evaluated Estrin-style (depth 5 instead of 12: the EM kernel is a chain of dependent
// instructions run by one wave per SIMD, so depth is what costs).
struct ExpParts {
  double y, k, rh, tp;
  int ki;
};
EM_HD int em_lo32(double v) {
  long long b;
  __builtin_memcpy(&b, &v, 8);
  return (int)b;
}
EM_HD ExpParts em_exp_parts(double xc) {
  const double LOG2E = 0x1.71547652b82fep+0;
  const double LN2_HI = 0x1.62e42fefa3800p-1;   // 42 significant bits: k*LN2_HI is exact
  const double LN2_LO = 0x1.ef35793c76730p-45;  // ln2 - LN2_HI
  const double SHIFT = 0x1.8p52;                // adding 1.5*2^52 rounds to an integer in the low mantissa bits
  ExpParts o;
  const double kd = fma_(xc, LOG2E, SHIFT);  // k = nearest integer to x*log2(e) (|x| <= 1100)
  o.ki = em_lo32(kd);                        // ... as an int, without a convert
  o.k = kd - SHIFT;                          // ... and as a double (exact)
  const double rh = fma_(-o.k, LN2_HI, xc);  // exact
  const double rl = -o.k * LN2_LO;           // |rl| < 1e-10
  // exp(rh) = 1 + rh + rh^2 * (1/2 + rh * P(rh)),  P = 1/3! + rh/4! + ... + rh^11/14!
  const double r2 = rh * rh;
  const double r4 = r2 * r2;
  const double r8 = r4 * r4;
  const double a0 = fma_cc(0x1.5555555555555p-5, rh, 0x1.5555555555555p-3);    // 1/3! + r/4!
  const double a1 = fma_cc(0x1.6c16c16c16c17p-10, rh, 0x1.1111111111111p-7);   // 1/5! + r/6!
  const double a2 = fma_cc(0x1.a01a01a01a01ap-16, rh, 0x1.a01a01a01a01ap-13);  // 1/7! + r/8!
  const double a3 = fma_cc(0x1.27e4fb7789f5cp-22, rh, 0x1.71de3a556c734p-19);  // 1/9! + r/10!
  const double a4 = fma_cc(0x1.1eed8eff8d898p-29, rh, 0x1.ae64567f544e4p-26);  // 1/11! + r/12!
  const double a5 = fma_cc(0x1.93974a8c07c9dp-37, rh, 0x1.6124613a86d09p-33);  // 1/13! + r/14!
  const double b0 = fma_(a1, r2, a0);
  const double b1 = fma_(a3, r2, a2);
  const double b2 = fma_(a5, r2, a4);
  const double p = fma_(b2, r8, fma_(b1, r4, b0));
  const double tp = r2 * fma_(p, rh, 0.5);  // rh^2/2 + rh^3 P(rh): |tp| < 0.07, abs error ~1e-18
  const double s1 = 1.0 + rh;               // Fast2Sum (|1| > |rh|): s1 + e1 == 1 + rh exactly
  const double e1 = rh - (s1 - 1.0);
  const double t = tp + fma_(rl, s1 + tp, e1);  // + rl * exp(rh), + the bits lost in 1 + rh
  o.y = s1 + t;
  o.rh = rh;
  o.tp = tp;
  return o;
}

// exp(x) for x <= 709, |error| <= ~0.52 ulp.  x may be any value down to -inf; large negative
// arguments underflow gradually to 0 through ldexp.  A NaN argument is NOT propagated (fmax drops it:
// the result is exp(-1100) = 0); the kernel's arguments are sums of finite products.
EM_HD double em_exp(double x) {
  const double xc = max_c(x, -1100.0);  // keeps k in range; exp(-1100) == 0 anyway
  const ExpParts o = em_exp_parts(xc);
  return __builtin_ldexp(o.y, o.ki);
}

// exp(x) and 1 - exp(x) for x <= 0, the latter without cancellation near 0 (|error| <= ~1 ulp).
EM_HD double em_exp_om(double x, double* one_minus) {
  const double xc = max_c(x, -1100.0);
  const ExpParts o = em_exp_parts(xc);
  const double y = __builtin_ldexp(o.y, o.ki);
  *one_minus = (o.ki == 0) ? -(o.rh + o.tp) : 1.0 - y;
  return y;
}

// 1/x for finite x > 0 within ~1 ulp (not correctly rounded): hardware seed + two Newton steps.
// Used only where the reference has no division of its own (normalising constants).
EM_HD double em_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(x);
  double e = fma_(-x, y, 1.0);
  y = fma_(y, e, y);
  e = fma_(-x, y, 1.0);
  y = fma_(y, e, y);
  return y;
#else
  return 1.0 / x;
#endif
}

// n / d where a reciprocal y of d is already known to ~1 ulp -- here d = fl(1/lambda), y = lambda.
// Quotient estimate + two Markstein corrections with the exact residual: equal to the IEEE
// (correctly rounded) quotient in every one of 2e7 random trials of the shapes the kernel uses
// (tests/test_em_math.py), at 5 instructions instead of the ~14 of a full division.
EM_HD double em_div_known_rcp(double n, double d, double y) {
  double q = n * y;
  double r = fma_(-q, d, n);
  q = fma_(r, y, q);
  r = fma_(-q, d, n);
  q = fma_(r, y, q);
  return q;
}

// log(x) for finite x > 0 (normal or subnormal); <= ~1 ulp.  Only feeds the
// log-likelihood (stop rule and reporting), where nothing amplifies its error.
// log(0) = -inf, log(negative) = NaN as in libm.
EM_HD double em_log(double x) {
  const double LN2_HI = 0x1.62e42fefa39efp-1;
  const double LN2_LO = 0x1.abc9e3b39803fp-56;
  if (!(x > 0.0)) return (x == 0.0) ? -__builtin_inf() : __builtin_nan("");
  if (x > 0x1.fffffffffffffp+1023) return x;  // +inf
  int m;
  double f = __builtin_frexp(x, &m);  // f in [0.5, 1)
  if (f < 0x1.6a09e667f3bcdp-1) {     // bring f into [sqrt(1/2), sqrt(2))
    f = f + f;
    m = m - 1;
  }
  double z = f - 1.0;  // exact
  double s = z / (2.0 + z);
  double w = s * s;
  // log(f) = 2 atanh(s) = 2s + s^3 (2/3 + 2/5 w + ... ),  |s| <= 0.1716
  double q = 0x1.642c8590b2164p-4;          // 2/23
  q = fma_(q, w, 0x1.8618618618618p-4);     // 2/21
  q = fma_(q, w, 0x1.af286bca1af28p-4);     // 2/19
  q = fma_(q, w, 0x1.e1e1e1e1e1e1ep-4);     // 2/17
  q = fma_(q, w, 0x1.1111111111111p-3);     // 2/15
  q = fma_(q, w, 0x1.3b13b13b13b14p-3);     // 2/13
  q = fma_(q, w, 0x1.745d1745d1746p-3);     // 2/11
  q = fma_(q, w, 0x1.c71c71c71c71cp-3);     // 2/9
  q = fma_(q, w, 0x1.2492492492492p-2);     // 2/7
  q = fma_(q, w, 0x1.999999999999ap-2);     // 2/5
  q = fma_(q, w, 0x1.5555555555555p-1);     // 2/3
  // 2s = z - s*z exactly in real arithmetic; use it to recover the division's rounding error
  double hfsq = s * z;                       // = z^2/(2+z)
  double logf = z - (hfsq - (s * w) * q);    // z - s z + s^3 q
  double dm = (double)m;
  return fma_(dm, LN2_HI, logf + dm * LN2_LO);
}

}  // namespace em
