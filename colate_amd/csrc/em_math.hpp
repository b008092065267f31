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

// exp(x), |error| <= ~0.52 ulp.  x may be any finite value or -inf; large
// negative arguments underflow gradually to 0 through ldexp.
EM_HD double em_exp(double x) {
  const double LOG2E = 0x1.71547652b82fep+0;
  const double LN2_HI = 0x1.62e42fefa3800p-1;   // 42 significant bits: k*LN2_HI is exact
  const double LN2_LO = 0x1.ef35793c76730p-45;  // ln2 - LN2_HI
  double xc = x > -1100.0 ? x : -1100.0;        // keeps k in range; exp(-1100) == 0 anyway
  double k = __builtin_rint(xc * LOG2E);
  double rh = fma_(-k, LN2_HI, xc);  // exact
  double rl = -k * LN2_LO;           // |rl| < 1e-10
  // exp(rh) = 1 + rh + rh^2 * (1/2 + rh * P(rh)),  P = 1/3! + rh/4! + ... + rh^11/14!
  double p = 0x1.93974a8c07c9dp-37;         // 1/14!
  p = fma_(p, rh, 0x1.6124613a86d09p-33);   // 1/13!
  p = fma_(p, rh, 0x1.1eed8eff8d898p-29);   // 1/12!
  p = fma_(p, rh, 0x1.ae64567f544e4p-26);   // 1/11!
  p = fma_(p, rh, 0x1.27e4fb7789f5cp-22);   // 1/10!
  p = fma_(p, rh, 0x1.71de3a556c734p-19);   // 1/9!
  p = fma_(p, rh, 0x1.a01a01a01a01ap-16);   // 1/8!
  p = fma_(p, rh, 0x1.a01a01a01a01ap-13);   // 1/7!
  p = fma_(p, rh, 0x1.6c16c16c16c17p-10);   // 1/6!
  p = fma_(p, rh, 0x1.1111111111111p-7);    // 1/5!
  p = fma_(p, rh, 0x1.5555555555555p-5);    // 1/4!
  p = fma_(p, rh, 0x1.5555555555555p-3);    // 1/3!
  double r2 = rh * rh;
  double t = r2 * fma_(p, rh, 0.5);  // rh^2/2 + rh^3 P(rh): |t| < 0.07, abs error ~1e-18
  double s1 = 1.0 + rh;              // Fast2Sum (|1| > |rh|): s1 + e1 == 1 + rh exactly
  double e1 = rh - (s1 - 1.0);
  t = t + fma_(rl, s1 + t, e1);      // + rl * exp(rh), + the bits lost in 1 + rh
  double y = s1 + t;
  y = __builtin_ldexp(y, (int)k);
  return (x != x) ? x : y;  // NaN in -> NaN out (the max() above would have dropped it)
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
