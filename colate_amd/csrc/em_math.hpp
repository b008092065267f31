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

// max(-x, lo): the negation as the instruction's source modifier (the compiler cannot fold a negation into the operand of an
// asm statement: it materialises -x first, two instructions)
EM_HD double max_c_neg(double x, double lo) {
#if defined(__HIP_DEVICE_COMPILE__)
  double d;
  asm("v_max_f64 %0, -%1, %2" : "=v"(d) : "v"(x), "s"(lo));
  return d;
#else
  return __builtin_fmax(-x, lo);
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

// ---- table-driven exp(): what the EM kernel uses ---------------------------------------------------------
// exp(x) = 2^e * 2^(j/32) * exp(r),  k = round(x * 32/ln2) = 32 e + j,  r = x - k ln2/32, |r| <= ln2/64 = 0.0108:
// the classic table method (the one glibc's own exp() is built on, with a 32-entry table here so that it fits the
// LDS budget of the throughput variant).  Against the series version above: a degree-7 instead of a degree-14
// polynomial and no double-double tail -- 20 instead of 36 instructions per call on gfx950, for one LDS read.
//   * ln2/32 is split into two 37-bit pieces, so k * piece is exact for |k| < 2^16 (|x| <= 1100) and r carries an
//     absolute error of ~1e-18 after two FMAs: no low part of r is needed;
//   * 2^(j/32) = th + tl to 106 bits; the result is th + (th * expm1(r) + tl), one rounding of a sum whose small
//     term is accurate to 1e-18 relative: ~0.51 ulp, like the series version (tests/test_em_math.py measures both).
// The table (32 x {hi, lo} = 512 bytes) lives in LDS on the device (the kernel copies kExpTable there once) and in
// a static array on the host; tools/gen_exp_table.py regenerates the literals (mpmath, 200 bits).
#define EM_EXP_TABLE_VALUES \
  0x1.0000000000000p+0, 0x0.0p+0, \
  0x1.059b0d3158574p+0, 0x1.d73e2a475b465p-55, \
  0x1.0b5586cf9890fp+0, 0x1.8a62e4adc610bp-54, \
  0x1.11301d0125b51p+0, -0x1.6c51039449b3ap-54, \
  0x1.172b83c7d517bp+0, -0x1.19041b9d78a76p-55, \
  0x1.1d4873168b9aap+0, 0x1.e016e00a2643cp-54, \
  0x1.2387a6e756238p+0, 0x1.9b07eb6c70573p-54, \
  0x1.29e9df51fdee1p+0, 0x1.612e8afad1255p-55, \
  0x1.306fe0a31b715p+0, 0x1.6f46ad23182e4p-55, \
  0x1.371a7373aa9cbp+0, -0x1.63aeabf42eae2p-54, \
  0x1.3dea64c123422p+0, 0x1.ada0911f09ebcp-55, \
  0x1.44e086061892dp+0, 0x1.89b7a04ef80d0p-59, \
  0x1.4bfdad5362a27p+0, 0x1.d4397afec42e2p-56, \
  0x1.5342b569d4f82p+0, -0x1.07abe1db13cadp-55, \
  0x1.5ab07dd485429p+0, 0x1.6324c054647adp-54, \
  0x1.6247eb03a5585p+0, -0x1.383c17e40b497p-54, \
  0x1.6a09e667f3bcdp+0, -0x1.bdd3413b26456p-54, \
  0x1.71f75e8ec5f74p+0, -0x1.16e4786887a99p-55, \
  0x1.7a11473eb0187p+0, -0x1.41577ee04992fp-55, \
  0x1.82589994cce13p+0, -0x1.d4c1dd41532d8p-54, \
  0x1.8ace5422aa0dbp+0, 0x1.6e9f156864b27p-54, \
  0x1.93737b0cdc5e5p+0, -0x1.75fc781b57ebcp-57, \
  0x1.9c49182a3f090p+0, 0x1.c7c46b071f2bep-56, \
  0x1.a5503b23e255dp+0, -0x1.d2f6edb8d41e1p-54, \
  0x1.ae89f995ad3adp+0, 0x1.7a1cd345dcc81p-54, \
  0x1.b7f76f2fb5e47p+0, -0x1.5584f7e54ac3bp-56, \
  0x1.c199bdd85529cp+0, 0x1.11065895048ddp-55, \
  0x1.cb720dcef9069p+0, 0x1.503cbd1e949dbp-56, \
  0x1.d5818dcfba487p+0, 0x1.2ed02d75b3707p-55, \
  0x1.dfc97337b9b5fp+0, -0x1.1a5cd4f184b5cp-54, \
  0x1.ea4afa2a490dap+0, -0x1.e9c23179c2893p-54, \
  0x1.f50765b6e4540p+0, 0x1.9d3e12dd8a18bp-54
constexpr int kExpTableDoubles = 64;
#if defined(__HIPCC__) || defined(__HIP__)
static __device__ const double kExpTableDevice[kExpTableDoubles] = {EM_EXP_TABLE_VALUES};
#endif
static const double kExpTableHost[kExpTableDoubles] = {EM_EXP_TABLE_VALUES};

struct ExpTab {
  double th, tl, p;  // exp(xc) = 2^e (th + (th p + tl)), p = expm1(r)
  int e;
};
EM_HD ExpTab em_exp_tab_parts(double xc, const double* tab) {
  const double INVLN2N = 0x1.71547652b82fep+5;   // 32 / ln2
  const double LN2N_HI = 0x1.62e42fefa0000p-6;   // ln2/32, first 37 bits
  const double LN2N_MID = 0x1.cf79abc9e0000p-45; // next 37 bits (what is left is 1e-25)
  const double SHIFT = 0x1.8p52;
  ExpTab o;
  const double kd = fma_(xc, INVLN2N, SHIFT);  // k = nearest integer to 32 x / ln2 (|x| <= 1100: |k| < 2^16)
  const int ki = em_lo32(kd);
  const double k = kd - SHIFT;
  const double r = fma_(-k, LN2N_MID, fma_(-k, LN2N_HI, xc));  // first FMA exact, second rounds at ~1e-18
  const int j = ki & 31;
  o.e = ki >> 5;
  o.th = tab[2 * j];
  o.tl = tab[2 * j + 1];
  // expm1(r) = r + r^2 (1/2 + r/3! + ... + r^5/7!), Estrin
  const double r2 = r * r;
  const double a = fma_cc(0x1.5555555555555p-3, r, 0x1.0000000000000p-1);   // 1/2! + r/3!
  const double b = fma_cc(0x1.1111111111111p-7, r, 0x1.5555555555555p-5);   // 1/4! + r/5!
  const double c = fma_cc(0x1.a01a01a01a01ap-13, r, 0x1.6c16c16c16c17p-10); // 1/6! + r/7!
  const double r4 = r2 * r2;
  const double q = fma_(c, r4, fma_(b, r2, a));
  o.p = fma_(r2, q, r);
  return o;
}
// exp(x) for x <= 709 (any value down to -inf; a NaN is not propagated, as in em_exp)
// (-DCOLATE_EXP_SERIES: experiments only -- the kernel then evaluates the series version through the same calls)
EM_HD double em_exp_t(double x, const double* tab) {
#ifdef COLATE_EXP_SERIES
  (void)tab;
  return em_exp(x);
#endif
  const double xc = max_c(x, -1100.0);
  const ExpTab o = em_exp_tab_parts(xc, tab);
  return __builtin_ldexp(o.th + fma_(o.th, o.p, o.tl), o.e);
}
// exp(x) and 1 - exp(x) for x <= 0, the latter without cancellation near 0: with yh = 2^e th (exact) and
// yl = 2^e (th p + tl), 1 - exp = (1 - yh) - yl, where 1 - yh is exact whenever exp(x) >= 1/2
// (NEG: the argument is -x -- same operations on the same values, the negation riding on the first instruction)
template <bool NEG = false>
EM_HD double em_exp_om_t(double x, double* one_minus, const double* tab) {
#ifdef COLATE_EXP_SERIES
  (void)tab;
  return em_exp_om(NEG ? -x : x, one_minus);
#endif
  const double xc = NEG ? max_c_neg(x, -1100.0) : max_c(x, -1100.0);
  const ExpTab o = em_exp_tab_parts(xc, tab);
  const double yh = __builtin_ldexp(o.th, o.e);
  const double yl = __builtin_ldexp(fma_(o.th, o.p, o.tl), o.e);
  *one_minus = (1.0 - yh) - yl;
  return yh + yl;
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

// 1 / x, the IEEE (correctly rounded) reciprocal: the very sequence the compiler expands `1.0 / x` to on gfx950 -- v_rcp_f64, two
// Newton steps, the quotient estimate (= y for a numerator of 1), one Markstein correction with the exact residual, v_div_fixup for
// x = 0, inf, NaN -- minus the two v_div_scale and the scaling of v_div_fmas, which only act on operands beyond 2^+-768 or so: four
// instructions fewer for the same bits on every rate the EM can hold (tests/test_gpu_em_math.py compares it with the device's own
// division over the rates' range and beyond).  Host: the division itself.
EM_HD double em_rcp_ieee(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(x);
  double e = fma_(-x, y, 1.0);
  y = fma_(y, e, y);
  e = fma_(-x, y, 1.0);
  y = fma_(y, e, y);
  const double r = fma_(-x, y, 1.0);
  const double q = fma_(r, y, y);
  return __builtin_amdgcn_div_fixup(q, x, 1.0);
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
