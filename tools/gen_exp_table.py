#!/usr/bin/env python3
"""Generates the constants of the table-driven exp() in colate_amd/csrc/em_math.hpp (mpmath, 200 bits):
2^(j/32) = hi + lo for j = 0..31, ln2/32 split in two 37-bit pieces (so that k * piece is exact for |k| < 2^16),
32/ln2.  Prints C hex-float literals."""
import mpmath as mp

mp.mp.prec = 200
N = 32


def to_double(x):
    return float(mp.mpf(x))


def hexf(x):
    return float(x).hex()


def chop(x, bits):
    """x rounded to `bits` significant bits (as an mpf)."""
    m, e = mp.frexp(x)
    return mp.ldexp(mp.nint(mp.ldexp(m, bits)), e - bits)


print("// 2^(j/32) = hi + lo, j = 0..31")
rows = []
for j in range(N):
    t = mp.power(2, mp.mpf(j) / N)
    hi = to_double(t)
    lo = to_double(t - mp.mpf(hi))
    rows.append(f"  {hexf(hi)}, {hexf(lo)},")
print("\n".join(rows))
l = mp.log(2) / N
hi = chop(l, 37)
mid = chop(l - hi, 37)
print("LN2N_HI ", hexf(to_double(hi)), " // 37 bits")
print("LN2N_MID", hexf(to_double(mid)), " // next 37 bits; remainder", mp.nstr(l - hi - mid, 5))
print("INVLN2N ", hexf(to_double(N / mp.log(2))))
for n in range(2, 8):
    print(f"C{n}", hexf(to_double(mp.mpf(1) / mp.factorial(n))))
