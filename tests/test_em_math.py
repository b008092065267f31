"""Accuracy of the kernel's exp / 1-exp / log (colate_amd/csrc/em_math.hpp) measured on the host: the
header is plain IEEE arithmetic (fma, rint, ldexp, frexp, divide), so the CPU build computes the very
same doubles as gfx950.  mpmath is the arbiter."""
import ctypes
import os
import subprocess

import mpmath as mp
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = ctypes.POINTER(ctypes.c_double)


@pytest.fixture(scope="module")
def m(tmp_path_factory):
    d = tmp_path_factory.mktemp("emmath")
    src = d / "h.cpp"
    src.write_text('#include "%s/colate_amd/csrc/em_math.hpp"\n#include <cmath>\nextern "C" {\n'
                   "void t_exp(int n,const double*x,double*y){for(int i=0;i<n;i++)y[i]=em::em_exp(x[i]);}\n"
                   "void t_om(int n,const double*x,double*y,double*z){for(int i=0;i<n;i++)y[i]=em::em_exp_om(x[i],&z[i]);}\n"
                   "void t_log(int n,const double*x,double*y){for(int i=0;i<n;i++)y[i]=em::em_log(x[i]);}\n"
                   "void t_exp_t(int n,const double*x,double*y){for(int i=0;i<n;i++)y[i]=em::em_exp_t(x[i],em::kExpTableHost);}\n"
                   "void t_om_t(int n,const double*x,double*y,double*z){for(int i=0;i<n;i++)y[i]=em::em_exp_om_t(x[i],&z[i],em::kExpTableHost);}\n"
                   "void g_exp(int n,const double*x,double*y){for(int i=0;i<n;i++)y[i]=std::exp(x[i]);}\n"
                   "long t_div(long n,unsigned long long seed){long bad=0;unsigned long long s=seed;\n"
                   " for(long i=0;i<n;i++){s=s*6364136223846793005ULL+1442695040888963407ULL;double u=(s>>11)*(1.0/9007199254740992.0);\n"
                   "  s=s*6364136223846793005ULL+1442695040888963407ULL;double v=(s>>11)*(1.0/9007199254740992.0);\n"
                   "  double lam=std::exp(std::log(5e-9)+u*(std::log(1e-1)-std::log(5e-9)));double inv=1.0/lam;\n"
                   "  double a=std::exp(std::log(0.05)+v*(std::log(1e7)-std::log(0.05)));double num=a+inv;\n"
                   "  if(em::em_div_known_rcp(num,inv,lam)!=num/inv)bad++;}return bad;}\n}\n" % ROOT)
    so = d / "libh.so"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so), str(src)])
    return ctypes.CDLL(str(so))


def _call(f, x):
    y = np.zeros_like(x)
    f(len(x), x.ctypes.data_as(dp), y.ctypes.data_as(dp))
    return y


def _ulp(y, exact):
    return np.array([float(abs(mp.mpf(a) - b) / mp.mpf(np.spacing(abs(a)) if a != 0 else 5e-324)) for a, b in zip(y, exact)])


def test_exp_is_nearly_correctly_rounded(m):
    mp.mp.prec = 120
    rng = np.random.default_rng(0)
    x = np.concatenate([-np.exp(rng.uniform(np.log(1e-12), np.log(700), 4000)), rng.uniform(-1, 1, 600), rng.uniform(0, 700, 200)])
    y = _call(m.t_exp, x)
    exact = [mp.exp(mp.mpf(v)) for v in x]
    assert _ulp(y, exact).max() < 0.6
    cr = np.array([float(v) for v in exact])
    assert np.mean(y == cr) > 0.99          # correctly rounded for > 99 % of arguments
    assert np.mean(y == _call(m.g_exp, x)) > 0.99  # i.e. the doubles the reference's libm exp() gives
    sp = np.array([0.0, -0.0, -np.inf, -745.2, -800.0, -1e5, -1e300, 709.7, -708.5, -740.0])
    assert np.array_equal(_call(m.t_exp, sp), _call(m.g_exp, sp))


def test_table_exp_the_kernel_uses(m):
    """em_exp_t / em_exp_om_t (32-entry table + degree-7 polynomial; what the EM kernel calls): at least as accurate as
    the series version -- <= 0.53 ulp, correctly rounded and equal to glibc's exp() for > 99.5 % of arguments; the
    1 - exp() companion within 1.7 ulp (its worst stretch is -0.014 < x < -0.011, where two terms of similar size cancel)."""
    mp.mp.prec = 120
    rng = np.random.default_rng(5)
    x = np.concatenate([-np.exp(rng.uniform(np.log(1e-12), np.log(700), 6000)), rng.uniform(-1, 0, 1500), -rng.uniform(0, 0.03, 1500)])
    y = _call(m.t_exp_t, x)
    exact = [mp.exp(mp.mpf(v)) for v in x]
    assert _ulp(y, exact).max() < 0.53
    cr = np.array([float(v) for v in exact])
    assert np.mean(y == cr) > 0.995 and np.mean(y == _call(m.g_exp, x)) > 0.995
    sp = np.array([0.0, -0.0, -np.inf, -745.2, -800.0, -1e5, -1e300, -708.5, -740.0, -709.8, -1100.0, -1e-320])
    assert np.array_equal(_call(m.t_exp_t, sp), _call(m.g_exp, sp))
    yo, zo = np.zeros_like(x), np.zeros_like(x)
    m.t_om_t(len(x), x.ctypes.data_as(dp), yo.ctypes.data_as(dp), zo.ctypes.data_as(dp))
    assert np.array_equal(yo, y)
    u = _ulp(zo, [-mp.expm1(mp.mpf(v)) for v in x])
    assert u.max() < 1.7 and np.mean(u > 1.0) < 0.01


def test_one_minus_exp(m):
    mp.mp.prec = 120
    rng = np.random.default_rng(1)
    x = -np.exp(rng.uniform(np.log(1e-14), np.log(700), 4000))
    y, z = np.zeros_like(x), np.zeros_like(x)
    m.t_om(len(x), x.ctypes.data_as(dp), y.ctypes.data_as(dp), z.ctypes.data_as(dp))
    assert np.array_equal(y, _call(m.t_exp, x))
    assert _ulp(z, [-mp.expm1(mp.mpf(v)) for v in x]).max() < 1.1


def test_log(m):
    mp.mp.prec = 120
    rng = np.random.default_rng(2)
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-300), np.log(1e300), 3000)), rng.uniform(0.5, 2, 1000),
                        1 + rng.uniform(-1e-6, 1e-6, 500), np.array([5e-324, 1e-310, 2.2250738585072014e-308])])
    assert _ulp(_call(m.t_log, x), [mp.log(mp.mpf(v)) for v in x]).max() < 1.1
    sp = _call(m.t_log, np.array([0.0, -1.0, np.inf, 1.0]))
    assert sp[0] == -np.inf and np.isnan(sp[1]) and sp[2] == np.inf and sp[3] == 0.0


def test_division_by_known_reciprocal_is_ieee(m):
    """(age + 1/lambda)/(1/lambda) through em_div_known_rcp equals the correctly rounded quotient."""
    m.t_div.restype = ctypes.c_long
    m.t_div.argtypes = [ctypes.c_long, ctypes.c_ulonglong]
    assert m.t_div(3_000_000, 12345) == 0
