"""The kernel's math header on the DEVICE against its host build, bit for bit.

tests/test_em_math.py measures em_math.hpp's accuracy on the host (against mpmath and glibc); this test closes the
gap "the same source gives the same doubles on gfx950": em_exp, em_exp_om and em_log must be bit-identical, the
device-only em_rcp (v_rcp_f64 seed + two Newton steps) within 1 ulp of the IEEE quotient, and the exact division by a
known reciprocal (em_div_known_rcp) equal to the device's own IEEE division in the shapes the kernel uses."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = ctypes.POINTER(ctypes.c_double)


def _host(tmp_path):
    src = tmp_path / "h.cpp"
    src.write_text('#include "%s/colate_amd/csrc/em_math.hpp"\nextern "C" {\n'
                   "void t_exp(int n,const double*x,double*y){for(int i=0;i<n;i++)y[i]=em::em_exp(x[i]);}\n"
                   "void t_om(int n,const double*x,double*y,double*z){for(int i=0;i<n;i++)y[i]=em::em_exp_om(x[i],&z[i]);}\n"
                   "void t_log(int n,const double*x,double*y){for(int i=0;i<n;i++)y[i]=em::em_log(x[i]);}\n"
                   "void t_exp_t(int n,const double*x,double*y){for(int i=0;i<n;i++)y[i]=em::em_exp_t(x[i],em::kExpTableHost);}\n"
                   "void t_om_t(int n,const double*x,double*y,double*z){for(int i=0;i<n;i++)y[i]=em::em_exp_om_t(x[i],&z[i],em::kExpTableHost);}\n}\n" % ROOT)
    so = tmp_path / "libh.so"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so), str(src)])
    return ctypes.CDLL(str(so))


def test_device_math_equals_host_build(tmp_path):
    rng = np.random.default_rng(11)
    x = np.concatenate([
        -np.exp(rng.uniform(np.log(1e-14), np.log(1100), 60000)),       # the kernel's exp arguments: -(rate * time)
        rng.uniform(-2.0 ** -10, 0, 8000), -rng.uniform(0, 2.0 ** -30, 2000),
        [-0.0, -1e-300, -745.13, -745.2, -800.0, -1100.0, -1e5, -708.4, -709.0, -1.0, -0.5, -np.log(2.0)],
        # for the reciprocals: many more magnitudes over the rates' range and beyond, and the significands next to 1 and 2
        -np.exp(rng.uniform(np.log(1e-12), np.log(1e4), 400000)),
        -np.ldexp(np.repeat([2.0 - 2.0 ** -52, 2.0 - 2.0 ** -51, 1.0 + 2.0 ** -52, 1.0 + 2.0 ** -51, 1.5, 1.0], 52), np.tile(np.arange(-40, 12), 6)),
    ])
    n = x.size
    aux = np.exp(rng.uniform(np.log(0.05), np.log(1e7), n))  # ages / epoch starts for the division shape (t + 1/lambda)/(1/lambda)
    # for the division the magnitudes play the rates: map |x| into [5e-9, 1e-1]
    (tmp_path / "in.bin").write_bytes(np.concatenate([x, aux]).tobytes())
    exe = os.path.join(ROOT, "colate_amd", "bin", "em_math_device")
    subprocess.check_call([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    out = np.frombuffer((tmp_path / "out.bin").read_bytes(), dtype=np.float64).reshape(11, n)
    h = _host(tmp_path)
    y = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    P = lambda a: a.ctypes.data_as(dp)  # noqa: E731
    h.t_exp(n, P(x), P(y[0]))
    h.t_om(n, P(x), P(y[1]), P(y[2]))
    a = np.abs(x)
    h.t_log(n, P(a), P(y[3]))
    assert np.array_equal(out[0], y[0]), "em_exp differs between gfx950 and the host build"
    assert np.array_equal(out[1], y[1]) and np.array_equal(out[2], y[2]), "em_exp_om differs"
    assert np.array_equal(out[3], y[3]), "em_log differs"
    yt, yo, zo = np.zeros(n), np.zeros(n), np.zeros(n)
    h.t_exp_t(n, P(x), P(yt))
    h.t_om_t(n, P(x), P(yo), P(zo))
    assert np.array_equal(out[7], yt), "em_exp_t (the table-driven exp the kernel uses) differs between gfx950 and the host build"
    assert np.array_equal(out[8], yo) and np.array_equal(out[9], zo), "em_exp_om_t differs"
    pos = a > 1e-290
    inv = 1.0 / a[pos]
    assert np.array_equal(out[6][pos], inv), "device IEEE division differs from the host's"
    # em_rcp_ieee: the division's own instruction sequence without the operand scaling -- the same doubles wherever no scaling is
    # due (every magnitude here: 1e-290 .. 1e5), and an infinity for 0
    assert np.array_equal(out[10][pos], inv), "em_rcp_ieee is not the IEEE reciprocal"
    assert np.all(np.isinf(out[10][a == 0.0]))  # (the harness hands -0.0 through: -inf, as the division gives)
    ulp = np.abs(out[4][pos] - inv) / np.spacing(inv)
    assert ulp.max() <= 1.0, ulp.max()  # em_rcp: hardware seed + two Newton steps, not correctly rounded
    lam = a[(a >= 5e-9) & (a <= 1e-1)]
    sel = (a >= 5e-9) & (a <= 1e-1)
    want = (aux[sel] + 1.0 / lam) / (1.0 / lam)
    assert sel.sum() > 10000 and np.array_equal(out[5][sel], want), "em_div_known_rcp is not the IEEE quotient"
