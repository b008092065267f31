"""ctypes access to oracle/liboracle.so (our CPU restatement) and, when present, to
oracle/_ref/libref_em.so (the reference itself).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = ctypes.POINTER(ctypes.c_double)
ip = ctypes.POINTER(ctypes.c_int)
c_int, c_double = ctypes.c_int, ctypes.c_double

O = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
O.oracle_logsumexp.restype = c_double
O.oracle_logsumexp.argtypes = [c_double, c_double]
O.oracle_logminusexp.restype = c_double
O.oracle_logminusexp.argtypes = [c_double, c_double]
O.oracle_get_AB.argtypes = [c_int, dp, dp, dp, dp]
for _f in (O.oracle_em_shared, O.oracle_em_notshared):
    _f.restype = c_double
    _f.argtypes = [c_int, dp, dp, dp, dp, c_double, dp, dp]
O.oracle_estep.restype = c_double
O.oracle_estep.argtypes = [c_int, c_int, dp, dp, dp, dp, dp, dp, dp, ip]
O.oracle_mstep.argtypes = [c_int, dp, dp, c_double, dp]
O.oracle_em_batch.argtypes = [c_int] * 3 + [dp] * 5 + [c_int, c_int, c_double, c_double, dp, ip, dp, ip]
O.oracle_age_grid.argtypes = [dp, c_int]
O.oracle_epochs_from_bins.argtypes = [ctypes.c_char_p, c_double, c_double, dp, c_int, ip]
O.oracle_bootstrap_counts.argtypes = [c_int, c_int, dp, c_double, dp, dp, dp, dp, dp, dp, dp]
O.oracle_mt_seed.argtypes = [ctypes.c_void_p, ctypes.c_uint]
O.oracle_mt_next.restype = ctypes.c_uint
O.oracle_mt_next.argtypes = [ctypes.c_void_p]
O.oracle_uniform_int.argtypes = [ctypes.c_void_p, c_int]
O.oracle_uniform_real01.restype = c_double
O.oracle_uniform_real01.argtypes = [ctypes.c_void_p]
O.oracle_block_weights.argtypes = [ctypes.c_void_p, c_int, c_int, dp]

_REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libref_em.so")
REF = None
if os.path.exists(_REF_PATH):
    REF = ctypes.CDLL(_REF_PATH)
    REF.ref_em_call.restype = c_double
    REF.ref_em_call.argtypes = [c_int, c_int, dp, dp, c_double, c_double, dp, dp]
    REF.ref_em_estep.restype = c_double
    REF.ref_em_estep.argtypes = [c_int, c_int, dp, dp, dp, dp, dp, dp, dp]
COLATE_REF_BIN = os.path.join(ROOT, "oracle", "_ref", "Colate_ref")


def P(a):
    return a.ctypes.data_as(dp)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def age_grid():
    g = np.zeros(256)
    n = O.oracle_age_grid(P(g), 256)
    return g[:n].copy()


def epochs_from_bins(bins, age=0.0, ypg=28.0):
    ep = np.zeros(512)
    en = c_int(0)
    n = O.oracle_epochs_from_bins(bins.encode(), age, ypg, P(ep), 512, ctypes.byref(en))
    assert n > 0
    return ep[:n].copy(), en.value


def get_AB(epochs, rates):
    ep, r = f64(epochs), f64(rates)
    A_ep, B_ep = np.zeros(ep.size), np.zeros(ep.size)
    O.oracle_get_AB(ep.size, P(ep), P(r), P(A_ep), P(B_ep))
    return A_ep, B_ep


def em_call(kind, epochs, rates, age):
    """oracle_em_shared (kind 0) / oracle_em_notshared (kind 1) -> (logl, num, denom)."""
    ep, r = f64(epochs), f64(rates)
    A_ep, B_ep = get_AB(ep, r)
    num, den = np.zeros(ep.size), np.zeros(ep.size)
    f = O.oracle_em_shared if kind == 0 else O.oracle_em_notshared
    ll = f(ep.size, P(ep), P(r), P(A_ep), P(B_ep), float(age), P(num), P(den))
    return ll, num, den


def ref_em_call(kind, epochs, rates, age):
    ep, r = f64(epochs), f64(rates)
    num, den = np.zeros(ep.size), np.zeros(ep.size)
    ll = REF.ref_em_call(kind, ep.size, P(ep), P(r), float(age), float(age), P(num), P(den))
    return ll, num, den


def estep(epochs, rates, grid, csh, cns):
    ep, r, g, s, n = f64(epochs), f64(rates), f64(grid), f64(csh), f64(cns)
    N, D = np.zeros(ep.size), np.zeros(ep.size)
    fl = c_int(0)
    ll = O.oracle_estep(ep.size, g.size, P(ep), P(r), P(g), P(s), P(n), P(N), P(D), ctypes.byref(fl))
    return N, D, ll, fl.value


def em_batch(grid, csh, cns, epochs, init=None, max_iter=100000, min_iter=1000, rel_tol=1e-7, floor=5e-9):
    g, s, n, ep = f64(grid), f64(np.atleast_2d(csh)), f64(np.atleast_2d(cns)), f64(epochs)
    B, A = s.shape
    E = ep.size
    init = f64(np.full(E, 1.0 / 20000.0) if init is None else init)
    rates = np.zeros((B, E))
    iters = np.zeros(B, dtype=np.int32)
    ll = np.zeros(B)
    fl = np.zeros(B, dtype=np.int32)
    O.oracle_em_batch(B, E, A, P(g), P(s), P(n), P(ep), P(init), max_iter, min_iter, rel_tol, floor, P(rates),
                      iters.ctypes.data_as(ip), P(ll), fl.ctypes.data_as(ip))
    return rates, iters, ll, fl


def stable_mask(grid, csh, cns, epochs, rates0, rtol=1e-8, **kw):
    """Epochs whose oracle result is reproducible under a 1-ulp perturbation of the inputs.

    In epochs far older than all data the reference's denominators are made of the rounding
    residue of its own `integ = 1 - num[0] - num[1] - ...` recurrence (coal_EM.cpp:270-274,
    445-449): multiplying every count by (1 + 2^-52) moves its answer there by per cent
    (DESIGN.md §6).  Such epochs carry no information in the reference itself and are excluded
    from the 1e-6 parity claim; everywhere else the claim is checked."""
    s, n = f64(np.atleast_2d(csh)), f64(np.atleast_2d(cns))
    r1, _, _, _ = em_batch(grid, s * (1 + 2.0 ** -52), n * (1 + 2.0 ** -52), epochs, **kw)
    r2, _, _, _ = em_batch(grid, s * (1 - 2.0 ** -53), n * (1 - 2.0 ** -53), epochs, **kw)
    den = np.maximum(np.abs(rates0), 1e-300)
    stable = (np.abs(r1 - rates0) / den < rtol) & (np.abs(r2 - rates0) / den < rtol)
    # ... and whose denominator, in the reference's own E-step at its final rates, is either at least 1e7 times the
    # rounding residue of `integ` (~4e-17 per unit count and unit epoch length, DESIGN.md §6: what the reference
    # adds there is known to +-50 % only) or nothing but that residue (ratio < 3: the rate is the floor).
    ep = f64(epochs)
    dt = np.append(np.diff(ep), 0.0)
    for b in range(s.shape[0]):
        _, D0, _, _ = estep(ep, rates0[b], grid, s[b], n[b])
        residue = dt * 4e-17 * (s[b].sum() + n[b].sum())
        with np.errstate(divide="ignore", invalid="ignore"):
            rho = np.where(residue > 0, D0 / residue, np.inf)
        stable[b] &= (rho > 1e7) | ((rho < 3) & (rates0[b] <= 5e-9))
    # An epoch older than an unresolved one inherits it: its survival probability is a product over the younger
    # epochs' rates, and the EM couples them iteration after iteration (seen with sparse tables, where a noise-
    # determined rate is followed by epochs that read "floor" in one run and 1e-6 in another implementation).
    # So only epochs with nothing unresolved before them count.
    return np.logical_and.accumulate(stable, axis=1)
