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
O.oracle_epochs_from_coal.argtypes = [ctypes.c_char_p, c_double, dp, dp, c_int]
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
    ep = np.zeros(2048)
    en = c_int(0)
    n = O.oracle_epochs_from_bins(bins.encode(), age, ypg, P(ep), 2048, ctypes.byref(en))
    assert n > 0
    return ep[:n].copy(), en.value


def epochs_from_coal(path, age=0.0):
    ep, r = np.zeros(512), np.zeros(512)
    n = O.oracle_epochs_from_coal(str(path).encode(), age, P(ep), P(r), 512)
    assert n > 0, n
    return ep[:n].copy(), r[:n].copy()


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


O.oracle_set_libm_noise.argtypes = [ctypes.c_ulonglong]
NOISE_SEEDS = (7919, 104729, 1299709)


def rerun_rates(grid, csh, cns, epochs, **kw):
    """The oracle's rates under (i) libm noise (three seeds: every exp/log/log1p result of its EM path moved to a
    neighbouring double at random -- the same source on another < 1-ulp libm) and (ii) all counts scaled by
    (1 + 2^-52) and by (1 - 2^-53).  Returns (noise_reruns, scaling_reruns), lists of [B][E] arrays."""
    s, n = f64(np.atleast_2d(csh)), f64(np.atleast_2d(cns))
    noise = []
    for seed in NOISE_SEEDS:
        O.oracle_set_libm_noise(seed)
        try:
            noise.append(em_batch(grid, s, n, epochs, **kw)[0])
        finally:
            O.oracle_set_libm_noise(0)
    scaled = [em_batch(grid, s * (1 + 2.0 ** -52), n * (1 + 2.0 ** -52), epochs, **kw)[0],
              em_batch(grid, s * (1 - 2.0 ** -53), n * (1 - 2.0 ** -53), epochs, **kw)[0]]
    return noise, scaled


def mask_from_reruns(rates0, reruns, rtol=1e-8):
    den = np.maximum(np.abs(rates0), 1e-300)
    stable = np.ones(rates0.shape, dtype=bool)
    for r in reruns:
        stable &= np.abs(r - rates0) / den < rtol
    return np.logical_and.accumulate(stable, axis=1)


def stable_mask(grid, csh, cns, epochs, rates0, rtol=1e-8, **kw):
    """Epochs whose ORACLE value is pinned by the reference's source, not by the last bit of its libm or inputs.

    An epoch is stable if every rerun of rerun_rates() reproduces its rate to `rtol` (1e-8: a hundred times
    tighter than the 1e-6 parity claim) AND every younger epoch is stable too (the EM couples an epoch to all
    younger ones through the survival probability).  In epochs far older than all data the reference's
    denominators are made of the rounding residue of its own `integ = 1 - num[0] - num[1] - ...` recurrence
    (coal_EM.cpp:270-274, 445-449) and its printed rate moves by per cent to orders of magnitude in these reruns
    (profiles/parity/): such epochs carry no information in the reference itself and are outside the parity
    claim; everywhere else the claim is checked.  No constant of the kernel enters this definition; the kernel's
    own verdict (COLATE_UNRESOLVED_EPOCHS in out_flags) is compared with it in the GPU tests."""
    noise, scaled = rerun_rates(grid, csh, cns, epochs, **kw)
    return mask_from_reruns(rates0, noise + scaled, rtol)


def noise_envelope(r_gpu, r0, noise_reruns, floor=1e-10):
    """GPU-vs-oracle relative difference in units of the oracle's OWN spread under 1-ulp libm noise (rerun_rates()[0]),
    per (replicate, epoch): (ratio, spread).  Outside the 1e-6 claim the bar is "inside the reference's noise envelope":
    a few spreads wherever the reference is reproducible at all (spread well below 1)."""
    r_gpu, r0 = np.atleast_2d(r_gpu), np.atleast_2d(r0)
    den = np.maximum(np.abs(r0), 1e-300)
    spread = np.max([np.abs(np.atleast_2d(n) - r0) for n in noise_reruns], axis=0) / den
    return (np.abs(r_gpu - r0) / den) / np.maximum(spread, floor), spread


def check_rates(r_gpu, flags, r0, mask, rtol=1e-6):
    """The two parity statements of the GPU tests, per replicate:
    (1) on every epoch the checker finds stable (stable_mask) the GPU rate equals the oracle's within rtol;
    (2) the kernel's own verdict is safe: every epoch it does NOT count as unresolved (out_flags >> 8 trailing
        epochs) is within rtol too, whatever the checker's mask says.
    Returns (kernel-unresolved counts [B], checker-unstable counts [B])."""
    r_gpu, r0 = np.atleast_2d(r_gpu), np.atleast_2d(r0)
    B, E = r0.shape
    rel = np.abs(r_gpu - r0) / np.maximum(np.abs(r0), 1e-300)
    unres = (np.asarray(flags).astype(np.int64) & 0xFFFFFFFF) >> 8
    assert rel[mask].max(initial=0.0) < rtol, ("stable epochs differ", float(rel[mask].max(initial=0.0)))
    for b in range(B):
        keep = E - int(unres[b])
        assert rel[b, :keep].max(initial=0.0) < rtol, ("an epoch the kernel calls resolved differs", b, int(unres[b]),
                                                       int(np.argmax(rel[b, :keep] >= rtol)), float(rel[b, :keep].max()))
    return unres, E - mask.sum(axis=1)


def observed_spread_e122():
    """What the REFERENCE's own real builds print for the 64-replicate whole-genome table at 122 epochs (tests/golden/
    ref_spread_e122.json, tools/ref_self_reproducibility.py: the stock flags and five alternative flag sets, three of them with
    FMA contraction): returns (table spec, first epoch kept, {build: rates[64][E - first]} as printed, iterations of the stock build).
    The observed counterpart of rerun_rates(): where real builds exist, the envelope is theirs."""
    import json

    fix = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_spread_e122.json")))
    rates = {b: np.array([[float(x) for x in row] for row in v]) for b, v in fix["rates_from_first_epoch"].items()}
    return fix, fix["first_epoch"], rates, fix["iterations"]["base"]
