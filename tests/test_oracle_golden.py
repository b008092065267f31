"""The oracle (oracle/colate_oracle.c) pinned against the reference: the committed golden vectors
(generated from the reference build by tests/golden/make_golden.py) and, when oracle/_ref is
present, the reference itself.  CPU only."""
import numpy as np
import pytest

import golden_lib as gl
import oracle_lib as ol


def test_l1_estep_bit_exact():
    n = 0
    for c in gl.l1_cases():
        ll, num, den = ol.em_call(c["kind"], c["epochs"], c["rates"], c["age"])
        assert ll == c["logl"] or (np.isnan(ll) and np.isnan(c["logl"]))
        assert np.array_equal(num, c["num"], equal_nan=True)
        assert np.array_equal(den, c["denom"], equal_nan=True)
        n += 1
    assert n > 250


@pytest.mark.parametrize("name", gl.l2_names())
def test_l2_em_text_and_iterations(name):
    c = gl.l2_case(name)
    grid = ol.age_grid()
    ep, _ = ol.epochs_from_bins(c["bins"])
    rates, iters, ll, fl = ol.em_batch(grid, c["csh"], c["cns"], ep)
    assert (fl == 0).all()
    assert iters.tolist() == c["iterations"]
    assert gl.coal_text(ep, rates) == c["coal"]


@pytest.mark.skipif(ol.REF is None, reason="oracle/_ref/libref_em.so not built (needs /root/reference)")
def test_oracle_equals_reference_on_random_inputs():
    rng = np.random.default_rng(99)
    grid = ol.age_grid()
    for bins in ("3,7,0.2", "2,7.95,0.05", "3,6,0.5"):
        ep, _ = ol.epochs_from_bins(bins)
        for _ in range(6):
            rates = np.exp(rng.uniform(np.log(5e-9), np.log(1e-2), ep.size))
            rates[rng.integers(0, ep.size, 2)] = 0.0
            if rates[-1] <= 0:
                rates[-1] = 1e-5
            for a in rng.choice(grid, 25):
                for kind in (0, 1):
                    l0, n0, d0 = ol.ref_em_call(kind, ep, rates, a)
                    l1, n1, d1 = ol.em_call(kind, ep, rates, a)
                    assert l0 == l1 or (np.isnan(l0) and np.isnan(l1))
                    assert np.array_equal(n0, n1, equal_nan=True) and np.array_equal(d0, d1, equal_nan=True)


def test_logsumexp_special_values():
    # inf OR NaN operands are "absent" (coal_EM.cpp:8-21); logminusexp(a, b) = -inf when a < b (:52-54)
    inf, nan = np.inf, np.nan
    assert ol.O.oracle_logsumexp(-inf, -1.0) == -1.0
    assert ol.O.oracle_logsumexp(nan, -2.0) == -2.0
    assert ol.O.oracle_logsumexp(inf, -2.0) == -2.0
    assert ol.O.oracle_logsumexp(-inf, nan) == -inf
    assert ol.O.oracle_logminusexp(-3.0, -2.0) == -inf
    assert ol.O.oracle_logminusexp(-2.0, -2.0) == -inf
    assert ol.O.oracle_logminusexp(-1.0, -inf) == -1.0
    assert abs(ol.O.oracle_logsumexp(-1.0, -1.0) - (-1.0 + np.log(2.0))) < 1e-15


def test_oracle_reports_the_last_epoch_assert():
    """coal_EM.cpp:351: for a not-shared count inside the last epoch the reference asserts rate > 0 (it aborts); the
    oracle reports it with the NaN flag, like the aborts at coal.cpp:3711 -- what the kernel does (tests/test_gpu_*::test_flags)."""
    grid = ol.age_grid()
    ep, _ = ol.epochs_from_bins("3,7,0.2")
    rates = np.full(ep.size, 1.0 / 20000.0)
    rates[-1] = 0.0
    csh, cns = np.zeros(grid.size), np.zeros(grid.size)
    cns[grid >= ep[-1]] = 2.0  # not-shared counts beyond the start of the last epoch
    assert (grid >= ep[-1]).any()
    N, D, ll, fl = ol.estep(ep, rates, grid, csh, cns)
    assert fl & 1
    cns[:] = 0.0
    cns[100] = 2.0  # the same zero rate is harmless for a bin in an earlier epoch
    N, D, ll, fl = ol.estep(ep, rates, grid, csh, cns)
    assert fl == 0
