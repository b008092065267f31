"""The algebra the HIP kernel implements (O(A+E) factorised E-step, tests/factorized_model.py is its
numpy statement) against the oracle's per-(bin, epoch) log-domain E-step.  CPU only."""
import numpy as np
import pytest

import factorized_model as fm
import oracle_lib as ol


def _rel(a, b):
    m = np.maximum(np.abs(a), np.abs(b))
    m[m == 0] = 1.0
    return np.abs(a - b) / m


@pytest.mark.parametrize("bins", ["3,7,0.2", "2,7.95,0.05"])
def test_estep_statistics(bins):
    grid = ol.age_grid()
    ep, _ = ol.epochs_from_bins(bins)
    E, A = ep.size, grid.size
    rng = np.random.default_rng(7)
    for t in range(10):
        rates = np.full(E, 1e-6 * 10 ** t) if t < 4 else np.exp(rng.uniform(np.log(1e-6), np.log(1e-3), E))
        csh = np.where(rng.uniform(size=A) < 0.8, rng.uniform(0, 50, A), 0.0)
        cns = np.where(rng.uniform(size=A) < 0.8, rng.uniform(0, 500, A), 0.0)
        csh[:40] = cns[:40] = 0
        N0, D0, ll0, fl = ol.estep(ep, rates, grid, csh, cns)
        assert fl == 0
        N1, D1, ll1 = fm.estep(ep, rates, grid, csh, cns)
        assert abs(ll1 - ll0) < 1e-12 * abs(ll0)
        assert _rel(N1, N0).max() < 1e-8
        dt = np.append(np.diff(ep), 0.0)
        assert (np.abs(D1 - D0) <= 1e-6 * np.abs(D0) + 1e-13 * dt * (csh.sum() + cns.sum())).all()


def test_em_trajectory():
    from colate_amd import workloads

    grid = ol.age_grid()
    ep, _ = ol.epochs_from_bins("3,7,0.2")
    csh, cns = workloads.bootstrap_tables(grid, 1, nb=9, scale=1.0)
    r0, it0, ll0, _ = ol.em_batch(grid, csh, cns, ep, max_iter=60, min_iter=1000)
    r = np.full(ep.size, 1.0 / 20000.0)
    for _ in range(60):
        N, D, ll = fm.estep(ep, r, grid, csh[0], cns[0])
        r = fm.mstep(N, D, r)
    assert _rel(r, r0[0]).max() < 1e-9 and abs(ll - ll0[0]) < 1e-12 * abs(ll0[0])
