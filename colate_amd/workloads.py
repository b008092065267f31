"""Synthetic count tables in the shape of BASELINE.json's configs.

Real SGDP / LBK / Loschbour files are not available offline, so the benchmark and the parity
tests use age-binned shared / not-shared mutation counts generated from a fixed seed:
for age bins 40..150 of the 185-bin grid, tot ~ U(50,500)*scale split into nb genome blocks,
p = 1 - exp(-age/Ne2) the shared fraction (0.8*p), scale = 1 (chr1-like, nb ~ 9) or 11
(whole-genome-like, nb = 115).  Replicates are multinomial block re-weightings, exactly what
the reference's bootstrap driver produces (coal.cpp:3350-3390)."""
import numpy as np


def block_tables(age_grid, nb=115, scale=11.0, ne2=12000.0, seed=12345, lo=40, hi=150):
    rng = np.random.default_rng(seed)
    A = age_grid.size
    sh = np.zeros((nb, A))
    ns = np.zeros((nb, A))
    for j in range(nb):
        tot = rng.uniform(50, 500, hi - lo + 1) * scale / nb
        p = 1.0 - np.exp(-age_grid[lo:hi + 1] / ne2)
        sh[j, lo:hi + 1] = 0.8 * p * tot
        ns[j, lo:hi + 1] = tot - sh[j, lo:hi + 1]
    return sh, ns


def bootstrap_tables(age_grid, B, nb=115, scale=11.0, ne2=12000.0, seed=12345):
    """Returns (cnt_shared[B][A], cnt_notshared[B][A]) for B bootstrap replicates."""
    sh_b, ns_b = block_tables(age_grid, nb, scale, ne2, seed)
    rng = np.random.default_rng(seed + 1)
    if B == 1:
        W = np.ones((1, nb))
    else:
        W = rng.multinomial(nb, np.full(nb, 1.0 / nb), size=B).astype(np.float64)
    return W @ sh_b, W @ ns_b


def sparse_tables(age_grid, B, seed=None):
    """Low-coverage-like tables: per replicate a Poisson number of mutations per age bin (mean 0.3 .. 30, bins
    35..159) split binomially into shared / not shared with p = 0.8 (1 - exp(-age / Ne2)), Ne2 ~ U(3000, 40000).
    Convergence is slow and irregular (1001 .. 70000 iterations), single epochs get very high rates, and the
    epochs behind them fall below the resolution of the reference's `integ`: the inputs of
    tools/parity_sweep.py and of the tests of that regime."""
    rng = np.random.default_rng(B if seed is None else seed)
    A = age_grid.size
    tot = rng.poisson(np.exp(rng.uniform(np.log(0.3), np.log(30), (B, 1))) * np.ones((1, A))).astype(float)
    tot[:, :35] = 0
    tot[:, 160:] = 0
    p = 1 - np.exp(-age_grid / rng.uniform(3000, 40000, (B, 1)))
    csh = rng.binomial(tot.astype(int), np.clip(0.8 * p, 0, 1)).astype(float)
    return csh, tot - csh
