"""Per-iteration slope and fixed cost (copies + launch + prologue + epilogue) of colate_em_batch at B = 100:
    gpurun -- python3 tools/iter_intercept.py"""
import time, numpy as np, sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import colate_amd
from colate_amd import workloads
import oracle_lib as ol
grid = ol.age_grid()
for bins in ("3,7,0.2", "2,7.95,0.05"):
    ep, _ = ol.epochs_from_bins(bins)
    csh, cns = workloads.bootstrap_tables(grid, 100, nb=115, scale=11.0, seed=1)
    res = {}
    for mi in (200, 500, 1001):
        ts = []
        for rep in range(12):
            t = time.perf_counter()
            r, it, ll, fl = colate_amd.em_batch(grid, csh, cns, ep, min_iter=1000000, max_iter=mi)
            ts.append(time.perf_counter() - t)
        res[mi] = (np.median(ts[2:]) * 1e3, it.max())
    (t1, i1), (t2, i2), (t3, i3) = res[200], res[500], res[1001]
    slope = (t3 - t1) / (i3 - i1)
    t = time.perf_counter(); r, it, ll, fl = colate_amd.em_batch(grid, csh, cns, ep, min_iter=100); t = time.perf_counter() - t
    ts = []
    for rep in range(8):
        t = time.perf_counter(); r, it, ll, fl = colate_amd.em_batch(grid, csh, cns, ep, min_iter=100); ts.append(time.perf_counter() - t)
    print(bins, "min_iter=100: %d iterations (log-likelihood and stop test from iteration 100 on), %.4f us per iteration" % (it.max(), 1e6 * np.median(ts) / it.max()))
    print(bins, res, "per iteration %.4f us, intercept (copies + launch + prologue + epilogue) %.1f us" % (slope * 1e3, (t3 - slope * i3) * 1e3))
