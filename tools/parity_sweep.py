#!/usr/bin/env python3
"""One-off confidence sweep (GPU box): many bootstrap replicates through the GPU path and through the oracle
(tests/oracle_lib: the C restatement, in worker processes); reports iteration-count mismatches and the largest
relative rate difference over the epochs whose oracle value is stable (oracle_lib.stable_mask).

    python tools/parity_sweep.py [replicates] [scale] [bins] [sample_age_in_years] [Ne2 of the dense tables]
(scale <= 0: sparse low-coverage-like tables; a sample age > 0 builds the epochs as for an ancient sample and
removes the counts of the age bins younger than it)
"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def work(args):
    import oracle_lib as ol
    grid, csh, cns, ep = args
    r0, it0, ll0, fl0 = ol.em_batch(grid, csh, cns, ep)
    mask = ol.stable_mask(grid, csh, cns, ep, r0)
    return r0, it0, ll0, fl0, mask


def main():
    import colate_amd
    from colate_amd import workloads
    import oracle_lib as ol

    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 11.0
    bins = sys.argv[3] if len(sys.argv) > 3 else "3,7,0.2"
    grid = ol.age_grid()
    age = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    ep, _ = ol.epochs_from_bins(bins, age, 28.0)
    if scale > 0:
        ne2 = float(sys.argv[5]) if len(sys.argv) > 5 else 12000.0
        csh, cns = workloads.bootstrap_tables(grid, B, nb=115 if scale > 2 else 9, scale=scale, ne2=ne2, seed=int(scale * 100) + B)
    else:  # scale <= 0: sparse, noisy tables (few mutations per bin): slow, irregular convergence
        csh, cns = workloads.sparse_tables(grid, B)
    if age > 0:
        csh[:, grid < age / 28.0] = 0.0
        cns[:, grid < age / 28.0] = 0.0
    t = time.time()
    r1, it1, ll1, fl1 = colate_amd.em_batch(grid, csh, cns, ep)
    print(f"GPU: {B} replicates in {time.time() - t:.3f} s (incl. transfers), flags nonzero: {(fl1 != 0).sum()}", flush=True)
    chunks = [(grid, csh[i:i + 8], cns[i:i + 8], ep) for i in range(0, B, 8)]
    t = time.time()
    with ProcessPoolExecutor(max_workers=14) as ex:
        res = list(ex.map(work, chunks))
    print(f"oracle: {time.time() - t:.1f} s on 14 processes", flush=True)
    r0 = np.concatenate([x[0] for x in res]); it0 = np.concatenate([x[1] for x in res])
    ll0 = np.concatenate([x[2] for x in res]); fl0 = np.concatenate([x[3] for x in res]); mask = np.concatenate([x[4] for x in res])
    ok = (fl0 & 3) == 0
    rel = np.abs(r1 - r0) / np.maximum(np.abs(r0), 1e-300)
    print(f"replicates the reference would abort on: {(~ok).sum()}")
    print(f"iteration counts: {(it1[ok] != it0[ok]).sum()} mismatches of {ok.sum()} (range {it0.min()}..{it0.max()})")
    print(f"log-likelihood max rel diff: {np.abs(ll1[ok] / ll0[ok] - 1).max():.2e}")
    print(f"stable epochs: {mask[ok].mean() * 100:.1f} %; max rel rate diff there: {rel[ok][mask[ok]].max():.2e}")
    bad = np.argwhere((rel > 1e-6) & mask & ok[:, None])
    print(f"entries beyond 1e-6 among stable epochs: {len(bad)}")
    for b, e in bad[:8]:
        lo, hi = max(0, e - 2), min(ep.size, e + 3)
        print(f"  replicate {b} epoch {e}/{ep.size} iters {it0[b]}: oracle {r0[b, lo:hi]} gpu {r1[b, lo:hi]}")
        nz = np.nonzero((csh[b] > 0) | (cns[b] > 0))[0]
        print(f"     data in bins {nz.min()}..{nz.max()} (ages {grid[nz.min()]:.1f}..{grid[nz.max()]:.1f}), epoch starts {ep[lo:hi]}")


if __name__ == "__main__":
    main()
