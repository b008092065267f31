#!/usr/bin/env python3
"""Bit-for-bit comparison of two builds of libcolate_amd.so on the same inputs (GPU box): every EM output (rates, iteration counts,
log-likelihoods, flags) of a set of workloads that reach every build of the kernel -- the latency builds for up to 64 and 65..128 epochs
with and without the register cap, the throughput variant, four epochs per lane, sparse tables, zero starting rates.

    gpurun -- 'python3 tools/compare_libs.py colate_amd/lib_r03/libcolate_amd.so colate_amd/lib/libcolate_amd.so'

Each library runs in a process of its own (COLATE_AMD_LIB, colate_amd/_lib.py)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import colate_amd
    import oracle_lib as ol
    from colate_amd import workloads

    grid = ol.age_grid()
    res = {}
    rng = np.random.default_rng(5)
    for name, bins, B, kw in (("e23_b100", "3,7,0.2", 100, {}), ("e23_b400", "3,7,0.2", 400, {}), ("e23_b3000", "3,7,0.2", 3000, dict(max_iter=300, min_iter=100)),
                              ("e122_b100", "2,7.95,0.05", 100, {}), ("e122_b600", "2,7.95,0.05", 600, dict(max_iter=300, min_iter=100)),
                              ("e202_b50", "2,7.95,0.03", 50, dict(max_iter=300, min_iter=100)), ("e43_b64", "3,7,0.1", 64, {}),
                              ("e122_b64_ll", "2,7.95,0.05", 64, dict(max_iter=400, min_iter=50)), ("e23_b64_ll", "3,7,0.2", 64, dict(max_iter=400, min_iter=50))):
        ep, _ = ol.epochs_from_bins(bins)
        csh, cns = workloads.bootstrap_tables(grid, B, nb=115, scale=1.0)
        r, it, ll, fl = colate_amd.em_batch(grid, csh, cns, ep, **kw)
        res[name + "_r"], res[name + "_it"], res[name + "_ll"], res[name + "_fl"] = r, it, ll, fl
    # sparse tables (few live bins, one bin group, long runs past min_iter) and zero starting rates
    ep, _ = ol.epochs_from_bins("3,7,0.2")
    csh = np.zeros((40, grid.size)); cns = np.zeros((40, grid.size))
    for b in range(40):
        idx = rng.choice(np.arange(60, 140), size=6 + b % 20, replace=False)
        csh[b, idx] = rng.uniform(0.1, 30, idx.size); cns[b, idx] = rng.uniform(0.1, 60, idx.size)
    r, it, ll, fl = colate_amd.em_batch(grid, csh, cns, ep, max_iter=20000)
    res["sparse_r"], res["sparse_it"], res["sparse_ll"], res["sparse_fl"] = r, it, ll, fl
    init = np.full(ep.size, 1 / 20000.0); init[:5] = 0.0; init[-1] = 0.0
    csh2, cns2 = workloads.bootstrap_tables(grid, 16, nb=9, scale=1.0)
    r, it, ll, fl = colate_amd.em_batch(grid, csh2, cns2, ep, init_rates=init, max_iter=50, min_iter=10)
    res["zero_r"], res["zero_it"], res["zero_ll"], res["zero_fl"] = r, it, ll, fl
    np.savez(out, **res)


if len(sys.argv) == 3 and sys.argv[1] == "--worker":
    worker(sys.argv[2])
    sys.exit(0)
libs = sys.argv[1:3]
outs = []
for i, lib in enumerate(libs):
    out = "/tmp/compare_libs_%d.npz" % i
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--worker", out], env=dict(os.environ, COLATE_AMD_LIB=os.path.abspath(lib)))
    outs.append(np.load(out))
bad = 0
for k in outs[0].files:
    a, b = outs[0][k], outs[1][k]
    same = a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))
    if not same:
        bad += 1
        d = np.abs(a.astype(float) - b.astype(float))
        print("DIFFERENT %-14s max abs diff %.3e, %d of %d entries differ" % (k, np.nanmax(d), int((a != b).sum()), a.size))
print("%d arrays compared between %s and %s: %s" % (len(outs[0].files), libs[0], libs[1], "all bit-identical" if bad == 0 else "%d differ" % bad))
sys.exit(1 if bad else 0)
