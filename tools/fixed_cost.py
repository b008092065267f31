#!/usr/bin/env python3
"""Where the EM kernel's time goes that is not the steady-state iterations (GPU box): kernel time by events on its stream for
max_iter = 1, 2, 3, 5, 9, 17, 33, 65, 129, 257, 1001 at B = 100 (min_iter beyond max_iter: no log-likelihood phase), for 23 and 122 epochs.
The tail model is refreshed in iterations 0, 1, 2, 4, 8, ... (from 128 on every 128th with more than 64 epochs): the steady-state
rate is the slope between max_iter 140 and 250, which contain no refresh; the step from 250 to 262 contains exactly one (at 256).
    gpurun -- python3 tools/fixed_cost.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import colate_amd  # noqa: E402
import oracle_lib as ol  # noqa: E402
from colate_amd import workloads  # noqa: E402

grid = ol.age_grid()
dev = torch.device("cuda")
f64 = dict(dtype=torch.float64, device=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for bins in ("3,7,0.2", "2,7.95,0.05"):
    ep, _ = ol.epochs_from_bins(bins)
    E = ep.size
    csh, cns = workloads.bootstrap_tables(grid, B, nb=115, scale=11.0, seed=1)
    g, s, n = torch.tensor(grid, **f64), torch.tensor(csh, **f64), torch.tensor(cns, **f64)
    eps = torch.tensor(np.tile(ep, (B, 1)), **f64)
    init = torch.full((B, E), 1.0 / 20000.0, **f64)
    out = torch.empty((B, E), **f64)
    it = torch.empty(B, dtype=torch.int32, device=dev)
    ll = torch.empty(B, **f64)
    fl = torch.empty(B, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream()
    res = []
    for mi in (1, 2, 3, 5, 9, 17, 33, 65, 129, 140, 250, 262, 1001):
        ts = []
        with torch.cuda.stream(st):
            for rep in range(12):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                colate_amd.em_batch_device(g, s, n, eps, init, out, it, ll, fl, max_iter=mi, min_iter=1000000, stream=st)
                e1.record(st)
                st.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
        res.append((mi, float(np.median(ts[3:]))))
    t140, t250 = dict(res)[140], dict(res)[250]
    slope = (t250 - t140) / 110.0
    print("%s (E = %d, B = %d): steady-state iteration %.4f us" % (bins, E, B, slope))
    prev = None
    for mi, t in res:
        extra = "" if prev is None else "   step: +%6.1f us for %4d iterations = %6.1f us beyond the steady-state rate" % (t - prev[1], mi - prev[0], t - prev[1] - slope * (mi - prev[0]))
        print("  max_iter %4d: %8.1f us%s" % (mi, t, extra))
        prev = (mi, t)
    print("  fixed cost (max_iter 1001 minus 1001 steady-state iterations): %.1f us" % (res[-1][1] - 1001 * slope))
    # the iterations from min_iter on (log-likelihood and stop test in every one): min_iter = 0 and a relative
    # tolerance of 0 -- ll_i / ll_{i-1} > 1 never holds for an ascending negative log-likelihood --, slope between 140 and 250
    res_ll = []
    for mi in (140, 250, 1001):
        ts = []
        with torch.cuda.stream(st):
            for rep in range(12):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                colate_amd.em_batch_device(g, s, n, eps, init, out, it, ll, fl, max_iter=mi, min_iter=0, rel_tol=0.0, stream=st)
                e1.record(st)
                st.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
        res_ll.append((mi, float(np.median(ts[3:]))))
    assert int(it.min()) == 1001, "the stop test fired"
    d = dict(res_ll)
    print("  log-likelihood phase (min_iter = 0, rel_tol = 0): %.4f us per iteration; max_iter 1001: %.1f us (without: %.1f)"
          % ((d[250] - d[140]) / 110.0, d[1001], dict(res)[1001]))
