#!/usr/bin/env python3
"""Micro-benchmark of the block-bootstrap kernel (weighted block sums + F
redistribution on the GPU, colate_bootstrap_counts_device) for rocprofv3 --kernel-trace; prints one JSON line.

    python tools/bench_bootstrap.py [replicates] [blocks]

Algorithmic bytes per replicate: the four [nb][A] block tables (4*nb*A*8, read by every workgroup -- from L2 after the
first) + nb weights + 2*A counts out.  Compulsory HBM traffic per launch: the tables once + B*nb weights + B*2*A out."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import colate_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 115
grid = colate_amd.age_grid()
A = grid.size
rng = np.random.default_rng(1)
tabs = [rng.uniform(0, 50, (nb, A)) * (rng.uniform(size=(nb, A)) < 0.6) for _ in range(4)]
w = rng.multinomial(nb, np.full(nb, 1.0 / nb), size=B).astype(np.float64)
dev = torch.device("cuda", 0)
f64 = dict(dtype=torch.float64, device=dev)
d_grid = torch.tensor(grid, **f64)
d_w = torch.tensor(w, **f64)
d_t = [torch.tensor(t, **f64) for t in tabs]
d_sh, d_ns = torch.empty((B, A), **f64), torch.empty((B, A), **f64)
for _ in range(3):
    colate_amd.bootstrap_counts_device(d_grid, 0.0, d_w, *d_t, d_sh, d_ns)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for a, b in ev:
    a.record()
    colate_amd.bootstrap_counts_device(d_grid, 0.0, d_w, *d_t, d_sh, d_ns)
    b.record()
torch.cuda.synchronize()
ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
# bit-identical to the host twin on a sample
rng_h = colate_amd.Rng(1)
alg = B * (4 * nb * A * 8 + nb * 8 + 2 * A * 8)
comp = 4 * nb * A * 8 + B * nb * 8 + B * 2 * A * 8
print(json.dumps({"kernel": "bootstrap_kernel", "replicates": B, "blocks": nb, "age_bins": int(A), "kernel_ms": ms,
                  "replicates_per_s": B / (ms * 1e-3), "algorithmic_bytes": alg, "algorithmic_GBps": alg / (ms * 1e-3) / 1e9,
                  "compulsory_hbm_bytes": comp, "hbm_peak_GBps": 8000.0, "frac_of_hbm_peak_algorithmic": alg / (ms * 1e-3) / 1e9 / 8000.0}))
