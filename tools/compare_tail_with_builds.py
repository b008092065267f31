#!/usr/bin/env python3
"""GPU rates in the tail of the 122-epoch table (tools/dump_gpu_rates.py) against the reference's real builds
(tests/golden/ref_spread_e122.json): per epoch the builds' own spread (stock build against the FMA builds), the GPU's distance to
the stock build, and how many replicates the GPU prints between (or on) the two builds' values.
    python tools/compare_tail_with_builds.py gpurun_out/gpu_tail_e122.json [OUT.json]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_spread_e122.json")))
g = json.load(open(sys.argv[1]))
f0 = f["first_epoch"]
base = np.array([[float(x) for x in r] for r in f["rates_from_first_epoch"]["base"]])
alts = {k: np.array([[float(x) for x in r] for r in v]) for k, v in f["rates_from_first_epoch"].items() if k != "base"}
fma = alts["alt_fma"]
gpu = np.array([[float(x) for x in r] for r in g["text"]])  # as printed, like the builds
rows = []
for e in range(base.shape[1]):
    den = np.abs(base[:, e])
    spread = np.max([np.abs(a[:, e] - base[:, e]) for a in alts.values()], axis=0) / den
    rel = np.abs(gpu[:, e] - base[:, e]) / den
    lo = np.minimum(base[:, e], fma[:, e]); hi = np.maximum(base[:, e], fma[:, e])
    inside = (gpu[:, e] >= lo) & (gpu[:, e] <= hi)
    # distance to the nearer build, in units of the distance between the builds (0 inside)
    gap = np.where(inside, 0.0, np.minimum(np.abs(gpu[:, e] - lo), np.abs(gpu[:, e] - hi)) / np.maximum(hi - lo, 1e-300))
    med_spread = float(np.median(spread))
    rows.append({"epoch": e + f0, "builds_spread_median": med_spread, "builds_spread_p90": float(np.percentile(spread, 90)),
                 "gpu_vs_stock_median": float(np.median(rel)), "gpu_vs_stock_p90": float(np.percentile(rel, 90)), "gpu_vs_stock_max": float(rel.max()),
                 "ratio_of_medians": float(np.median(rel) / med_spread) if med_spread > 0 else (0.0 if np.median(rel) == 0 else float("inf")),
                 "gpu_between_the_builds": int(inside.sum()), "gpu_tokens_equal_stock": int((gpu[:, e] == base[:, e]).sum()),
                 "gpu_tokens_equal_fma": int((gpu[:, e] == fma[:, e]).sum()),
                 "log10_range": {"stock": [float(np.log10(base[:, e].min())), float(np.log10(base[:, e].max()))],
                                 "fma": [float(np.log10(fma[:, e].min())), float(np.log10(fma[:, e].max()))],
                                 "gpu": [float(np.log10(gpu[:, e].min())), float(np.log10(gpu[:, e].max()))]}})
for r in rows:
    if r["epoch"] >= 103:
        print(f"epoch {r['epoch']}: builds' spread med {r['builds_spread_median']:.2e} p90 {r['builds_spread_p90']:.2e} | gpu vs stock med {r['gpu_vs_stock_median']:.2e} "
              f"p90 {r['gpu_vs_stock_p90']:.2e} | ratio of medians {r['ratio_of_medians']:.2f} | gpu between the builds {r['gpu_between_the_builds']}/64, = stock {r['gpu_tokens_equal_stock']}, = fma {r['gpu_tokens_equal_fma']} "
              f"| log10 rate: stock {r['log10_range']['stock'][0]:.2f}..{r['log10_range']['stock'][1]:.2f} fma {r['log10_range']['fma'][0]:.2f}..{r['log10_range']['fma'][1]:.2f} gpu {r['log10_range']['gpu'][0]:.2f}..{r['log10_range']['gpu'][1]:.2f}")
if len(sys.argv) > 2:
    json.dump({"what": __doc__.split("\n\n")[0], "gpu_iterations_equal_stock": g["iterations"] == f["iterations"]["base"], "per_epoch": rows},
              open(sys.argv[2], "w"), indent=1)
