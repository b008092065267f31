#!/usr/bin/env python3
"""GPU rates (full precision and as printed) of the 64-replicate, 122-epoch table of tests/golden/ref_spread_e122.json
from epoch `first_epoch` on, for the comparison with the reference's real builds (tools/ref_self_reproducibility.py):
    python tools/dump_gpu_rates.py OUT.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import colate_amd  # noqa: E402
import oracle_lib as ol  # noqa: E402
from colate_amd import workloads  # noqa: E402

fix = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_spread_e122.json")))
t = fix["table"]
grid = ol.age_grid()
ep, _ = ol.epochs_from_bins(fix["bins"])
csh, cns = workloads.bootstrap_tables(grid, t["replicates"], nb=t["nb"], scale=t["scale"], ne2=t["ne2"], seed=t["seed"])
r, it, ll, fl = colate_amd.em_batch(grid, csh, cns, ep)
f0 = fix["first_epoch"]
json.dump({"first_epoch": f0, "iterations": it.tolist(), "unresolved": colate_amd.unresolved_epochs(fl).tolist(),
           "rates": [[float(x) for x in row[f0:]] for row in r], "text": [["%g" % x for x in row[f0:]] for row in r]},
          open(sys.argv[1], "w"))
print("iterations equal to the stock build:", it.tolist() == fix["iterations"]["base"])
