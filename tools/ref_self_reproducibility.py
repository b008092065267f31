#!/usr/bin/env python3
"""How far does the REFERENCE reproduce ITSELF at 122 epochs?  ("next" #2; build container only.)

The unmodified reference sources are compiled several ways by oracle/Makefile (`make -C oracle ref alts`): the stock
flags (`_ref/Colate_ref`: g++ -O3, what every golden fixture comes from), g++ -O2, clang++ -O3, and three builds in which
the compiler may contract a*b+c into one FMA instruction (g++ -mfma / -march=native with -ffp-contract=fast, clang++ -mfma
with -ffp-contract=on = the default of Apple's arm64 clang, i.e. of the binaries the reference's authors ship).  Same
sources, same glibc libm: only the rounding of the compiled arithmetic differs.  Every build fits the same count tables
through the reference's own .colate_mat hook (coal.cpp:3471-3499):

  * the golden fixture l2_em_wg_e122 (2 replicates), and
  * the 64-replicate whole-genome table of the parity sweep (tools/parity_sweep.py ... 64 11 2,7.95,0.05),

and the printed .coal rates (6 significant digits: what a user of the reference sees) are compared build against build.
Output: profiles/parity/ref_self_reproducibility_e122.json (per-epoch token differences and observed spread) and the
fixture tests/golden/ref_spread_e122.json (the printed rates of every build from epoch 96 on, for the GPU-side envelope
test and tools/parity_sweep.py).

    python tools/ref_self_reproducibility.py
"""
import json
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_lib as gl  # noqa: E402
import oracle_lib as ol  # noqa: E402
from colate_amd import workloads  # noqa: E402

BINS = "2,7.95,0.05"
REFDIR = os.path.join(ROOT, "oracle", "_ref")
BUILDS = ["base"] + sorted(d for d in os.listdir(REFDIR) if d.startswith("alt_") and os.path.exists(os.path.join(REFDIR, d, "Colate_ref")))
FIRST_EPOCH_KEPT = 96


def binary(build):
    return os.path.join(REFDIR, "Colate_ref") if build == "base" else os.path.join(REFDIR, build, "Colate_ref")


def flags_of(build):
    if build == "base":
        return "g++ -O3 -std=c++14 -fPIC (oracle/Makefile REFFLAGS: the stock build)"
    return open(os.path.join(REFDIR, build, "FLAGS")).read().strip()


def run(build, grid, csh, cns):
    B = csh.shape[0]
    with tempfile.TemporaryDirectory() as d:
        gl.write_colate_mat(os.path.join(d, "OUT.colate_mat"), grid, csh, cns)
        r = subprocess.run([binary(build), "--mode", "mut", "--mut", "dummy", "--bins", BINS, "--num_bootstraps", str(B), "-o", "OUT"],
                           cwd=d, capture_output=True)
        assert r.returncode == 0, (build, r.stderr.decode()[-800:])
        iters = []
        for line in r.stderr.decode().split("\n"):
            m = re.match(r"Bootstrap (\d+): Total iterations (\d+)$", line.split("\r")[-1])
            if m:
                iters.append(int(m.group(2)))
        lines = open(os.path.join(d, "OUT.coal")).read().split("\n")
    tokens = [l.split()[2:] for l in lines[2:] if l.strip()]
    assert len(tokens) == B and len(iters) == B, (build, len(tokens), len(iters))
    return tokens, iters


def analyse(name, grid, csh, cns):
    with ThreadPoolExecutor(max_workers=min(6, len(BUILDS))) as ex:
        res = dict(zip(BUILDS, ex.map(lambda b: run(b, grid, csh, cns), BUILDS)))
    base_tok, base_it = res["base"]
    B, E = len(base_tok), len(base_tok[0])
    base = np.array([[float(x) for x in row] for row in base_tok])
    out = {"tables": name, "replicates": B, "epochs": E, "builds": {}}
    vals = {}
    for b in BUILDS:
        tok, it = res[b]
        v = np.array([[float(x) for x in row] for row in tok])
        vals[b] = v
        differ = np.array([[tok[i][e] != base_tok[i][e] for e in range(E)] for i in range(B)])
        out["builds"][b] = {"flags": flags_of(b), "iterations_equal_to_base": it == base_it,
                            "iteration_differences": [int(x - y) for x, y in zip(it, base_it) if x != y],
                            "tokens_differing_from_base": int(differ.sum()),
                            "first_epoch_with_a_differing_token": int(np.argmax(differ.any(axis=0))) if differ.any() else None,
                            "replicates_differing_per_epoch": {str(e): int(differ[:, e].sum()) for e in range(E) if differ[:, e].any()}}
    stack = np.stack([vals[b] for b in BUILDS])  # [build][replicate][epoch]
    den = np.maximum(np.abs(base), 1e-300)
    spread = np.max(np.abs(stack - base[None]), axis=0) / den  # observed: largest deviation of any build from the stock build
    per_epoch = []
    for e in range(E):
        if spread[:, e].max() > 0:
            n_tok = int(np.sum(np.any(stack[:, :, e] != base[None, :, e], axis=0)))
            per_epoch.append({"epoch": e, "replicates_with_differing_tokens": n_tok, "spread_median": float(np.median(spread[:, e])),
                              "spread_p90": float(np.percentile(spread[:, e], 90)), "spread_max": float(spread[:, e].max()),
                              "base_rate_median": float(np.median(base[:, e]))})
    out["per_epoch"] = per_epoch
    out["first_epoch_any_build_differs"] = per_epoch[0]["epoch"] if per_epoch else None
    return out, res, spread


def main():
    assert len(BUILDS) > 1, "no alternative builds: make -C oracle ref alts"
    grid = ol.age_grid()
    ep, _ = ol.epochs_from_bins(BINS)
    c = gl.l2_case("wg_e122")
    rec_l2, _, _ = analyse("tests/golden/l2_em_wg_e122.json (2 replicates)", grid, c["csh"], c["cns"])
    csh, cns = workloads.bootstrap_tables(grid, 64, nb=115, scale=11.0, ne2=12000.0, seed=1164)
    rec_sw, res, spread = analyse("workloads.bootstrap_tables(grid, 64, nb=115, scale=11, ne2=12000, seed=1164): the table of "
                                  "profiles/parity/wg_e122_modern.json", grid, csh, cns)
    rec = {"what": __doc__.split("\n\n")[0], "bins": BINS, "epochs": int(ep.size), "builds": {b: flags_of(b) for b in BUILDS},
           "l2_em_wg_e122": rec_l2, "sweep_64": rec_sw}
    os.makedirs(os.path.join(ROOT, "profiles", "parity"), exist_ok=True)
    json.dump(rec, open(os.path.join(ROOT, "profiles", "parity", "ref_self_reproducibility_e122.json"), "w"), indent=1)
    # the fixture: what every build printed from epoch FIRST_EPOCH_KEPT on (the epochs before are token-identical in all builds)
    for b in BUILDS:
        for i, row in enumerate(res[b][0]):
            assert row[:FIRST_EPOCH_KEPT] == res["base"][0][i][:FIRST_EPOCH_KEPT], (b, i)
    fix = {"generator": "tools/ref_self_reproducibility.py (oracle/_ref/Colate_ref and oracle/_ref/alt_*/Colate_ref, .colate_mat hook)",
           "bins": BINS, "table": {"replicates": 64, "nb": 115, "scale": 11.0, "ne2": 12000.0, "seed": 1164},
           "first_epoch": FIRST_EPOCH_KEPT, "builds": {b: flags_of(b) for b in BUILDS},
           "iterations": {b: res[b][1] for b in BUILDS},
           "rates_from_first_epoch": {b: [row[FIRST_EPOCH_KEPT:] for row in res[b][0]] for b in BUILDS}}
    json.dump(fix, open(os.path.join(ROOT, "tests", "golden", "ref_spread_e122.json"), "w"))
    for r in (rec_l2, rec_sw):
        print(r["tables"])
        for b, v in r["builds"].items():
            print(f"  {b:14s} iterations equal: {v['iterations_equal_to_base']}; tokens differing from the stock build: "
                  f"{v['tokens_differing_from_base']} (first epoch {v['first_epoch_with_a_differing_token']})")
        for pe in r["per_epoch"]:
            print(f"  epoch {pe['epoch']:3d}: replicates with differing tokens {pe['replicates_with_differing_tokens']:3d}, "
                  f"spread median {pe['spread_median']:.2e} p90 {pe['spread_p90']:.2e} max {pe['spread_max']:.2e}, stock rate (median) {pe['base_rate_median']:.3g}")


if __name__ == "__main__":
    main()
