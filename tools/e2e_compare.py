#!/usr/bin/env python3
"""End-to-end drop-in check at whole-genome scale (run on the GPU box):
synthetic 22-chromosome .mut + two .colate.in (tests/synth_files.py), then the SAME command line
through the reference CLI (oracle/_ref/Colate_ref, if it travelled) and through colate_amd's
`Colate`; reports wall times and compares the .coal files (iteration counts and rates).

    python tools/e2e_compare.py [snps_per_chr] [num_bootstraps]
"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth_files  # noqa: E402

snps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ref_bin = os.path.join(ROOT, "oracle", "_ref", "Colate_ref")
cli = os.path.join(ROOT, "colate_amd", "bin", "Colate")

with tempfile.TemporaryDirectory() as d:
    t0 = time.perf_counter()
    synth_files.write_inputs(d, chroms=tuple(str(c) for c in range(1, 23)), snps_per_chr=snps, seed=1, span=240_000_000)
    print(f"inputs: 22 chromosomes x {snps} SNPs in {time.perf_counter() - t0:.1f} s")
    args = ["--mode", "mut", "--mut", "P", "--target_tmp", "T.colate.in", "--reference_tmp", "R.colate.in", "--chr", "chr.txt",
            "--bins", "3,7,0.2", "--seed", "1", "--num_bootstraps", str(B)]
    res = {}
    for name, exe in (("colate_amd", cli), ("reference", ref_bin)):
        if not os.path.exists(exe):
            print(name, "binary missing, skipped")
            continue
        runs = []
        for _ in range(3 if name == "colate_amd" else 1):  # (ours three times: the first run also pages the binaries in)
            t0 = time.perf_counter()
            r = subprocess.run([exe] + args + ["-o", name], cwd=d, capture_output=True)
            runs.append(time.perf_counter() - t0)
            assert r.returncode == 0, r.stderr.decode()[-500:]
        dt = min(runs)
        err = r.stderr.decode()
        iters = [int(l.split("\r")[-1].rsplit(" ", 1)[1]) for l in err.split("\n") if l.split("\r")[-1].startswith("Bootstrap ")]
        lines = open(os.path.join(d, name + ".coal")).read().split("\n")
        rates = np.array([[float(x) for x in l.split()[2:]] for l in lines[2:] if l])
        nblocks = [l for l in err.split("\n") if l.startswith("Number of blocks")]
        res[name] = (dt, iters, rates, lines[:2])
        print(f"{name}: {dt:.2f} s wall" + (f" (runs: {' '.join('%.2f' % x for x in runs)})" if len(runs) > 1 else "")
              + f", {nblocks[0] if nblocks else ''}, iterations min/max {min(iters)}/{max(iters)}")
    if len(res) == 2:
        a, b = res["colate_amd"], res["reference"]
        print("header lines identical:", a[3] == b[3], "| iteration counts identical:", a[1] == b[1])
        rel = np.abs(a[2] - b[2]) / np.maximum(np.abs(b[2]), 1e-300)
        same = (a[2] == b[2])
        print(f"6-digit rates identical in {same.mean() * 100:.2f} % of the {same.size} entries; "
              f"max relative difference {rel.max():.2e} (epoch {np.unravel_index(rel.argmax(), rel.shape)[1]} of {a[2].shape[1]})")
        per_epoch = rel.max(axis=0)
        print("per-epoch max rel diff:", " ".join(f"{x:.0e}" for x in per_epoch))
        print(f"speed-up end to end: x{b[0] / a[0]:.1f}")
