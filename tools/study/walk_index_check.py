#!/usr/bin/env python3
"""Differential check of the indexed walk of `Colate --pairs` (csrc/mut_pairs.cpp, build_walk_index) against the cursor walk
(COLATE_INDEXED_WALK=0: the code the reference-made fixtures pin): random inputs with absent records, allele mismatches, DAF = 0,
records written twice (equal positions, other counts) and .mut rows written twice; samples in both roles, a sample against itself.
Build container (no GPU needed):   python tools/study/walk_index_check.py [first_seed [n_seeds]]"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth_files  # noqa: E402
from test_host_driver import _drop_out_of_order_records  # noqa: E402

CLI = os.path.join(ROOT, "colate_amd", "bin", "Colate")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for seed in range(first, first + (int(sys.argv[2]) if len(sys.argv) > 2 else 16)):
    d = tempfile.mkdtemp(prefix="idx")
    synth_files.write_inputs(d, chroms=("1", "2", "3"), snps_per_chr=15000, seed=seed, span=100_000_000, extra_targets=1, extra_refs=1)
    dup = (0, 5, 23)[seed % 3]
    for f in ("T", "T1", "R", "R1"):
        _drop_out_of_order_records(os.path.join(d, f + ".colate.in"), dup_every=dup)
    if seed % 2:  # every 50th .mut row twice
        for c in ("1", "2", "3"):
            L = open(f"{d}/P_chr{c}.mut").read().split("\n")
            out = [L[0]]
            for i, l in enumerate(L[1:]):
                out.append(l)
                if l and i % 50 == 0:
                    out.append(l)
            open(f"{d}/P_chr{c}.mut", "w").write("\n".join(out))
    pairs = [(f"{t}.colate.in", f"{r}.colate.in", f"o_{t}_{r}") for t in ("T", "T1", "R") for r in ("R", "R1", "T")]
    open(d + "/pairs.txt", "w").write("".join(" ".join(p) + "\n" for p in pairs))
    args = [CLI, "--mode", "mut", "--mut", "P", "--chr", "chr.txt", "--bins", "3,7,0.2", "--seed", str(seed), "--num_bootstraps", "2",
            "--pairs", "pairs.txt", "--counts_only"]
    outs = {}
    for mode in ("0", "1"):
        r = subprocess.run(args, cwd=d, capture_output=True, env=dict(os.environ, COLATE_INDEXED_WALK=mode, COLATE_TIMING="1", COLATE_UNIFORM_WINDOW_MB="4"))
        assert r.returncode == 0, r.stderr.decode()[-500:]
        outs[mode] = {p[2]: open(f"{d}/{p[2]}.counts", "rb").read() for p in pairs}
        if mode == "1":
            assert "(4 of 4 files)" in r.stderr.decode(), r.stderr.decode()[-400:]
    diff = [k for k in outs["0"] if outs["0"][k] != outs["1"][k]]
    print(seed, "records twice every", dup, "| rows twice:", bool(seed % 2), "| differing pairs:", diff)
    bad += len(diff)
    shutil.rmtree(d)
print("total differing pairs:", bad)
sys.exit(1 if bad else 0)
