#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the REFERENCE ITSELF.

Runs only where /root/reference exists (this container): oracle/Makefile compiles the
reference's own sources into oracle/_ref/ (libref_em.so = class coal_EM behind our extern "C"
shim; Colate_ref = its `Colate` CLI) and this script records inputs + outputs:

  l1_estep.json     coal_EM::EM_shared / EM_notshared (coal_EM.cpp:153-468) on grids of
                    (epochs, rates, age): logl, num[E], denom[E] as hex floats (bit-exact)
  l2_em_*.json      count tables -> `Colate_ref --mode mut` through its .colate_mat hook
                    (coal.cpp:3169-3170, 3471-3499): .coal text and "Total iterations" per replicate
  l3_pairs/         four samples over one set of .mut files, the reference CLI run once per (target, reference) pair
  l3_*/             synthetic .mut(.gz) + .colate.in + chr.txt inputs (tests/synth_files.py) and the
                    .coal the reference CLI writes for them with --seed (full path: readers, age
                    sampling, block bootstrap, F redistribution, epochs, EM, writer)

The fixtures are data (inputs and expected outputs); no reference source is copied.
    python tests/golden/make_golden.py
"""
import gzip
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import oracle_lib as ol  # noqa: E402
import synth_files  # noqa: E402
from colate_amd import workloads  # noqa: E402

REF_BIN = ol.COLATE_REF_BIN


def hexes(a):
    return [float(x).hex() for x in np.atleast_1d(a)]


def run_ref(args, cwd):
    r = subprocess.run([REF_BIN] + args, cwd=cwd, capture_output=True)  # bytes: text mode would turn \r into \n
    stderr = r.stderr.decode()
    assert r.returncode == 0, stderr[-2000:]
    # progress messages end in "\r", the final one per replicate (coal.cpp:3823) in "\n"
    iters = []
    for line in stderr.split("\n"):
        m = re.match(r"Bootstrap (\d+): Total iterations (\d+)$", line.split("\r")[-1])
        if m:
            iters.append(int(m.group(2)))
    return stderr.replace("\r", "\n"), iters


def make_l1():
    assert ol.REF is not None, "oracle/_ref/libref_em.so missing: make -C oracle ref"
    rng = np.random.default_rng(2024)
    cases = []
    grid = ol.age_grid()
    for bins in ("3,7,0.2", "2,7.95,0.05"):
        ep, _ = ol.epochs_from_bins(bins)
        E = ep.size
        rate_sets = [np.full(E, 1.0 / 20000.0), np.full(E, 1e-7), np.exp(rng.uniform(np.log(5e-9), np.log(1e-3), E))]
        r = np.exp(rng.uniform(np.log(1e-6), np.log(1e-3), E))
        r[0] = 0.0
        r[3] = 5e-9
        rate_sets.append(r)
        ages = [0.0, grid[1], grid[30], grid[41], grid[64], grid[65], grid[90], grid[120], grid[150], grid[170],
                grid[184], float(ep[5]), float(ep[E - 1])]
        for rates in rate_sets:
            for age in ages:
                for kind in (0, 1):
                    ll, num, den = ol.ref_em_call(kind, ep, rates, age)
                    cases.append({"bins": bins, "kind": kind, "age": float(age).hex(), "rates": hexes(rates),
                                  "logl": float(ll).hex(), "num": hexes(num), "denom": hexes(den)})
    # the reference's own unit-test grid (include/test/test_aDNA.cpp:74-116): E = 21, constant rates
    E = 21
    ep = np.zeros(E)
    ep[1] = 1e3 / 28.0
    log10 = float(np.float32(np.log(10)))
    for e in range(2, E - 1):
        ep[e] = np.exp(log10 * (3.0 + 4.0 * (e - 1.0) / (E - 3.0))) / 28.0
    ep[E - 1] = 1e8 / 28.0
    for f in (1, 4, 7):
        rates = np.full(E, 1e-7 * np.exp(np.log(10) * (f - 1)))
        for b in range(0, 92, 7):
            age = np.exp(b / 5.0) / 10.0
            for kind in (0, 1):
                ll, num, den = ol.ref_em_call(kind, ep, rates, age)
                cases.append({"epochs": hexes(ep), "kind": kind, "age": float(age).hex(), "rates": hexes(rates),
                              "logl": float(ll).hex(), "num": hexes(num), "denom": hexes(den)})
    json.dump({"generator": "tests/golden/make_golden.py (oracle/_ref/libref_em.so)", "cases": cases},
              open(os.path.join(HERE, "l1_estep.json"), "w"))
    print("l1_estep.json:", len(cases), "cases")


def make_l2():
    grid = ol.age_grid()
    for name, bins, B, nb, scale, ne2 in (("wg_e23", "3,7,0.2", 3, 115, 11.0, 12000.0),
                                          ("chr1_e23", "3,7,0.2", 1, 9, 1.0, 12000.0),
                                          ("wg_e122", "2,7.95,0.05", 2, 115, 11.0, 12000.0),
                                          ("smallne_e23", "3,7,0.2", 2, 115, 11.0, 3000.0),
                                          ("largene_e23", "3,7,0.2", 2, 115, 11.0, 2000000.0)):
        csh, cns = workloads.bootstrap_tables(grid, B, nb=nb, scale=scale, ne2=ne2, seed=4242)
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "OUT.colate_mat"), "w") as f:
                f.write(" ".join("%.17g" % x for x in grid) + "\n")
                for b in range(B):
                    f.write(" ".join("%.17g" % x for x in csh[b]) + "\n")
                    f.write(" ".join("%.17g" % x for x in cns[b]) + "\n")
            err, iters = run_ref(["--mode", "mut", "--mut", "dummy", "--bins", bins, "--num_bootstraps", str(B), "-o", "OUT"], d)
            coal = open(os.path.join(d, "OUT.coal")).read()
        assert len(iters) == B, err[-500:]
        json.dump({"generator": "tests/golden/make_golden.py (oracle/_ref/Colate_ref, .colate_mat hook)", "bins": bins,
                   "cnt_shared": [hexes(r) for r in csh], "cnt_notshared": [hexes(r) for r in cns],
                   "coal": coal, "iterations": iters}, open(os.path.join(HERE, f"l2_em_{name}.json"), "w"))
        print(f"l2_em_{name}.json: B={B} iterations={iters}")


def make_l3():
    cases = {
        "l3_modern": dict(gen=dict(chroms=("1", "2"), snps_per_chr=1500, seed=7, gz=True),
                          args=["--bins", "3,7,0.2", "--seed", "1", "--num_bootstraps", "3", "--chr", "chr.txt"]),
        "l3_ancient": dict(gen=dict(chroms=("1", "2", "3"), snps_per_chr=800, seed=11, gz=True),
                           args=["--bins", "3,7,0.2", "--seed", "5", "--num_bootstraps", "2", "--chr", "chr.txt",
                                 "--target_age", "7000", "--reference_age", "0"]),
        "l3_nochr": dict(gen=dict(chroms=("1",), snps_per_chr=1200, seed=3, gz=False, with_chr_file=False),
                         args=["--bins", "3,6,0.5", "--seed", "9", "--num_bootstraps", "1"]),
    }
    # --coal warm start (coal.cpp:3508-3549, 3638-3646): epochs and starting rates from a Relate-style .coal file with
    # strictly increasing epochs (the reference asserts that, :3544-3546 -- a .coal written by `--mode mut --bins` itself
    # starts "0 0 ..." and is refused); modern, and ancient (the sample age replaces the epochs younger than it)
    rng = np.random.default_rng(99)
    prev_epochs = np.concatenate([[0.0], 10 ** np.arange(3.0, 7.01, 0.25) / 28.0, [1e8 / 28.0]])
    prev_rates = np.exp(rng.uniform(np.log(2e-6), np.log(4e-4), (2, prev_epochs.size)))
    prev_coal = "group\n" + "".join("%g " % x for x in prev_epochs) + "\n" + "".join(
        "0 %d " % i + "".join("%g " % x for x in r) + "\n" for i, r in enumerate(prev_rates))
    cases["l3_coal_modern"] = dict(gen=dict(chroms=("1", "2"), snps_per_chr=1000, seed=21, gz=True), extra={"prev.coal": prev_coal},
                                   args=["--coal", "prev.coal", "--seed", "3", "--num_bootstraps", "2", "--chr", "chr.txt"])
    cases["l3_coal_ancient"] = dict(inputs_from="l3_coal_modern",
                                    args=["--coal", "prev.coal", "--seed", "4", "--num_bootstraps", "2", "--chr", "chr.txt",
                                          "--target_age", "7000"])
    # --target_mask / --reference_mask (coal.cpp:2169-2174, data.cpp:213-235): per-chromosome fasta files, a site is used
    # only where both masks say 'P' (case-insensitive: the reader upper-cases); the reference mask of chromosome 2 is
    # shorter than the chromosome (sites beyond its end are not masked)
    def mask_text(n, seed, lower=False):
        r = np.random.default_rng(seed)
        seq = []
        while sum(len(x) for x in seq) < n:
            seq.append(("p" if lower and r.uniform() < 0.3 else "P") * int(r.integers(20_000, 120_000)))
            seq.append(("N" if r.uniform() < 0.7 else "n") * int(r.integers(5_000, 40_000)))
        seq = "".join(seq)[:n]
        return ">mask\n" + "\n".join(seq[i:i + 100] for i in range(0, n, 100)) + "\n"
    span = 3_000_000
    cases["l3_masks"] = dict(gen=dict(chroms=("1", "2"), snps_per_chr=1200, seed=31, gz=True, span=span),
                             extra_gz={"TM_chr1.fa": mask_text(span, 1), "TM_chr2.fa": mask_text(span, 2, lower=True),
                                       "RM_chr1.fa": mask_text(span, 3, lower=True), "RM_chr2.fa": mask_text(2_000_000, 4)},
                             args=["--bins", "3,7,0.2", "--seed", "6", "--num_bootstraps", "4", "--chr", "chr.txt",
                                   "--target_mask", "TM", "--reference_mask", "RM"])
    for name, c in cases.items():
        d = os.path.join(HERE, name)
        shutil.rmtree(d, ignore_errors=True)
        with tempfile.TemporaryDirectory() as work:
            src = os.path.join(HERE, c["inputs_from"]) if "inputs_from" in c else None
            gen = c.get("gen") or cases[c["inputs_from"]]["gen"]
            if src:  # same input files as another case: stage that case (its files are stored once)
                import golden_lib
                golden_lib.l3_stage(c["inputs_from"], work)
                os.makedirs(d)
            else:
                synth_files.write_inputs(d, **gen)
                for fn, text in c.get("extra", {}).items():
                    open(os.path.join(d, fn), "w").write(text)
                for fn, text in c.get("extra_gz", {}).items():  # the readers fall back to <name>.gz (igzstream)
                    with gzip.GzipFile(os.path.join(d, fn + ".gz"), "wb", mtime=0) as g:
                        g.write(text.encode())
            run_dir = work if src else d
            mut_arg = "P" if gen.get("with_chr_file", True) else "P.mut"
            args = ["--mode", "mut", "--mut", mut_arg, "--target_tmp", "T.colate.in", "--reference_tmp", "R.colate.in"] + c["args"] + ["-o", "expected"]
            err, iters = run_ref(args, run_dir)
            if src:
                shutil.copy(os.path.join(work, "expected.coal"), os.path.join(d, "expected.coal"))
        nblocks = int(re.search(r"Number of blocks: (\d+)", err).group(1))
        meta = {"generator": "tests/golden/make_golden.py (oracle/_ref/Colate_ref)", "args": args, "iterations": iters,
                "num_blocks": nblocks}
        if src:
            meta["inputs_from"] = c["inputs_from"]
        if "--coal" in args:  # the starting rates the reference printed (coal.cpp:3642) pin the --coal reader on their own
            line = [l for l in err.split("\n") if l.strip() and all(_isnum(t) for t in l.split())]
            meta["init_rates_printed"] = line[-1].split()
        json.dump(meta, open(os.path.join(d, "case.json"), "w"))
        for fn in ("T.colate.in", "R.colate.in"):  # keep the fixtures small
            if os.path.exists(os.path.join(d, fn)):
                with open(os.path.join(d, fn), "rb") as f, gzip.GzipFile(os.path.join(d, fn + ".gz"), "wb", mtime=0) as g:
                    g.write(f.read())
                os.remove(os.path.join(d, fn))
        print(f"{name}: blocks={nblocks} iterations={iters}")


def make_l3_pairs():
    """Batched all-pairs (BASELINE configs[4]): the reference is run ONCE PER PAIR -- it has no
    list mode -- on samples that share one set of .mut files, every run with the same --seed; the fixture holds each
    pair's .coal and iteration counts.  `Colate --pairs` of this repo must write exactly these files.  Two of the pairs have
    an ancient sample: 500 years (one more epoch than the others, coal.cpp:3597-3624: a second launch) and 7000 years
    (same number of epochs, the ancient .coal layout)."""
    d = os.path.join(HERE, "l3_pairs")
    shutil.rmtree(d, ignore_errors=True)
    gen = dict(chroms=("1", "2", "3"), snps_per_chr=700, seed=41, gz=True, extra_targets=1, extra_refs=1)
    synth_files.write_inputs(d, **gen)
    common = ["--bins", "3,7,0.2", "--seed", "11", "--num_bootstraps", "3", "--chr", "chr.txt"]
    pairs = [("T.colate.in", "R.colate.in", "p0", "0", "0"), ("T1.colate.in", "R.colate.in", "p1", "500", "0"),
             ("T.colate.in", "R1.colate.in", "p2", "0", "0"), ("T1.colate.in", "R1.colate.in", "p3", "7000", "0"),
             ("R.colate.in", "T.colate.in", "p4", "0", "0"), ("R1.colate.in", "T1.colate.in", "p5", "0", "500")]
    meta = {"generator": "tests/golden/make_golden.py (oracle/_ref/Colate_ref, one run per pair)", "common_args": common, "pairs": []}
    for tgt, ref, out, ta, ra in pairs:
        args = ["--mode", "mut", "--mut", "P", "--target_tmp", tgt, "--reference_tmp", ref, "--target_age", ta, "--reference_age", ra] \
            + common + ["-o", "expected_" + out]
        err, iters = run_ref(args, d)
        nblocks = int(re.search(r"Number of blocks: (\d+)", err).group(1))
        meta["pairs"].append({"target": tgt, "reference": ref, "output": out, "target_age": ta, "reference_age": ra,
                              "iterations": iters, "num_blocks": nblocks})
        print(f"l3_pairs {out}: blocks={nblocks} iterations={iters}")
    json.dump(meta, open(os.path.join(d, "case.json"), "w"))
    for fn in sorted(os.listdir(d)):  # keep the fixtures small
        if fn.endswith(".colate.in"):
            with open(os.path.join(d, fn), "rb") as f, gzip.GzipFile(os.path.join(d, fn + ".gz"), "wb", mtime=0) as g:
                g.write(f.read())
            os.remove(os.path.join(d, fn))


def _isnum(t):
    try:
        float(t)
        return True
    except ValueError:
        return False


def make_l4():
    """Rows f4 of SURVEY.md section 8 that need no htslib: `--mode make_tmp --target_table` (coal.cpp:2682-2808, 2923-3069)
    and the .colate_mat the reference writes for BCF/BAM inputs (coal.cpp:3336-3343, 3453-3470; format pinned by the
    reference READING the file our CLI wrote, coal.cpp:3471-3499)."""
    import golden_lib

    CLI = os.path.join(ROOT, "colate_amd", "bin", "Colate")
    # ---- make_tmp from a table of haploid calls, on the .mut files and target masks of l3_masks
    d = os.path.join(HERE, "l4_maketmp")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    with tempfile.TemporaryDirectory() as work:
        golden_lib.l3_stage("l3_masks", work)
        rng = np.random.default_rng(77)
        rows = []
        for chrom in ("1", "2"):
            with gzip.open(os.path.join(work, f"P_chr{chrom}.mut.gz"), "rt") as f:
                next(f)
                for line in f:
                    t = line.split(";")
                    bp, (anc, der) = int(t[1]), (t[10].split("/") + [""])[:2]
                    u = rng.uniform()
                    if u < 0.8:  # the target has a call at this site: ancestral, derived or (rarely) a third allele
                        v = rng.uniform()
                        allele = der if v < 0.45 else (anc if v < 0.9 else "ACGT"[rng.integers(4)])
                        rows.append(f"{chrom} {bp} {allele}")
                    if rng.uniform() < 0.15:  # calls at sites without a dated mutation
                        rows.append(f"{chrom} {bp + 1} {'ACGT'[rng.integers(4)]}")
        table = "\n".join(rows) + "\n"
        with gzip.GzipFile(os.path.join(d, "table.txt.gz"), "wb", mtime=0) as g:
            g.write(table.encode())
        open(os.path.join(work, "table.txt"), "w").write(table)
        for chrom in ("1", "2"):  # the reference opens (and requires) the reference genome; only its presence matters here
            for where in (d, work):
                open(os.path.join(where, f"G_chr{chrom}.fa"), "w").write(f">chr{chrom}\nACGTACGTNN\n")
        args = ["--mode", "make_tmp", "--mut", "P", "--chr", "chr.txt", "--target_table", "table.txt", "--ref_genome", "G",
                "--target_mask", "TM", "-o", "expected"]
        r = subprocess.run([REF_BIN] + args, cwd=work, capture_output=True)
        assert r.returncode == 0, r.stderr.decode()[-1000:]
        blob = open(os.path.join(work, "expected.colate.in"), "rb").read()
        with gzip.GzipFile(os.path.join(d, "expected.colate.in.gz"), "wb", mtime=0) as g:
            g.write(blob)
        # and the same without the mask (other branch of coal.cpp:2749)
        args2 = [a for i, a in enumerate(args) if a != "--target_mask" and (i == 0 or args[i - 1] != "--target_mask")]
        args2[args2.index("-o") + 1] = "expected_nomask"
        r = subprocess.run([REF_BIN] + args2, cwd=work, capture_output=True)
        assert r.returncode == 0, r.stderr.decode()[-1000:]
        blob2 = open(os.path.join(work, "expected_nomask.colate.in"), "rb").read()
        with gzip.GzipFile(os.path.join(d, "expected_nomask.colate.in.gz"), "wb", mtime=0) as g:
            g.write(blob2)
    json.dump({"generator": "tests/golden/make_golden.py (oracle/_ref/Colate_ref --mode make_tmp)", "inputs_from": "l3_masks",
               "args": args, "args_nomask": args2, "records": len(blob) // 18, "records_nomask": len(blob2) // 18},
              open(os.path.join(d, "case.json"), "w"))
    print(f"l4_maketmp: {len(blob) // 18} records with mask, {len(blob2) // 18} without")
    # ---- .colate_mat written by OUR CLI (--write_colate_mat) and read by the REFERENCE
    d = os.path.join(HERE, "l4_colate_mat")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    with tempfile.TemporaryDirectory() as work:
        case = golden_lib.l3_stage("l3_modern", work)
        args = list(case["args"])
        args[args.index("-o") + 1] = "OUT"
        r = subprocess.run([CLI] + args + ["--write_colate_mat", "--counts_only"], cwd=work, capture_output=True)
        assert r.returncode == 0, r.stderr.decode()[-1000:]
        shutil.copy(os.path.join(work, "OUT.colate_mat"), os.path.join(d, "OUT.colate_mat"))
        B = args[args.index("--num_bootstraps") + 1]
        ref_args = ["--mode", "mut", "--mut", "dummy", "--bins", args[args.index("--bins") + 1], "--num_bootstraps", B, "-o", "OUT"]
        err, iters = run_ref(ref_args, work)
        assert "Loading precomputed file" in err
        shutil.copy(os.path.join(work, "OUT.coal"), os.path.join(d, "expected.coal"))
    json.dump({"generator": "tests/golden/make_golden.py (our CLI --write_colate_mat, then oracle/_ref/Colate_ref on that file)",
               "inputs_from": "l3_modern", "writer_args": args + ["--write_colate_mat", "--counts_only"], "reader_args": ref_args,
               "iterations": iters}, open(os.path.join(d, "case.json"), "w"))
    print(f"l4_colate_mat: iterations={iters}")


if __name__ == "__main__":
    assert os.path.exists(REF_BIN), "oracle/_ref/Colate_ref missing: make -C oracle ref (needs /root/reference)"
    make_l1()
    make_l2()
    make_l3()
    make_l3_pairs()
    make_l4()
