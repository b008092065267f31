"""Host side of the drop-in (C ABI + `Colate` command line) on the CPU: symbol table, the pieces of
mut() around the EM (age grid, epochs, block bootstrap + F redistribution, .coal writer, readers) and
the full CLI up to the count tables, checked against the oracle and the golden fixtures."""
import ctypes
import json
import os
import re
import subprocess

import numpy as np
import pytest

import golden_lib as gl
import synth_files
import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "colate_amd", "bin", "Colate")
HOOKS_LIB_DIR = os.path.join(ROOT, "colate_amd", "lib", "testhooks")  # the library built with -DCOLATE_TEST_HOOKS


@pytest.fixture(scope="module")
def ca():
    import colate_amd

    return colate_amd


def test_library_exports_every_declared_symbol(ca):
    header = open(os.path.join(ROOT, "include", "colate_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(colate_[a-z_]+)\s*\(", header))
    assert len(declared) >= 15
    from colate_amd._lib import SIGNATURES, lib

    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/colate_amd.h but not exported"
    assert declared == set(SIGNATURES), declared ^ set(SIGNATURES)
    assert "gfx950" in ca.version()


def test_no_cpu_fallback(ca):
    """Without a HIP device the compute entry points must fail loudly (never compute on the CPU)."""
    if ca.device_count() > 0:
        pytest.skip("a GPU is present")
    grid = ca.age_grid()
    ep, _ = ca.epochs_from_bins("3,7,0.2")
    with pytest.raises(ca.ColateError) as e:
        ca.em_batch(grid, np.ones((1, grid.size)), np.ones((1, grid.size)), ep)
    assert e.value.code == -2
    with pytest.raises(ca.ColateError):
        ca.em_estep(grid, np.ones((1, grid.size)), np.ones((1, grid.size)), ep, np.full((1, ep.size), 1e-4))


def test_argument_validation(ca):
    grid = ca.age_grid()
    ep, _ = ca.epochs_from_bins("3,7,0.2")
    one = np.ones((1, grid.size))
    with pytest.raises(ca.ColateError) as e:  # unsorted epochs
        ca.em_batch(grid, one, one, ep[::-1].copy())
    assert e.value.code == -1
    with pytest.raises(ca.ColateError) as e:  # too many epochs (the compiled limit is 1024: 16 per lane of a wave)
        ca.em_batch(grid, one, one, np.arange(1100.0))
    assert e.value.code == -4
    with pytest.raises(ca.ColateError):  # bad --bins
        ca.epochs_from_bins("3,7")


def test_age_grid_matches_oracle(ca):
    g = ca.age_grid()
    assert g.size == 185 and np.array_equal(g, ol.age_grid())
    assert g[0] == 0.0 and g[1] == 0.1 and g[184] == np.exp(18.3) / 10.0


@pytest.mark.parametrize("bins", ["3,7,0.2", "2,7.95,0.05", "3,6,0.5", "2.5,7,0.3", "4,6.1,0.7"])
@pytest.mark.parametrize("age_years", [0.0, 500.0, 1000.0, 1200.0, 7000.0, 45000.0])
def test_epochs_from_bins_matches_oracle(ca, bins, age_years):
    age = age_years / 28.0
    e1, n1 = ca.epochs_from_bins(bins, age, 28.0)
    e0, n0 = ol.epochs_from_bins(bins, age, 28.0)
    assert n1 == n0 and np.array_equal(e1, e0)
    assert e1[0] == 0.0 and np.all(np.diff(e1) >= 0)


def test_epochs_23_and_122(ca):
    assert ca.epochs_from_bins("3,7,0.2")[0].size == 23     # BASELINE configs[0..2]
    assert ca.epochs_from_bins("2,7.95,0.05")[0].size == 122  # configs[3] (2,8,0.05 aborts the reference)


def test_bins_2_8_005_is_rejected_like_the_reference_aborts(ca):
    """BASELINE configs[3] as written, `--bins 2,8,0.05`: the float-accumulated start of the last regular epoch,
    exp(ln10 * 8)/28 = 3571428.571428578, exceeds the hard-coded final epoch 1e8/28 (coal.cpp:3628-3629) and the
    reference aborts at coal_EM.cpp:114 (assert t_end >= t_begin).  Here the epoch builder returns the same
    (decreasing) grid and every EM entry point refuses it -- before touching a device."""
    ep, _ = ca.epochs_from_bins("2,8,0.05")
    assert ep.size == 123 and ep[-1] < ep[-2]
    grid = ca.age_grid()
    z = np.zeros((1, grid.size))
    with pytest.raises(ca.ColateError) as e:
        ca.em_batch(grid, z + 1.0, z + 1.0, ep)
    assert e.value.code == -1 and "non-decreasing" in str(e.value)


def test_block_bootstrap_matches_oracle_and_restated_rng(ca):
    """std::mt19937 + uniform_int_distribution in the product vs the oracle's restatement of both
    (Matsumoto-Nishimura + libstdc++-11 Lemire), then the weighted sums and the F redistribution."""
    rng = np.random.default_rng(5)
    grid = ol.age_grid()
    A, nb, B = grid.size, 17, 6
    sh = rng.uniform(0, 3, (nb, A)) * (rng.uniform(size=(nb, A)) < 0.6)
    ns = rng.uniform(0, 9, (nb, A)) * (rng.uniform(size=(nb, A)) < 0.6)
    she = rng.uniform(0, 1, (nb, A)) * (rng.uniform(size=(nb, A)) < 0.2)
    nse = rng.uniform(0, 1, (nb, A)) * (rng.uniform(size=(nb, A)) < 0.2)
    for age in (0.0, 250.0):
        for nboot in (1, B):
            c_sh, c_ns = ca.bootstrap_counts(ca.Rng(12345), nboot, grid, age, sh, ns, she, nse)
            g = (ctypes.c_uint * 625)()
            ol.O.oracle_mt_seed(g, 12345)
            for i in range(nboot):
                w = np.zeros(nb)
                ol.O.oracle_block_weights(g, nb, nboot, ol.P(w))
                o_sh, o_ns = np.zeros(A), np.zeros(A)
                ol.O.oracle_bootstrap_counts(nb, A, ol.P(grid), age, ol.P(w), ol.P(sh), ol.P(ns), ol.P(she), ol.P(nse),
                                             ol.P(o_sh), ol.P(o_ns))
                assert np.array_equal(c_sh[i], o_sh) and np.array_equal(c_ns[i], o_ns)


def test_write_coal_format(ca, tmp_path):
    ep = np.array([0.0, 35.7142857, 56.6033287, 1e5 / 3, 3571428.5714285714])
    rates = np.array([[0.0, 6.66486e-05, 5e-9, 1.23456789e-7, 5e-05], [0.0, 1e-4, 2e-4, 3e-4, 4e-4]])
    p = tmp_path / "x.coal"
    ca.write_coal(p, ep, rates)
    assert p.read_text() == gl.coal_text(ep, rates)
    assert p.read_text().split("\n")[1] == "0 35.7143 56.6033 33333.3 3.57143e+06 "
    ca.write_coal(p, ep, rates, is_ancient=True, ep_null=1)
    assert p.read_text() == gl.coal_text(ep, rates, True, 1)
    # and back: --coal gives epochs (through std::stof) and starting rates
    ca.write_coal(p, ep, rates[:1])
    e2, r2 = ca.epochs_from_coal(p)
    assert np.array_equal(e2, np.array([float(np.float32(float("%g" % x))) for x in ep]))
    assert np.array_equal(r2, np.array([float("%g" % x) for x in rates[0]]))


def _run_cli(args, cwd, env=None):
    return subprocess.run([CLI] + args, cwd=cwd, capture_output=True, env=env)


@pytest.mark.parametrize("name", gl.l3_names())
def test_cli_feeder_and_bootstrap_reproduce_reference(name, tmp_path):
    """Full host path of `Colate --mode mut` (.mut/.colate.in readers, age sampling from the shared
    mt19937, 30-Mb blocks, block bootstrap, F redistribution) -> count tables; the oracle's EM on those
    tables must print exactly the .coal the reference CLI printed for the same files and --seed."""
    case = gl.l3_stage(name, str(tmp_path))
    args = [a for a in case["args"]]
    args[args.index("-o") + 1] = "mine"
    B = int(args[args.index("--num_bootstraps") + 1])
    r = _run_cli(args + ["--counts_out", "mine.counts", "--counts_only"], str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-800:]
    assert f"Number of blocks: {case['num_blocks']}" in r.stderr.decode()
    grid, csh, cns = gl.read_counts(tmp_path / "mine.counts", B)
    assert np.array_equal(grid, ol.age_grid())
    age = 0.0
    if "--target_age" in args:
        ref_age = args[args.index("--reference_age") + 1] if "--reference_age" in args else "0"
        age = max(float(np.float32(args[args.index("--target_age") + 1])), float(np.float32(ref_age))) / 28.0
    init = None
    if "--coal" in args:  # warm start: epochs (float-parsed) and starting rates from the file (coal.cpp:3508-3549, 3638-3646)
        prev = tmp_path / args[args.index("--coal") + 1]
        ep, init = ol.epochs_from_coal(prev, age)
        ep_null = 0
        import colate_amd

        ep_p, init_p = colate_amd.epochs_from_coal(prev, age)  # the product's reader against the oracle's restatement ...
        assert np.array_equal(ep_p, ep) and np.array_equal(init_p, init)
        assert ["%g" % x for x in init] == case["init_rates_printed"]  # ... and both against what the reference printed
    else:
        ep, ep_null = ol.epochs_from_bins(args[args.index("--bins") + 1], age, 28.0)
    rates, iters, ll, fl = ol.em_batch(grid, csh, cns, ep, init=init)
    assert iters.tolist() == case["iterations"]
    assert gl.coal_text(ep, rates, age > 0, ep_null) == (tmp_path / "expected.coal").read_text()


def test_cli_option_errors(tmp_path):
    r = _run_cli(["--mode", "mut", "--nonsense", "1"], str(tmp_path))
    assert r.returncode != 0 and b"does not exist" in r.stderr
    r = _run_cli(["--mode", "mut"], str(tmp_path))
    assert b"Not enough arguments supplied." in r.stdout
    r = _run_cli(["--mode", "preprocess_mut"], str(tmp_path))
    assert r.returncode != 0
    r = _run_cli(["--mode", "make_tmp", "--mut", "P", "-o", "x", "--target_bcf", "y"], str(tmp_path))
    assert r.returncode != 0 and b"htslib" in r.stderr
    # README spelling --num_bootstrap is accepted (the reference only knows --num_bootstraps)
    case = gl.l3_stage("l3_nochr", str(tmp_path))
    args = [("--num_bootstrap" if a == "--num_bootstraps" else a) for a in case["args"]]
    r = _run_cli(args + ["--counts_out", "c.txt", "--counts_only"], str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-500:]


def test_pairs_mode_counts_equal_separate_runs(tmp_path):
    """`--pairs` (batched all-pairs, configs[4]): every pair's count tables are exactly what a run of
    that pair alone produces (RNG re-seeded per pair; the .mut files are parsed once)."""
    case = gl.l3_stage("l3_ancient", str(tmp_path))
    common = ["--mode", "mut", "--mut", "P", "--chr", "chr.txt", "--bins", "3,7,0.2", "--seed", "5", "--num_bootstraps", "2"]
    specs = [("T.colate.in", "R.colate.in", "ab", "7000", "0"), ("R.colate.in", "T.colate.in", "ba", "0", "0")]
    (tmp_path / "pairs.txt").write_text("".join(" ".join(sp) + "\n" for sp in specs))
    r = _run_cli(common + ["--pairs", "pairs.txt", "--counts_only"], str(tmp_path), env=dict(os.environ, COLATE_TIMING="1"))
    assert r.returncode == 0, r.stderr.decode()[-800:]
    # every input file is read once, whatever number of pairs it takes part in (here: both files in both pairs)
    assert "3 .mut files" in r.stderr.decode() and "2 .colate.in files" in r.stderr.decode(), r.stderr.decode()[-800:]
    for tgt, ref, out, ta, ra in specs:
        # (COLATE_SINGLE_FEEDER=1: the pair alone through the single-pair feeder, not through the engine of the batched front end)
        r = _run_cli(common + ["--target_tmp", tgt, "--reference_tmp", ref, "--target_age", ta, "--reference_age", ra,
                               "-o", out + "_single", "--counts_out", out + "_single.counts", "--counts_only"], str(tmp_path),
                     env=dict(os.environ, COLATE_SINGLE_FEEDER="1"))
        assert r.returncode == 0, r.stderr.decode()[-800:]
        assert (tmp_path / (out + ".counts")).read_text() == (tmp_path / (out + "_single.counts")).read_text()


def test_pairs_with_a_file_that_ends_inside_a_record(tmp_path):
    """A .colate.in that ends in the middle of a record: the reference's fread calls leave the fields they do not reach as they were
    (coal.cpp:2126-2133).  The batched front end decodes every file once into fixed records -- but not such a file, whose last record
    depends on what the walk's variables held: its walks go through the byte cursor, and the tables are those of the single-pair feeder."""
    case = gl.l3_stage("l3_modern", str(tmp_path))
    for name, cut in (("T.colate.in", 7), ("R.colate.in", 3)):
        raw = (tmp_path / name).read_bytes()
        (tmp_path / name).write_bytes(raw[:len(raw) - cut])
    common = ["--mode", "mut", "--mut", "P", "--chr", "chr.txt", "--bins", "3,7,0.2", "--seed", "5", "--num_bootstraps", "2"]
    specs = [("T.colate.in", "R.colate.in", "ab"), ("R.colate.in", "T.colate.in", "ba")]
    (tmp_path / "pairs.txt").write_text("".join(" ".join(sp) + "\n" for sp in specs))
    r = _run_cli(common + ["--pairs", "pairs.txt", "--counts_only"], str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-800:]
    for tgt, ref, out in specs:
        r = _run_cli(common + ["--target_tmp", tgt, "--reference_tmp", ref, "-o", out + "_single", "--counts_out", out + "_single.counts",
                               "--counts_only"], str(tmp_path), env=dict(os.environ, COLATE_SINGLE_FEEDER="1"))
        assert r.returncode == 0, r.stderr.decode()[-800:]
        assert (tmp_path / (out + ".counts")).read_text() == (tmp_path / (out + "_single.counts")).read_text()


def _drop_out_of_order_records(path, dup_every=0):
    """Rewrites a .colate.in without the records whose position lies below that of the record in front of them (same chromosome);
    `dup_every`: every so many records one is written twice with other counts (equal positions are within what the indices handle)."""
    import struct
    b = open(path, "rb").read()
    i, out, last = 0, bytearray(), None
    while i + 4 <= len(b):
        (l,) = struct.unpack_from("<i", b, i)
        name = b[i + 4:i + 4 + l]
        (bp,) = struct.unpack_from("<i", b, i + 4 + l)
        rec = b[i:i + 4 + l + 14]
        i += 4 + l + 14
        if last is not None and last[0] == name and bp < last[1]:
            continue
        out += rec
        last = (name, bp)
        n_out = getattr(_drop_out_of_order_records, "_n", 0) + 1
        _drop_out_of_order_records._n = n_out
        if dup_every and n_out % dup_every == 0:
            out += rec[:-8] + struct.pack("<ii", 1, 1)  # the same site again, AAF = DAF = 1
    open(path, "wb").write(bytes(out))


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_indexed_walk_equals_the_cursor_walk(tmp_path, seed):
    """Where every file of a pair is well-formed (each chromosome one run of records in ascending positions) the batched front end walks
    through per-file indices instead of the two cursors of coal.cpp:2125-2243 -- same used SNPs, same tables: against the cursor walk
    (COLATE_INDEXED_WALK=0, the code the reference-made fixtures pin) on inputs with absent records, allele mismatches, DAF = 0 and rows
    whose record an earlier row's search had already reached, three targets x two references, several stream windows."""
    d = str(tmp_path)
    synth_files.write_inputs(d, chroms=("1", "2", "3", "4"), snps_per_chr=6000, seed=seed, span=70_000_000, extra_targets=2, extra_refs=1)
    for f in ("T", "T1", "T2", "R", "R1"):
        _drop_out_of_order_records(os.path.join(d, f + ".colate.in"), dup_every=37 if seed != 11 else 0)
    pairs = [(f"{t}.colate.in", f"{r}.colate.in", f"out_{t}_{r}") for t in ("T", "T1", "T2") for r in ("R", "R1")]
    (tmp_path / "pairs.txt").write_text("".join(" ".join(p) + "\n" for p in pairs))
    args = ["--mode", "mut", "--mut", "P", "--chr", "chr.txt", "--bins", "3,7,0.2", "--seed", "4", "--num_bootstraps", "2", "--pairs", "pairs.txt", "--counts_only"]
    env = dict(os.environ, COLATE_TIMING="1", COLATE_UNIFORM_WINDOW_MB="4")
    r = _run_cli(args, d, env=dict(env, COLATE_INDEXED_WALK="0"))
    assert r.returncode == 0 and "(0 of 5 files)" in r.stderr.decode(), r.stderr.decode()[-800:]
    want = {p[2]: (tmp_path / (p[2] + ".counts")).read_bytes() for p in pairs}
    for p in pairs:
        os.remove(tmp_path / (p[2] + ".counts"))
    r = _run_cli(args, d, env=env)
    assert r.returncode == 0 and "(5 of 5 files)" in r.stderr.decode(), r.stderr.decode()[-800:]
    for p in pairs:
        assert (tmp_path / (p[2] + ".counts")).read_bytes() == want[p[2]], p[2]
    assert len(set(want.values())) == len(pairs)  # (six different pairs, six different tables)


def test_pairs_fixture_counts_reproduce_reference(tmp_path):
    """`--pairs` against the REFERENCE (fixture l3_pairs: Colate_ref run once per pair, same --seed): the count tables of
    every pair, through the oracle's EM, print exactly the .coal the reference wrote for that pair, with its iteration
    counts -- modern pairs, a 500-year-old target (one epoch more: a launch of its own) and a 7000-year-old one."""
    meta = gl.l3_pairs_stage(str(tmp_path))
    common = ["--mode", "mut", "--mut", "P"] + meta["common_args"]
    B = int(common[common.index("--num_bootstraps") + 1])
    r = _run_cli(common + ["--pairs", "pairs.txt", "--counts_only"], str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-800:]
    for p in meta["pairs"]:
        assert f"{p['target']} x {p['reference']}: Number of blocks: {p['num_blocks']}" in r.stderr.decode()
        grid, csh, cns = gl.read_counts(tmp_path / (p["output"] + ".counts"), B)
        age = max(float(np.float32(p["target_age"])), float(np.float32(p["reference_age"]))) / 28.0
        ep, ep_null = ol.epochs_from_bins(common[common.index("--bins") + 1], age, 28.0)
        rates, iters, ll, fl = ol.em_batch(grid, csh, cns, ep)
        assert iters.tolist() == p["iterations"], p["output"]
        assert gl.coal_text(ep, rates, age > 0, ep_null) == (tmp_path / f"expected_{p['output']}.coal").read_text(), p["output"]


def test_masks_change_the_tables(tmp_path):
    """The mask fixture really exercises the mask branch (coal.cpp:2169-2174): without the two masks the same
    inputs give different count tables."""
    case = gl.l3_stage("l3_masks", str(tmp_path))
    args = [a for a in case["args"]]
    args[args.index("-o") + 1] = "mine"
    r = _run_cli(args + ["--counts_out", "with.counts", "--counts_only"], str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-800:]
    nomask = [a for i, a in enumerate(args) if a not in ("--target_mask", "--reference_mask")
              and (i == 0 or args[i - 1] not in ("--target_mask", "--reference_mask"))]
    r = _run_cli(nomask + ["--counts_out", "without.counts", "--counts_only"], str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-800:]
    _, sh1, ns1 = gl.read_counts(tmp_path / "with.counts", 4)
    _, sh2, ns2 = gl.read_counts(tmp_path / "without.counts", 4)
    assert (sh1 + ns1).sum() < 0.9 * (sh2 + ns2).sum()  # the masks remove sites


def test_coal_file_written_by_bins_run_is_refused_like_the_reference(tmp_path):
    """A .coal written by `--mode mut --bins` starts "0 0 ...": the reference asserts strictly increasing epochs
    (coal.cpp:3544-3546) and aborts on it; the library returns an error instead."""
    import colate_amd

    with pytest.raises(colate_amd.ColateError):
        colate_amd.epochs_from_coal(os.path.join(gl.HERE, "l3_modern", "expected.coal"))


def test_pairs_refuses_masks_and_coal(tmp_path):
    """--pairs shares one option set over many samples: per-sample masks and a --coal warm start have no meaning
    there and are refused (instead of being silently ignored)."""
    case = gl.l3_stage("l3_masks", str(tmp_path))
    (tmp_path / "pairs.txt").write_text("T.colate.in R.colate.in ab\n")
    base = ["--mode", "mut", "--mut", "P", "--chr", "chr.txt", "--bins", "3,7,0.2", "--seed", "5", "--pairs", "pairs.txt", "--counts_only"]
    for extra in (["--target_mask", "TM"], ["--reference_mask", "RM"], ["--coal", "x.coal"]):
        r = _run_cli(base + extra, str(tmp_path))
        assert r.returncode != 0 and b"--pairs" in r.stderr, r.stderr.decode()[-300:]


@pytest.mark.parametrize("name", ["wg_e23", "wg_e122"])
def test_colate_mat_loader_round_trip(name, tmp_path):
    """load_colate_mat (the reference's precomputed-table hook, coal.cpp:3471-3499) without a GPU: the tables our
    CLI loads from OUT.colate_mat come back bit for bit through --counts_out (both sides 17 significant digits)."""
    c = gl.l2_case(name)
    B = len(c["iterations"])
    gl.write_colate_mat(tmp_path / "OUT.colate_mat", ol.age_grid(), c["csh"], c["cns"])
    r = _run_cli(["--mode", "mut", "--mut", "dummy", "--bins", c["bins"], "--num_bootstraps", str(B), "-o", "OUT",
                  "--counts_out", "back.counts", "--counts_only"], str(tmp_path))
    assert r.returncode == 0 and b"Loading precomputed file OUT.colate_mat" in r.stderr, r.stderr.decode()[-500:]
    grid, csh, cns = gl.read_counts(tmp_path / "back.counts", B)
    assert np.array_equal(grid, ol.age_grid()) and np.array_equal(csh, c["csh"]) and np.array_equal(cns, c["cns"])


@pytest.mark.parametrize("which", ["args", "args_nomask"])
def test_make_tmp_from_table_matches_reference(which, tmp_path):
    """`--mode make_tmp --target_table` (coal.cpp:2682-2808, 2923-3069): the .colate.in our CLI writes from a table of
    haploid calls is byte for byte the reference's, with and without a target mask."""
    import gzip
    import shutil

    gl.l3_stage("l3_masks", str(tmp_path))
    src = os.path.join(gl.HERE, "l4_maketmp")
    case = json.load(open(os.path.join(src, "case.json")))
    with gzip.open(os.path.join(src, "table.txt.gz"), "rb") as g:
        (tmp_path / "table.txt").write_bytes(g.read())
    for f in ("G_chr1.fa", "G_chr2.fa"):
        shutil.copy(os.path.join(src, f), str(tmp_path / f))
    args = list(case[which])
    out = args[args.index("-o") + 1]
    args[args.index("-o") + 1] = "mine"
    r = _run_cli(args, str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-800:]
    assert b"parsing CHR: 2 / 2" in r.stderr
    with gzip.open(os.path.join(src, out + ".colate.in.gz"), "rb") as g:
        want = g.read()
    got = (tmp_path / "mine.colate.in").read_bytes()
    assert len(want) > 18 * 1000 and got == want


def test_make_tmp_output_feeds_mode_mut(tmp_path):
    """The two halves together: make_tmp's .colate.in as --target_tmp of `--mode mut` (tables only, no GPU here)."""
    import gzip
    import shutil

    gl.l3_stage("l3_masks", str(tmp_path))
    src = os.path.join(gl.HERE, "l4_maketmp")
    with gzip.open(os.path.join(src, "table.txt.gz"), "rb") as g:
        (tmp_path / "table.txt").write_bytes(g.read())
    for f in ("G_chr1.fa", "G_chr2.fa"):
        shutil.copy(os.path.join(src, f), str(tmp_path / f))
    r = _run_cli(["--mode", "make_tmp", "--mut", "P", "--chr", "chr.txt", "--target_table", "table.txt", "--ref_genome", "G",
                  "-o", "T2"], str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-500:]
    r = _run_cli(["--mode", "mut", "--mut", "P", "--chr", "chr.txt", "--target_tmp", "T2.colate.in", "--reference_tmp",
                  "R.colate.in", "--bins", "3,7,0.2", "--seed", "2", "--num_bootstraps", "2", "-o", "x", "--counts_out",
                  "x.counts", "--counts_only"], str(tmp_path))
    assert r.returncode == 0 and b"Number of blocks: 2" in r.stderr, r.stderr.decode()[-500:]
    _, csh, cns = gl.read_counts(tmp_path / "x.counts", 2)
    assert csh.sum() > 0 and cns.sum() > 0


def test_colate_mat_writer_is_read_by_the_reference(tmp_path):
    """--write_colate_mat (what the reference writes for BCF/BAM inputs, coal.cpp:3336-3343, 3453-3470: counts / 1e3,
    6 significant digits): our CLI reproduces the committed file, which the reference binary loaded as "precomputed
    file" (coal.cpp:3471-3499) when the fixture was made; the oracle's EM on the file's numbers prints the reference's
    .coal."""
    src = os.path.join(gl.HERE, "l4_colate_mat")
    case = json.load(open(os.path.join(src, "case.json")))
    gl.l3_stage("l3_modern", str(tmp_path))
    r = _run_cli(case["writer_args"], str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-800:]
    assert (tmp_path / "OUT.colate_mat").read_text() == open(os.path.join(src, "OUT.colate_mat")).read()
    B = len(case["iterations"])
    grid, csh, cns = gl.read_counts(tmp_path / "OUT.colate_mat", B)
    ep, _ = ol.epochs_from_bins("3,7,0.2")
    rates, iters, _, _ = ol.em_batch(grid, csh, cns, ep)
    assert iters.tolist() == case["iterations"]
    assert gl.coal_text(ep, rates) == open(os.path.join(src, "expected.coal")).read()


def test_allgather_entry_points_check_their_grids():
    """colate_em_batch_allgather / colate_bootstrap_em_batch_allgather refuse unsorted grids like every other
    host-pointer entry point (before any rank could enter the collective) -- checked without a communicator."""
    from colate_amd._lib import lib

    A, E, B, nb = 185, 23, 2, 3
    grid = np.ascontiguousarray(ol.age_grid())
    ep, _ = ol.epochs_from_bins("3,7,0.2", 0.0, 28.0)
    bad = np.ascontiguousarray(ep[::-1])
    cnt = np.ones((B, A))
    init = np.full(E, 5e-5)
    rates, ll = np.zeros((B, E)), np.zeros(B)
    iters, flags = np.zeros(B, np.int32), np.zeros(B, np.int32)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)

    def em(epochs):
        return lib.colate_em_batch_allgather(None, B, E, A, ptr(grid), ptr(cnt), ptr(cnt), ptr(epochs), ptr(init), 10, 0,
                                             1e-7, 5e-9, ptr(rates), ptr(iters), ptr(ll), ptr(flags))

    assert em(bad) == -1 and b"non-decreasing" in lib.colate_last_error()
    assert em(np.ascontiguousarray(ep)) == -1 and b"communicator" in lib.colate_last_error()
    tabs = np.ones((nb, A))
    w = np.ones((B, nb))
    rc = lib.colate_bootstrap_em_batch_allgather(None, B, nb, E, A, ptr(grid), 0.0, ptr(w), ptr(tabs), ptr(tabs), ptr(tabs),
                                                 ptr(tabs), ptr(bad), ptr(init), 10, 0, 1e-7, 5e-9, ptr(rates), ptr(iters),
                                                 ptr(ll), ptr(flags))
    assert rc == -1 and b"non-decreasing" in lib.colate_last_error()


def test_status_flags_macro_matches_python(ca):
    header = open(os.path.join(ROOT, "include", "colate_amd.h")).read()
    m = re.search(r"#define COLATE_STATUS_FLAGS\(flags\) \(\(flags\) & (0x[0-9a-f]+)\)", header)
    assert m and int(m.group(1), 16) == ca.STATUS_MASK == 0x07  # UNRESOLVED (8) is not an error bit


def test_ranks_launcher_ends_waiting_ranks_after_a_failure(tmp_path):
    """`Colate --ranks 2` where rank 1 hangs (as a rank blocked in a collective whose peer died would) and rank 0
    fails: the launcher must kill the survivor after the grace period and exit non-zero instead of waiting forever."""
    import time

    mat = tmp_path / "out.colate_mat"  # the .colate_mat hook: no readers needed to reach the ranked section
    grid = ol.age_grid()
    with open(mat, "w") as f:
        f.write(" ".join(repr(float(x)) for x in grid) + "\n")
        for _ in range(2):
            f.write(" ".join("1" for _ in grid) + "\n")
            f.write(" ".join("2" for _ in grid) + "\n")
    cmd = [CLI, "--mode", "mut", "--mut", "dummy", "--bins", "3,7,0.2", "--num_bootstraps", "2", "--seed", "1",
           "--ranks", "2", "-o", str(tmp_path / "out")]
    # (the hooks exist in lib/testhooks/libcolate_amd.so only: the product library ignores these variables)
    env = dict(os.environ, LD_LIBRARY_PATH=HOOKS_LIB_DIR, COLATE_TEST_HANG_RANK="1", COLATE_RANK_GRACE_SEC="1",
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=60)
    took = time.time() - t0
    assert r.returncode != 0
    assert "rank 0 failed" in r.stderr and "ended by the launcher" in r.stderr, r.stderr
    assert took < 30
    # Rank 0 itself hangs before it has published the communicator id, rank 1 waits for the id: nobody ever exits, so
    # only the bound on the wait for the id can end the run (the relay used to block in front of the watchdog).
    # (Without a device rank 1 fails on its own before it waits; the grace period is set long so that the bound on the id is
    # what ends the run, as it must where rank 1 does wait: tests/test_gpu_multi_device.py has the two-GPU form.)
    env.update(COLATE_TEST_HANG_RANK="0", COLATE_RANK_ID_TIMEOUT_SEC="2", COLATE_RANK_GRACE_SEC="60")
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=60)
    took = time.time() - t0
    assert r.returncode != 0 and "has not published the communicator id" in r.stderr, r.stderr
    assert "rank 0 was ended by the launcher" in r.stderr and took < 30, (took, r.stderr)


def test_product_library_has_no_test_hooks():
    """The failure-injection / hang hooks are compiled into lib/testhooks/libcolate_amd.so only."""
    prod = open(os.path.join(ROOT, "colate_amd", "lib", "libcolate_amd.so"), "rb").read()
    hooks = open(os.path.join(HOOKS_LIB_DIR, "libcolate_amd.so"), "rb").read()
    for name in (b"COLATE_TEST_HANG_RANK", b"COLATE_TEST_FAIL_RANK"):
        assert name not in prod and name in hooks


def test_ranks_refused_once_the_process_has_used_the_device(ca, tmp_path):
    """colate_mut_main --ranks forks one process per GPU: refused (never forked, never re-exec'd) in a process whose HIP
    runtime is already up."""
    code = (
        "import sys, ctypes\n"
        "from colate_amd._lib import lib\n"
        "assert lib.colate_device_touched() == 0\n"
        "lib.colate_device_count()\n"
        "assert lib.colate_device_touched() == 1\n"
        "argv = [b'Colate', b'--mode', b'mut', b'--mut', b'x', b'--bins', b'3,7,0.2', b'--seed', b'1', b'--ranks', b'2', b'-o', sys.argv[1].encode()]\n"
        "arr = (ctypes.c_char_p * len(argv))(*argv)\n"
        "sys.exit(lib.colate_mut_main(len(argv), arr))\n"
    )
    r = subprocess.run([os.sys.executable, "-c", code, str(tmp_path / "o")], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "must run before this process first uses a GPU" in r.stderr, (r.returncode, r.stderr)


def _counts(tmp_path, threads, extra=(), single_feeder=False):
    """single_feeder: the pair through the single-pair feeder of mut_driver.cpp (reader threads, uniform-stream thread, sampling
    workers) instead of the engine of the batched front end (mut_pairs.cpp), which is what a pair without masks takes by default."""
    env = dict(os.environ, COLATE_THREADS=str(threads), COLATE_TIMING="1")
    if single_feeder:
        env["COLATE_SINGLE_FEEDER"] = "1"
    out = f"c{threads}{'s' if single_feeder else ''}"
    r = subprocess.run([CLI, "--mode", "mut", "--mut", "P", "--target_tmp", "T.colate.in", "--reference_tmp", "R.colate.in", "--chr",
                        "chr.txt", "--bins", "3,7,0.2", "--seed", "11", "--num_bootstraps", "7", "--counts_only", "--counts_out",
                        out + ".counts", "-o", out] + list(extra), cwd=str(tmp_path), capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-500:]
    assert "Timing: parse_mut" in r.stderr
    _counts.redone = "repeated sequentially" in r.stderr or "1 pair(s) redone sequentially" in r.stderr
    return (tmp_path / (out + ".counts")).read_text()


def test_threaded_table_fill_is_bit_identical_to_sequential(tmp_path):
    """The reader threads, the uniform-stream thread and the sampling workers (mut_driver.cpp) leave the count tables -- and
    the std::mt19937 state the bootstrap weights are drawn from afterwards -- exactly as the sequential code does."""
    import synth_files

    synth_files.write_inputs(str(tmp_path), chroms=("1", "2", "3"), snps_per_chr=4000, seed=3, gz=True)
    threaded = _counts(tmp_path, 8, single_feeder=True)
    assert not _counts.redone
    assert threaded == _counts(tmp_path, 1)
    assert threaded == _counts(tmp_path, 8)  # ... and so does the engine of the batched front end with a list of one pair
    assert not _counts.redone


def test_threaded_table_fill_redoes_sequentially_when_a_sample_is_redrawn(tmp_path):
    """A mutation older than the age grid makes the reference draw again (coal.cpp:2286-2287), which breaks the fixed 100
    draws per SNP the threaded fill relies on: it must notice and repeat the fill sequentially, with the same result."""
    import gzip

    import synth_files

    synth_files.write_inputs(str(tmp_path), chroms=("1", "2"), snps_per_chr=2500, seed=5, gz=True)
    p = tmp_path / "P_chr2.mut.gz"
    lines = gzip.open(p, "rt").read().split("\n")
    n_changed = 0
    for i in range(1, len(lines)):
        f = lines[i].split(";")
        if len(f) > 10 and f[7] == "0" and f[5] == "7" and float(f[8]) > 1e4 and n_changed < 40:
            f[8], f[9] = "5e+06", "4e+07"  # most sampled ages lie beyond the last grid point (8.9e6 generations)
            lines[i] = ";".join(f)
            n_changed += 1
    assert n_changed == 40
    with gzip.open(p, "wt") as g:
        g.write("\n".join(lines))
    threaded = _counts(tmp_path, 8, single_feeder=True)
    assert _counts.redone  # (the threaded pass gave up ...)
    assert threaded == _counts(tmp_path, 1)  # (... and the repeat is the sequential result)
    assert threaded == _counts(tmp_path, 8)  # (the engine of the batched front end notices too and hands the pair to the feeder)
    assert _counts.redone


def test_malformed_mut_line_is_reported_from_the_reader_thread(tmp_path):
    """mutations.cpp:77-246 exits with `Error reading following line in mut file` on a row it cannot parse; ours does the same
    from its reader thread (exit code 1, the line echoed), threaded or not."""
    import gzip

    import synth_files

    synth_files.write_inputs(str(tmp_path), chroms=("1", "2"), snps_per_chr=300, seed=9, gz=True)
    p = tmp_path / "P_chr2.mut.gz"
    lines = gzip.open(p, "rt").read().split("\n")
    lines[50] = lines[50].replace(";", ";x", 2)  # the position field is no number any more
    with gzip.open(p, "wt") as g:
        g.write("\n".join(lines))
    for threads in ("8", "1"):
        r = subprocess.run([CLI, "--mode", "mut", "--mut", "P", "--target_tmp", "T.colate.in", "--reference_tmp", "R.colate.in", "--chr",
                            "chr.txt", "--bins", "3,7,0.2", "--seed", "1", "--counts_only", "-o", "x"], cwd=str(tmp_path),
                           capture_output=True, text=True, env=dict(os.environ, COLATE_THREADS=threads), timeout=60)
        assert r.returncode == 1 and "Error reading following line in mut file" in r.stderr and lines[50] in r.stderr, r.stderr[-400:]
