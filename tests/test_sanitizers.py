"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU builds only: `make -C colate_amd/csrc asan`,
`make -C oracle asan`).  The host side of the library -- option parsing, .mut / .colate.in / fasta readers, table fill,
block bootstrap, epoch builders, .coal/.colate_mat/--counts_out writers, make_tmp -- runs over the committed L3/L4
fixtures with device entry points that fail (tools/no_device_stubs.cpp); the oracle runs a capped EM on the tables
that come out.  A sanitizer report makes the binary exit non-zero (-fno-sanitize-recover, ASan's default abort)."""
import gzip
import json
import os
import shutil
import subprocess

import pytest

import golden_lib as gl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_CLI = os.path.join(ROOT, "colate_amd", "bin", "Colate_asan")
TSAN_CLI = os.path.join(ROOT, "colate_amd", "bin", "Colate_tsan")
ASAN_ORACLE = os.path.join(ROOT, "oracle", "oracle_asan")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=97", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


@pytest.fixture(scope="module", autouse=True)
def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "colate_amd", "csrc"), "asan", "tsan"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)


def _run(exe, args, cwd, **env):
    r = subprocess.run([exe] + args, cwd=cwd, capture_output=True, env=dict(ENV, **env))
    err = r.stderr.decode()
    assert "ERROR: AddressSanitizer" not in err and "runtime error:" not in err and "LeakSanitizer" not in err, err[-3000:]
    assert "ThreadSanitizer" not in err, err[-3000:]
    return r, err


@pytest.mark.parametrize("name", gl.l3_names())
def test_host_driver_and_oracle_clean_on_l3_fixtures(name, tmp_path):
    case = gl.l3_stage(name, str(tmp_path))
    args = list(case["args"])
    args[args.index("-o") + 1] = "mine"
    B = args[args.index("--num_bootstraps") + 1]
    r, err = _run(ASAN_CLI, args + ["--counts_out", "mine.counts", "--counts_only"], str(tmp_path))
    assert r.returncode == 0, err[-1500:]
    assert f"Number of blocks: {case['num_blocks']}" in err
    # the same tables as the regular build writes
    cli = os.path.join(ROOT, "colate_amd", "bin", "Colate")
    args2 = list(args)
    args2[args2.index("-o") + 1] = "plain"
    subprocess.check_call([cli] + args2 + ["--counts_out", "plain.counts", "--counts_only"], cwd=str(tmp_path),
                          stderr=subprocess.DEVNULL)
    assert (tmp_path / "mine.counts").read_text() == (tmp_path / "plain.counts").read_text()
    # without --counts_only the sanitizer binary must stop at the device boundary, loudly
    r, err = _run(ASAN_CLI, args, str(tmp_path))
    assert r.returncode == 1 and "no device code linked" in err
    # oracle: epoch builder, capped EM, E-step, RNG restatement on those tables
    age = "0"
    if "--target_age" in args:
        age = str(float(args[args.index("--target_age") + 1]) / 28.0)
    oargs = ["mine.counts", B, args[args.index("--bins") + 1] if "--bins" in args else "x", age]
    if "--coal" in args:
        oargs.append(args[args.index("--coal") + 1])
    r, err = _run(ASAN_ORACLE, oargs, str(tmp_path))
    assert r.returncode == 0 and b"oracle_asan ok" in r.stdout, (r.returncode, err[-1500:])


def test_make_tmp_and_writers_clean(tmp_path):
    gl.l3_stage("l3_masks", str(tmp_path))
    src = os.path.join(gl.HERE, "l4_maketmp")
    case = json.load(open(os.path.join(src, "case.json")))
    with gzip.open(os.path.join(src, "table.txt.gz"), "rb") as g:
        (tmp_path / "table.txt").write_bytes(g.read())
    for f in ("G_chr1.fa", "G_chr2.fa"):
        shutil.copy(os.path.join(src, f), str(tmp_path / f))
    args = list(case["args"])
    args[args.index("-o") + 1] = "mine"
    r, err = _run(ASAN_CLI, args, str(tmp_path))
    assert r.returncode == 0, err[-1500:]
    with gzip.open(os.path.join(src, "expected.colate.in.gz"), "rb") as g:
        assert (tmp_path / "mine.colate.in").read_bytes() == g.read()
    # --write_colate_mat and the .colate_mat loader
    wcase = json.load(open(os.path.join(gl.HERE, "l4_colate_mat", "case.json")))
    d2 = tmp_path / "w"
    gl.l3_stage("l3_modern", str(d2))
    r, err = _run(ASAN_CLI, wcase["writer_args"], str(d2))
    assert r.returncode == 0, err[-1500:]
    r, err = _run(ASAN_CLI, wcase["reader_args"] + ["--counts_out", "back.counts", "--counts_only"], str(d2))
    assert r.returncode == 0 and "Loading precomputed file" in err, err[-1500:]
    # malformed inputs end in error messages, not in sanitizer reports
    (tmp_path / "bad.mut").write_text("header\n1;2;3\n")
    r, err = _run(ASAN_CLI, ["--mode", "mut", "--mut", "bad.mut", "--target_tmp", "T.colate.in", "--reference_tmp", "R.colate.in",
                             "--bins", "3,7,0.2", "-o", "z", "--counts_only"], str(tmp_path))
    assert r.returncode != 0
    (tmp_path / "trunc.colate.in").write_bytes(open(tmp_path / "T.colate.in", "rb").read()[:1001])
    r, err = _run(ASAN_CLI, ["--mode", "mut", "--mut", "P", "--chr", "chr.txt", "--target_tmp", "trunc.colate.in", "--reference_tmp",
                             "R.colate.in", "--bins", "3,7,0.2", "--seed", "1", "-o", "z", "--counts_out", "z.counts", "--counts_only"],
                  str(tmp_path))
    assert r.returncode in (0, 1)


def test_pairs_front_end_clean_and_equal_to_single_runs(tmp_path):
    """The batched all-pairs front end (mut_pairs.cpp: thread pool, shared uniform stream in windows, mapped .colate.in
    files, per-block sampling jobs) under ASan/UBSan and under ThreadSanitizer, on inputs large enough for several stream
    windows (COLATE_UNIFORM_WINDOW_MB=4: about 5 000 used SNPs per window), with 1, 3 and 8 workers: no report, and every
    pair's tables equal to those of the pair run alone through the single-pair feeder of the regular build."""
    import synth_files

    synth_files.write_inputs(str(tmp_path), chroms=("1", "2"), snps_per_chr=16000, seed=5, gz=True, extra_targets=1, extra_refs=1)
    specs = [("T.colate.in", "R.colate.in", "p0", "0", "0"), ("T1.colate.in", "R.colate.in", "p1", "7000", "0"),
             ("T.colate.in", "R1.colate.in", "p2", "0", "0"), ("T1.colate.in", "R1.colate.in", "p3", "0", "0"),
             ("T1.colate.in", "missing.colate.in", "p4", "0", "0")]
    common = ["--mode", "mut", "--mut", "P", "--chr", "chr.txt", "--bins", "3,7,0.2", "--seed", "3", "--num_bootstraps", "2"]
    cli = os.path.join(ROOT, "colate_amd", "bin", "Colate")
    (tmp_path / "pairs.txt").write_text("".join(" ".join(sp) + "\n" for sp in specs[:4]))
    expected = {}
    for tgt, ref, out, ta, ra in specs[:4]:
        subprocess.check_call([cli] + common + ["--target_tmp", tgt, "--reference_tmp", ref, "--target_age", ta, "--reference_age", ra,
                                                 "-o", out + "_single", "--counts_out", out + "_single.counts", "--counts_only"],
                              cwd=str(tmp_path), stderr=subprocess.DEVNULL, env=dict(os.environ, COLATE_SINGLE_FEEDER="1"))
        expected[out] = (tmp_path / (out + "_single.counts")).read_text()
    for exe, threads in ((ASAN_CLI, "3"), (TSAN_CLI, "1"), (TSAN_CLI, "3"), (TSAN_CLI, "8")):
        r, err = _run(exe, common + ["--pairs", "pairs.txt", "--counts_only"], str(tmp_path), COLATE_THREADS=threads,
                      COLATE_UNIFORM_WINDOW_MB="4", COLATE_TIMING="1")
        assert r.returncode == 0, err[-1500:]
        windows = int(err.split(" stream window(s)")[0].rsplit(" ", 1)[1])
        assert windows >= 2, err[-800:]
        for out, text in expected.items():
            assert (tmp_path / (out + ".counts")).read_text() == text, (exe, threads, out)
            os.remove(tmp_path / (out + ".counts"))
    # a pair whose reference file does not exist: the reference goes on with an unreadable stream and finds no SNP; here the
    # run ends with a message (no genome blocks), not with a report
    (tmp_path / "pairs2.txt").write_text(" ".join(specs[4]) + "\n")
    r, err = _run(ASAN_CLI, common + ["--pairs", "pairs2.txt", "--counts_only"], str(tmp_path))
    assert "Failed to open missing.colate.in" in err
    # a .colate.in cut off inside a record
    (tmp_path / "trunc.colate.in").write_bytes(open(tmp_path / "T.colate.in", "rb").read()[:100001])
    (tmp_path / "pairs3.txt").write_text("trunc.colate.in R.colate.in p5\n")
    r, err = _run(ASAN_CLI, common + ["--pairs", "pairs3.txt", "--counts_only"], str(tmp_path))
    assert r.returncode == 0, err[-800:]
    subprocess.check_call([cli] + common + ["--target_tmp", "trunc.colate.in", "--reference_tmp", "R.colate.in", "-o", "p5_single",
                                             "--counts_out", "p5_single.counts", "--counts_only"], cwd=str(tmp_path), stderr=subprocess.DEVNULL,
                          env=dict(os.environ, COLATE_SINGLE_FEEDER="1"))
    assert (tmp_path / "p5.counts").read_text() == (tmp_path / "p5_single.counts").read_text()
