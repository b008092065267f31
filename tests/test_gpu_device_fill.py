"""The age sampling of the table fill on the GPU (colate_amd/csrc/fill_device.h, fill_kernel.hip; reference: coal.cpp:2260-2295)
against the same sampling on the host (csrc/mut_pairs.cpp, Engine::sample, itself pinned by the reference's fixtures in
tests/test_host_driver.py): the count tables must be the same bytes.  Through the C-ABI library's CLI entry point, as a user runs it."""
import os
import re
import subprocess

import pytest

import golden_lib as gl
import synth_files

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "colate_amd", "bin", "Colate")


def _run(args, cwd, **env):
    e = dict(os.environ, COLATE_TIMING="1")
    e.update(env)
    return subprocess.run([CLI] + args, cwd=cwd, capture_output=True, env=e)


def _both_ways(args, cwd, outputs, **env):
    """Runs `args` with the sampling on the host, then on the device; returns (host files, device files, device stderr)."""
    r = _run(args, cwd, COLATE_DEVICE_FILL="0", **env)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    assert "age sampling on the host (COLATE_DEVICE_FILL=0)" in r.stderr.decode()
    host = {}
    for o in outputs:
        host[o] = open(os.path.join(cwd, o), "rb").read()
        os.remove(os.path.join(cwd, o))
    r = _run(args, cwd, **env)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    err = r.stderr.decode()
    m = re.search(r"age sampling on the GPU: (\d+) \(pair, block\) jobs, (\d+) SNPs in (\d+) launches", err)
    assert m, err[-1500:]  # (the device path must be the one that ran: no silent fall-back)
    assert int(m.group(1)) > 0 and int(m.group(2)) > 0
    assert "0 pair(s) redone sequentially" in err, err[-1500:]
    dev = {o: open(os.path.join(cwd, o), "rb").read() for o in outputs}
    return host, dev, err


def test_pairs_fixture_tables_are_the_hosts(tmp_path):
    """The reference-made pairs fixture (modern pairs, a 500- and a 7000-year-old target; rows with age_begin = 0: the F path),
    one stream window and twelve (blocks that straddle hand-overs to the device)."""
    meta = gl.l3_pairs_stage(str(tmp_path))
    common = ["--mode", "mut", "--mut", "P"] + meta["common_args"]
    outs = [p["output"] + ".counts" for p in meta["pairs"]]
    for window in ("64", "4"):
        host, dev, err = _both_ways(common + ["--pairs", "pairs.txt", "--counts_only"], str(tmp_path), outs, COLATE_UNIFORM_WINDOW_MB=window)
        for o in outs:
            assert host[o] == dev[o], (o, window)


@pytest.mark.parametrize("name", ["l3_modern", "l3_ancient", "l3_nochr"])
def test_single_pair_cli_tables_are_the_hosts(name, tmp_path):
    """`Colate --mode mut` on one pair goes through the same engine (and so through the device)."""
    case = gl.l3_stage(name, str(tmp_path))
    args = [a for a in case["args"]]
    args[args.index("-o") + 1] = "mine"
    host, dev, err = _both_ways(args + ["--counts_out", "mine.counts", "--counts_only"], str(tmp_path), ["mine.counts"])
    assert host["mine.counts"] == dev["mine.counts"]


def test_many_blocks_small_batches(tmp_path):
    """Synthetic inputs that stress the device path: five chromosomes of four 30-Mb blocks, 8 % of the rows with age_begin = 0 (the F
    path; ranges from the first age bin over more than 64 bins), three targets x two references, 4-MB stream windows and batches of
    20 000 records (several submissions per hand-over)."""
    d = str(tmp_path)
    synth_files.write_inputs(d, chroms=("1", "2", "3", "4", "5"), snps_per_chr=20000, seed=11, span=110_000_000, extra_targets=2, extra_refs=1)
    pairs = [(f"{t}.colate.in", f"{r}.colate.in", f"out_{t}_{r}") for t in ("T", "T1", "T2") for r in ("R", "R1")]
    open(os.path.join(d, "pairs.txt"), "w").write("".join(" ".join(p) + "\n" for p in pairs))
    args = ["--mode", "mut", "--mut", "P", "--chr", "chr.txt", "--bins", "3,7,0.2", "--seed", "3", "--num_bootstraps", "3", "--pairs", "pairs.txt",
            "--counts_only"]
    outs = [p[2] + ".counts" for p in pairs]
    host, dev, err = _both_ways(args, d, outs, COLATE_UNIFORM_WINDOW_MB="4", COLATE_DEVICE_FILL_BATCH="20000")
    for o in outs:
        assert host[o] == dev[o], o
    m = re.search(r"age sampling on the GPU: (\d+) \(pair, block\) jobs", err)
    assert int(m.group(1)) >= 6 * 5 * 3  # (every pair, every chromosome, most blocks)
