"""GPU: the multi-device forms of the replicate sharding.  The tests that need DISTINCT devices run only where at
least two GPUs are visible (the driver's 8-GPU node; a one-GPU box skips them, visibly), so that the first
multi-GPU box exercises `--devices 2`, `--ranks 2` (RCCL all-gather between two processes) and
colate_em_batch_sharded on different devices; the failure path of the ranked launcher runs on any box."""
import os
import subprocess

import numpy as np
import pytest

import golden_lib as gl
import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "colate_amd", "bin", "Colate")
HOOKS_LIB_DIR = os.path.join(ROOT, "colate_amd", "lib", "testhooks")  # the library built with -DCOLATE_TEST_HOOKS (failure injection)


@pytest.fixture(scope="module")
def ca():
    import colate_amd

    assert colate_amd.device_count() >= 1
    return colate_amd


def _two(ca):
    if ca.device_count() < 2:
        pytest.skip("needs two visible GPUs (runs on the multi-GPU node)")


def _stage(tmp_path):
    case = gl.l3_stage("l3_modern", str(tmp_path))
    args = list(case["args"])
    args[args.index("-o") + 1] = "mine"
    return case, args


def _run(args, cwd, **env):
    return subprocess.run([CLI] + args, cwd=str(cwd), capture_output=True, text=True, timeout=600,
                          env=dict(os.environ, **env))


def test_sharded_entry_point_on_two_devices(ca):
    _two(ca)
    from colate_amd import workloads

    grid = ol.age_grid()
    ep, _ = ol.epochs_from_bins("3,7,0.2")
    csh, cns = workloads.bootstrap_tables(grid, 11, nb=9, scale=1.0)
    r1, it1, ll1, fl1 = ca.em_batch(grid, csh, cns, ep)
    for devs in ([0, 1], [1, 0, 1]):
        r, it, ll, fl = ca.em_batch_sharded(devs, grid, csh, cns, ep)
        assert np.array_equal(r, r1) and np.array_equal(ll, ll1) and (it == it1).all() and (fl == fl1).all()


@pytest.mark.parametrize("how", ["--devices", "--ranks"])
def test_cli_two_gpus_equal_one(ca, how, tmp_path):
    """`Colate --devices 2` (one process, two GPUs) and `Colate --ranks 2` (two processes, one RCCL all-gather):
    byte-identical .coal and the same iteration lines as the one-GPU run."""
    _two(ca)
    case, args = _stage(tmp_path)
    one = _run(args, tmp_path)
    assert one.returncode == 0, one.stderr[-800:]
    ref_text = (tmp_path / "mine.coal").read_text()
    os.remove(tmp_path / "mine.coal")
    two = _run(args + [how, "2"], tmp_path)
    assert two.returncode == 0, two.stderr[-800:]
    assert (tmp_path / "mine.coal").read_text() == ref_text
    pick = lambda err: [l for l in err.split("\n") if l.startswith("Bootstrap ")]
    assert pick(two.stderr) == pick(one.stderr) and len(pick(one.stderr)) == len(case["iterations"])


@pytest.mark.parametrize("nranks", [1, 2, 3])
def test_pairs_with_ranks_equal_one_process(ca, nranks, tmp_path):
    """`Colate --pairs FILE --ranks N` (BASELINE configs[4] on N GPUs): the (pair, replicate) rows of every launch are
    sharded over N processes, each rank reads and fills only the pairs its rows belong to, one RCCL all-gather per launch;
    the .coal files and the iteration lines are those of the one-process run.  N = 1 runs on any box (the same code with a
    communicator of one); N = 2, 3 need as many visible GPUs (3 ranks over 6 pairs x 3 replicates cut inside pairs)."""
    if nranks > 1 and ca.device_count() < nranks:
        pytest.skip(f"needs {nranks} visible GPUs (runs on the multi-GPU node)")
    meta = gl.l3_pairs_stage(str(tmp_path))
    common = ["--mode", "mut", "--mut", "P"] + meta["common_args"] + ["--pairs", "pairs.txt"]
    one = _run(common, tmp_path)
    assert one.returncode == 0, one.stderr[-800:]
    outs = [p["output"] for p in meta["pairs"]]
    texts = {o: (tmp_path / (o + ".coal")).read_text() for o in outs}
    for o in outs:
        os.remove(tmp_path / (o + ".coal"))
    many = _run(common + ["--ranks", str(nranks)], tmp_path)
    assert many.returncode == 0, many.stderr[-800:]
    for o in outs:
        assert (tmp_path / (o + ".coal")).read_text() == texts[o], o
    pick = lambda err: [l for l in err.split("\n") if " Bootstrap " in l]
    assert pick(many.stderr) == pick(one.stderr) and len(pick(one.stderr)) == sum(len(p["iterations"]) for p in meta["pairs"])


def test_two_ranks_one_fails_nobody_hangs(ca, tmp_path):
    """Rank 1's local work fails (injected): it still joins the all-gather with its code, rank 0 learns of it from the
    gathered codes, both exit non-zero and the launcher returns promptly."""
    _two(ca)
    _, args = _stage(tmp_path)
    r = _run(args + ["--ranks", "2"], tmp_path, LD_LIBRARY_PATH=HOOKS_LIB_DIR, COLATE_TEST_FAIL_RANK="1", COLATE_RANK_GRACE_SEC="20")
    assert r.returncode != 0 and "rank 1" in r.stderr and not (tmp_path / "mine.coal").exists(), r.stderr[-800:]
    assert "ended by the launcher" not in r.stderr  # nobody had to be killed: the failing rank took part in the collective


def test_rank0_hangs_before_the_id_launcher_still_returns(ca, tmp_path):
    """Rank 0 hangs before it has published the communicator id while rank 1 waits for the id on its pipe: nobody exits,
    so the launcher's bound on the wait for the id (COLATE_RANK_ID_TIMEOUT_SEC) must end both ranks."""
    import time

    _two(ca)
    _, args = _stage(tmp_path)
    t0 = time.time()
    r = _run(args + ["--ranks", "2"], tmp_path, LD_LIBRARY_PATH=HOOKS_LIB_DIR, COLATE_TEST_HANG_RANK="0",
             COLATE_RANK_ID_TIMEOUT_SEC="20", COLATE_RANK_GRACE_SEC="60")
    assert r.returncode != 0 and "has not published the communicator id" in r.stderr, r.stderr[-800:]
    assert r.stderr.count("ended by the launcher") == 2 and time.time() - t0 < 120 and not (tmp_path / "mine.coal").exists()


def test_one_rank_injected_failure_is_reported(ca, tmp_path):
    """The same path with a communicator of one (any box): the injected failure comes back as this rank's error."""
    _, args = _stage(tmp_path)
    r = _run(args + ["--ranks", "1"], tmp_path, LD_LIBRARY_PATH=HOOKS_LIB_DIR, COLATE_TEST_FAIL_RANK="0")
    assert r.returncode != 0 and "injected failure on rank 0" in r.stderr, r.stderr[-800:]
    assert not (tmp_path / "mine.coal").exists()


def test_worker_threads_release_their_workspace(ca):
    """A host that calls colate_em_batch from short-lived threads must not leak one device buffer, one pinned buffer and
    one stream per thread (the thread_local workspace is freed at thread exit)."""
    import threading

    import torch
    from colate_amd import workloads

    grid = ol.age_grid()
    ep, _ = ol.epochs_from_bins("3,7,0.2")
    csh, cns = workloads.bootstrap_tables(grid, 64, nb=9, scale=1.0)
    big = np.repeat(csh, 64, axis=0), np.repeat(cns, 64, axis=0)  # 4096 x 185 x 2 doubles = 12 MB per workspace

    def work():
        ca.em_batch(grid, big[0], big[1], ep, max_iter=2, min_iter=0)

    def free_bytes():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info(0)[0]

    t = threading.Thread(target=work)
    t.start(), t.join()
    base = free_bytes()
    for _ in range(12):
        t = threading.Thread(target=work)
        t.start(), t.join()
    assert base - free_bytes() < 40 << 20, (base, free_bytes())  # 12 leaked workspaces would be > 140 MB


def test_two_ranks_on_one_gpu_fail_cleanly(ca, tmp_path):
    """On a one-GPU box `--ranks 2` puts both ranks on device 0, which RCCL refuses (ncclCommInitRank: invalid usage):
    both ranks must report that and the launcher must return non-zero promptly -- no hang, no .coal."""
    import time

    if ca.device_count() != 1:
        pytest.skip("one-GPU box only (with two GPUs the run succeeds: test_cli_two_gpus_equal_one)")
    _, args = _stage(tmp_path)
    t0 = time.time()
    r = _run(args + ["--ranks", "2"], tmp_path, COLATE_RANK_GRACE_SEC="5")
    assert r.returncode != 0 and time.time() - t0 < 120, (r.returncode, r.stderr[-500:])
    assert "rank" in r.stderr and not (tmp_path / "mine.coal").exists()
