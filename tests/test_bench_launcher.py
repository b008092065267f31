"""bench.py's multi-rank entry (`python bench.py --gpus N` spawns N rank processes itself).

CPU: a dry run (COLATE_BENCH_DRY=1 -- no kernel launch, everything else: spawning, rendezvous on 127.0.0.1, replicate
sharding, the packed all-gather through colate_amd.distributed over gloo, the JSON contract).  GPU box (one GPU):
the same entry with the real kernel and two ranks sharing the device over gloo, against the one-rank line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **env):
    e = dict(os.environ, **env)
    e.pop("WORLD_SIZE", None), e.pop("RANK", None), e.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE JSON line, the other ranks nothing
    return json.loads(lines[0])


@pytest.mark.parametrize("extra,total,rank0", [([], 10, 5), (["--total-replicates", "7"], 7, 4)])
def test_dry_run_two_ranks_gloo(extra, total, rank0):
    d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--replicates", "5"] + extra, COLATE_BENCH_DRY="1")
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["value"] is None and d["data"].startswith("DRY RUN")
    assert d["scaling"] == ("strong" if extra else "weak")
    assert d["config"]["replicates_total"] == total and d["config"]["replicates_rank0"] == rank0
    assert d["config"]["em_iterations_mean"] == 1001.0  # every rank's (fake) results arrived through the all-gather
    for key in ("metric", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "roofline"):
        assert key in d


def test_dry_run_pairs_two_ranks_gloo():
    """`--pairs P --replicates B` (BASELINE configs[4]): P x B rows sharded over the ranks, each rank builds only the pairs its rows
    belong to (3 pairs x 5 replicates over 2 ranks cut inside pair 1), one all-gather."""
    d = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--pairs", "3", "--replicates", "5"], COLATE_BENCH_DRY="1")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["pairs"] == 3 and d["config"]["replicates_per_pair"] == 5
    assert d["config"]["replicates_total"] == 15 and d["config"]["replicates_rank0"] == 8 and d["config"]["em_iterations_mean"] == 1001.0


def test_dry_run_one_rank_needs_no_launcher():
    d = _run(["--steps", "1", "--warmup", "0", "--replicates", "3"], COLATE_BENCH_DRY="1")
    assert d["n_gpus"] == 1 and d["config"]["replicates_total"] == 3


def test_refuses_more_ranks_than_gpus_unless_rehearsing():
    """Without a GPU (here) or with fewer GPUs than ranks the nccl path must fail loudly, not fall back."""
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None), e.pop("COLATE_BENCH_DRY", None), e.pop("COLATE_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, env=e, timeout=600)
    import torch

    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "{" not in r.stdout


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_over_gloo_and_match_one_rank():
    """Real kernel, two rank processes on this box's GPU, gather over gloo (COLATE_BENCH_BACKEND=gloo): the JSON
    says n_gpus 2, and 2 x 12 replicates arrive (weak scaling); strong scaling shards one table of 13."""
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-host-path", "--no-cxx-rccl-check"]
    one = _run(common + ["--replicates", "12"])
    assert one["n_gpus"] == 1 and one["value"] > 0 and one["config"]["status_flags_nonzero"] == 0
    two = _run(common + ["--gpus", "2", "--replicates", "12"], COLATE_BENCH_BACKEND="gloo")
    assert two["n_gpus"] == 2 and two["config"]["replicates_total"] == 24 and two["value"] > 0
    assert two["config"]["status_flags_nonzero"] == 0 and two["scaling"] == "weak"
    strong = _run(common + ["--gpus", "2", "--total-replicates", "13"], COLATE_BENCH_BACKEND="gloo")
    assert strong["n_gpus"] == 2 and strong["scaling"] == "strong"
    assert strong["config"]["replicates_total"] == 13 and strong["config"]["replicates_rank0"] == 7
    assert strong["config"]["em_iterations_mean"] >= 1001


@pytest.mark.gpu
def test_pairs_shape_and_other_configs_on_the_gpu():
    """The pairs workload as the headline (bootstrap kernel + EM kernel per pass, per-row epochs) with its `other_configs`, and two
    ranks sharing the GPU over gloo: the rows arrive from both ranks."""
    common = ["--steps", "1", "--warmup", "1", "--passes-per-step", "2", "--no-cpu-baseline", "--no-host-path", "--no-cxx-rccl-check"]
    one = _run(common + ["--pairs", "6", "--replicates", "4"])
    assert one["value"] > 0 and one["config"]["replicates_total"] == 24 and one["config"]["status_flags_nonzero"] == 0
    assert one["roofline"]["bootstrap_kernel"]["kernel_ms"] > 0 and one["roofline"]["kernel_ms"] > 0
    names = [o["config"] for o in one["other_configs"]]
    assert len(names) == 4 and all(o["value"] > 0 and o["status_flags_nonzero"] == 0 for o in one["other_configs"]), names
    two = _run(common + ["--gpus", "2", "--pairs", "6", "--replicates", "4", "--no-other-configs"], COLATE_BENCH_BACKEND="gloo")
    assert two["n_gpus"] == 2 and two["config"]["replicates_total"] == 24 and two["config"]["em_iterations_mean"] >= 1001
    assert two["config"]["status_flags_nonzero"] == 0


@pytest.mark.gpu
def test_cxx_rccl_path_one_rank():
    """`Colate --ranks 1` = the C++ multi-process path (fork, communicator id through the pipe, ncclCommInitRank,
    ncclAllGather) with a communicator of one rank, against the plain run; bench.py records it as `cxx_rccl`."""
    d = _run(["--steps", "1", "--warmup", "0", "--replicates", "6", "--no-cpu-baseline", "--no-host-path"])
    assert d["cxx_rccl"]["ok"], d["cxx_rccl"]
