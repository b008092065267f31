#!/usr/bin/env python3
"""bench.py -- bootstrap replicates / second to EM convergence on MI355X.

A "step" is one pass of the hot path over one batch: B bootstrap replicates (count tables
already resident in HBM) run to the reference's stop rule (coal.cpp:3822) by ONE launch of the
EM kernel through the C ABI (colate_em_batch_device), followed -- when N > 1 -- by the single
RCCL all-gather of the rates.  Workload = BASELINE.json configs[1]: whole-genome-like counts,
num_bootstrap = 100 per GPU, --bins 3,7,0.2 (23 epochs), 185 age bins (synthetic: the SGDP /
LBK / Loschbour files are not available offline; colate_amd/workloads.py).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` prices the EM kernel against HBM as SURVEY.md §8(d)
prescribes (algorithmic bytes per replicate-iteration = 2*A*8 counts + A*8 grid + 3*E*8
epochs/rates); `cpu_baseline` times the reference itself (oracle/_ref/Colate_ref, built from
/root/reference by oracle/Makefile) -- or, if that binary did not travel, our C restatement
(oracle/liboracle.so) -- on one host core over a bounded sample of the same replicates.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BINS = "3,7,0.2"
B_PER_GPU = 100
CPU_SAMPLE = 64  # replicates timed on the host (about 10 s on one core)


def cpu_baseline(grid, csh, cns, epochs, gpu_rates, gpu_iters, bins=BINS):
    """Time the CPU path on `CPU_SAMPLE` of the benchmark's replicates, one core."""
    S = min(CPU_SAMPLE, csh.shape[0])
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "Colate_ref")
    E = epochs.size
    if os.path.exists(ref_bin) and os.access(ref_bin, os.X_OK):
        with tempfile.TemporaryDirectory() as d:
            # the reference's own hook for precomputed count tables (coal.cpp:3169-3170, 3471-3499)
            with open(os.path.join(d, "OUT.colate_mat"), "w") as f:
                f.write(" ".join("%.17g" % x for x in grid) + "\n")
                for b in range(S):
                    f.write(" ".join("%.17g" % x for x in csh[b]) + "\n")
                    f.write(" ".join("%.17g" % x for x in cns[b]) + "\n")
            cmd = [ref_bin, "--mode", "mut", "--mut", "dummy", "--bins", bins, "--num_bootstraps", str(S), "-o", "OUT"]
            t0 = time.perf_counter()
            r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
            dt = time.perf_counter() - t0
            if r.returncode == 0 and os.path.exists(os.path.join(d, "OUT.coal")):
                lines = open(os.path.join(d, "OUT.coal")).read().split("\n")
                same = all(
                    lines[2 + b] == "0 %d " % b + " ".join("%g" % x for x in gpu_rates[b]) + " " for b in range(S)
                )
                return {
                    "value": S / dt, "unit": "replicates/s", "cores": 1, "kind": "reference",
                    "sample": f"{S} of the benchmark's replicates through the reference binary (.colate_mat hook), "
                              f"{dt:.1f} s wall incl. its start-up",
                    "coal_text_identical_to_gpu": bool(same),
                }
    # fall back to our C restatement of the reference
    import ctypes

    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], capture_output=True)
    O = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    O.oracle_em_batch.argtypes = [ctypes.c_int] * 3 + [dp] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                                                  ctypes.c_double, dp, ip, dp, ip]
    P = lambda a: a.ctypes.data_as(dp)  # noqa: E731
    sh, ns = np.ascontiguousarray(csh[:S]), np.ascontiguousarray(cns[:S])
    init = np.full(E, 1.0 / 20000.0)
    rates, iters = np.zeros((S, E)), np.zeros(S, dtype=np.int32)
    ll, fl = np.zeros(S), np.zeros(S, dtype=np.int32)
    t0 = time.perf_counter()
    O.oracle_em_batch(S, E, grid.size, P(grid), P(sh), P(ns), P(epochs), P(init), 100000, 1000, 1e-7, 5e-9,
                      P(rates), iters.ctypes.data_as(ip), P(ll), fl.ctypes.data_as(ip))
    dt = time.perf_counter() - t0
    rel = np.abs(rates - gpu_rates[:S]) / np.maximum(np.abs(rates), 1e-300)
    return {
        "value": S / dt, "unit": "replicates/s", "cores": 1, "kind": "port",
        "sample": f"{S} of the benchmark's replicates through oracle/liboracle.so, {dt:.1f} s",
        "max_rel_diff_vs_gpu": float(rel.max()), "iters_equal": bool((iters == gpu_iters[:S]).all()),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--replicates", type=int, default=B_PER_GPU, help="bootstrap replicates per GPU")
    ap.add_argument("--bins", default=BINS, help="epoch grid (default: the BASELINE config; 2,7.95,0.05 = 122 epochs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive (host-pointer ABI) timing")
    args = ap.parse_args()

    import torch

    import colate_amd
    from colate_amd import workloads

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        # COLATE_BENCH_BACKEND=gloo: rehearsal of the multi-rank control flow on a box with fewer GPUs than
        # ranks (ranks share devices, the gather is staged through host memory); never the measured setup
        backend = os.environ.get("COLATE_BENCH_BACKEND", "nccl")
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    B = args.replicates
    grid = colate_amd.age_grid()
    bins = args.bins
    epochs, _ = colate_amd.epochs_from_bins(bins)
    E, A = epochs.size, grid.size
    # every rank bootstraps its own B replicates of the same genome (weak scaling)
    csh, cns = workloads.bootstrap_tables(grid, B, nb=115, scale=11.0, seed=12345 + 1000 * rank)

    f64 = dict(dtype=torch.float64, device=dev)
    d_grid = torch.tensor(grid, **f64)
    d_sh = torch.tensor(csh, **f64)
    d_ns = torch.tensor(cns, **f64)
    d_ep = torch.tensor(epochs, **f64)
    d_init = torch.full((E,), colate_amd.DEFAULT_INIT_RATE, **f64)
    d_rates = torch.empty((B, E), **f64)
    d_iters = torch.empty((B,), dtype=torch.int32, device=dev)
    d_ll = torch.empty((B,), **f64)
    d_flags = torch.empty((B,), dtype=torch.int32, device=dev)
    d_all = torch.empty((world * B, E), **f64) if world > 1 else None

    def gather():  # the one collective: B*E doubles per rank
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(d_all, d_rates)
        else:
            h = d_rates.cpu()
            h_all = torch.empty((world * B, E), dtype=torch.float64)
            dist.all_gather_into_tensor(h_all, h)
            d_all.copy_(h_all)

    def step():
        colate_amd.em_batch_device(d_grid, d_sh, d_ns, d_ep, d_init, d_rates, d_iters, d_ll, d_flags)
        if world > 1:
            gather()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()  # on the stream the kernel is launched on (torch's current stream)
        colate_amd.em_batch_device(d_grid, d_sh, d_ns, d_ep, d_init, d_rates, d_iters, d_ll, d_flags)
        ev[k][1].record()
        if world > 1:
            gather()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], **f64) if dist.get_backend() == "nccl" else torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    iters = d_iters.cpu().numpy()
    flags = d_flags.cpu().numpy()
    rates = d_rates.cpu().numpy()

    if rank == 0:
        esteps = int((iters.astype(np.int64) + 1).sum())  # E-steps executed per launch
        traffic = None  # HBM bytes per launch from the PMC passes committed under profiles/ (same workload only)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tj["workload"] == {"replicates": B, "epochs": int(E), "age_bins": int(A)}:
                traffic = tj["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        bytes_per_rep_iter = 2 * A * 8 + A * 8 + 3 * E * 8  # 4992 B at E=23
        # the instantiation colate_em_launch picks for this shape (the name rocprofv3 reports, profiles/)
        nch = 1 if E <= 64 else (2 if E <= 128 else 4)
        rows = (1 if E <= 16 else (2 if E <= 32 else 4)) if nch == 1 else 4
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        variant = os.environ.get("COLATE_EM_VARIANT")
        tput = nch <= 2 and (variant == "throughput" or (variant != "latency" and B > 2 * cus))
        kernel_name = f"em_kernel<0, {nch}, {rows}, {'true' if tput else 'false'}>"
        achieved = esteps * bytes_per_rep_iter / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "bootstrap replicates/sec to EM convergence, whole-genome SGDP mut, 20 epochs",
            "value": world * B * args.steps / elapsed,
            "unit": "replicates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"whole-genome-like LBK-vs-Loschbour-shaped count tables (nb=115 blocks), "
                            f"num_bootstrap={B} per GPU, --bins {bins} (E={E} epochs), A={A} age bins, "
                            f"EM to the reference stop rule (min 1001 iterations)",
                "replicates_per_gpu": B, "epochs": E, "age_bins": A,
                "parallelism": f"replicates sharded over {world} GPU(s), one RCCL all-gather of rates per step",
                "em_iterations_mean": float(iters.mean()), "flags_nonzero": int((flags != 0).sum()),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": kernel_name,
                "kernel_ms": kern_ms,
                "algorithmic_bytes_per_launch": esteps * bytes_per_rep_iter,
                "note": "algorithmic bytes = (2*A*8 + A*8 + 3*E*8) B per replicate-iteration x E-steps per launch "
                        "(SURVEY.md §8d); the kernel keeps all of it on chip, compulsory HBM traffic is "
                        "(2*A + E)*8 B per replicate, so the real limiter is FP64 VALU latency (DESIGN.md §5)",
            },
        }
        if world == 1 and not args.no_host_path:
            # PCIe-inclusive rate through the host-pointer entry point (hipMalloc + copies + launch + copies
            # back inside the call); reported beside `value`, never as it
            colate_amd.em_batch(grid, csh, cns, epochs)
            t1 = time.perf_counter()
            for _ in range(5):
                colate_amd.em_batch(grid, csh, cns, epochs)
            out["host_path"] = {"value": 5 * B / (time.perf_counter() - t1), "unit": "replicates/s",
                                "note": "colate_em_batch with host buffers, PCIe and allocation inclusive"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(grid, csh, cns, epochs, rates, iters, bins)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
