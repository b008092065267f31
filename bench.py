#!/usr/bin/env python3
"""bench.py -- bootstrap replicates / second to EM convergence on MI355X.

A "step" is one pass of the hot path over one batch, from "count tables resident in HBM" to "rates gathered on
the host" (SURVEY.md section 8d): ONE launch of the EM kernel through the C ABI (colate_em_batch_device) runs
this rank's bootstrap replicates to the reference's stop rule (coal.cpp:3822); when N > 1 the single RCCL
all-gather of the packed results follows (colate_amd/distributed.py: the code the gloo test covers); then one
device-to-host copy of the results and a stream synchronisation.  All of that is inside the timed region.

Workload = BASELINE.json configs[1]: whole-genome-like counts, num_bootstrap = 100 per GPU, --bins 3,7,0.2
(23 epochs), 185 age bins (synthetic: the SGDP / LBK / Loschbour files are not available offline;
colate_amd/workloads.py).  `--total-replicates 1000` is configs[2] as written: 1000 replicates sharded over
the N GPUs (strong scaling) instead of 100 per GPU (weak scaling, the default).

    python bench.py [--gpus N] [--steps K] [--warmup W]           (N > 1: spawns N rank processes itself)
    python bench.py --pairs 100 --replicates 20                   (BASELINE configs[4]: 100 pairs x 20 replicates = 2000 rows
                                                                   with per-row epochs; a pass = bootstrap kernel + EM kernel)

At N = 1 the line also carries `other_configs`: the same pass loop, about a second each, for the other single-GPU shapes of
BASELINE.json -- configs[3] (--bins 2,7.95,0.05, 122 epochs), configs[2] on one GPU (1000 replicates) and configs[4]
(100 pairs x 20 replicates) -- each with its value, kernel time and roofline.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (also fine)

Rank 0 prints ONE JSON line.  `roofline` prices the EM kernel against HBM as SURVEY.md section 8(d) prescribes
(algorithmic bytes per replicate-iteration = 2*A*8 counts + A*8 grid + 3*E*8 epochs/rates) and names what
really bounds it (`roofline.latency`); `cpu_baseline` times the reference itself (oracle/_ref/Colate_ref, built
from /root/reference by oracle/Makefile) -- or, if that binary did not travel, our C restatement
(oracle/liboracle.so) -- on one host core over a bounded sample of the same replicates.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_CLOCK_HZ = 2.4e9  # same guide: max clock
# FP64 vector peak: half the guide's FP32 vector peak (157.3 TFLOP/s = 256 CUs x 4 SIMD-32 x 2 flop x 2.4 GHz); an FP64
# wave instruction occupies its SIMD for 4 cycles (csrc/tools/ubench.hip: ~5 per independent v_fma_f64 on a lone wave)
FP64_VALU_PEAK_TFLOPS = 78.6
TIMED_REGION_S = 6.0  # the timed region lasts at least this long whatever --steps is (see `passes_per_step`): a sampler outside
                      # this process that looks every 5 s (the driver's GPU-busy probe) cannot miss it
OTHER_CONFIG_S = 1.0  # timed region of each entry of `other_configs`
BINS = "3,7,0.2"
B_PER_GPU = 100
CPU_SAMPLE = 64  # replicates timed on the host (about 10 s on one core at 23 epochs)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--replicates", type=int, default=B_PER_GPU, help="bootstrap replicates per GPU (weak scaling)")
    ap.add_argument("--total-replicates", type=int, default=0,
                    help="strong scaling: this many replicates in total, sharded over the GPUs (BASELINE configs[2]: 1000)")
    ap.add_argument("--pairs", type=int, default=0,
                    help="batched all-pairs (BASELINE configs[4]): this many (target, reference) pairs with --replicates bootstrap "
                         "replicates each, rows sharded over the GPUs; a pass = block bootstrap of all pairs + EM, one launch each")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the `other_configs` entries (N = 1)")
    ap.add_argument("--bins", default=BINS, help="epoch grid (default: the BASELINE config; 2,7.95,0.05 = 122 epochs)")
    ap.add_argument("--passes-per-step", type=int, default=0,
                    help="passes of the hot path per step (0 = chosen so that the timed region lasts >= %.1f s)" % TIMED_REGION_S)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive (host-pointer ABI) timing")
    ap.add_argument("--no-cxx-rccl-check", action="store_true",
                    help="skip the run of `Colate --ranks N` (C++ host + RCCL all-gather) after the measurement")
    return ap.parse_args()


# --------------------------------------------------------------------------------------------- launcher
def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: start N fresh rank processes (this process
    never touches a GPU) with the environment torch.distributed.run would give them; rank 0 inherits stdout."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = None if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    rc = 0
    pending = set(range(n))
    failed_at = None
    while pending:
        for r in list(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0:
                    rc = rc or code
                    failed_at = failed_at or time.time()
        if failed_at and pending and time.time() - failed_at > 15:  # a rank died: do not leave the others in a collective
            for r in pending:
                procs[r].kill()
        time.sleep(0.2)
    return rc


# --------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(grid, csh, cns, epochs, gpu_rates, gpu_iters, bins=BINS):
    """Time the CPU path on `CPU_SAMPLE` of the benchmark's replicates, one core."""
    E = epochs.size
    S = min(CPU_SAMPLE if E <= 64 else CPU_SAMPLE // 4, csh.shape[0])
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "Colate_ref")
    if os.path.exists(ref_bin) and os.access(ref_bin, os.X_OK):
        with tempfile.TemporaryDirectory() as d:
            # the reference's own hook for precomputed count tables (coal.cpp:3169-3170, 3471-3499)
            write_colate_mat(os.path.join(d, "OUT.colate_mat"), grid, csh[:S], cns[:S])
            cmd = [ref_bin, "--mode", "mut", "--mut", "dummy", "--bins", bins, "--num_bootstraps", str(S), "-o", "OUT"]
            t0 = time.perf_counter()
            r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
            dt = time.perf_counter() - t0
            if r.returncode == 0 and os.path.exists(os.path.join(d, "OUT.coal")):
                lines = open(os.path.join(d, "OUT.coal")).read().split("\n")
                last = {}  # the reference rewrites its progress line ("...iterations N\r"): keep the last count per replicate
                for l in r.stderr.replace("\r", "\n").split("\n"):
                    if l.startswith("Bootstrap "):
                        last[int(l.split()[1].rstrip(":"))] = int(l.rsplit(" ", 1)[1])
                ref_iters = [last.get(b + 1) for b in range(S)]
                diff_per_epoch = np.zeros(E, dtype=np.int64)
                for b in range(S):
                    ref_tok = lines[2 + b].split()[2:]
                    gpu_tok = ["%g" % x for x in gpu_rates[b]]
                    diff_per_epoch += np.array([a != c for a, c in zip(ref_tok, gpu_tok)], dtype=np.int64)
                # ... and replicate-parallel over the host cores this process may use: P copies
                # of the single-threaded reference binary side by side, `per` replicates each
                P = max(1, min(len(os.sched_getaffinity(0)), 32))
                per = max(4, S // 4)
                for k in range(P):
                    os.makedirs(os.path.join(d, f"p{k}"))
                    idx = [(k * per + i) % csh.shape[0] for i in range(per)]
                    write_colate_mat(os.path.join(d, f"p{k}", "OUT.colate_mat"), grid, csh[idx], cns[idx])
                cmd_p = [ref_bin, "--mode", "mut", "--mut", "dummy", "--bins", bins, "--num_bootstraps", str(per), "-o", "OUT"]
                t1 = time.perf_counter()
                procs = [subprocess.Popen(cmd_p, cwd=os.path.join(d, f"p{k}"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                         for k in range(P)]
                ok_all = all(q.wait() == 0 for q in procs)
                dt_all = time.perf_counter() - t1
                all_cores = {"value": P * per / dt_all, "unit": "replicates/s", "cores": P,
                             "sample": f"{P} concurrent copies of the reference binary x {per} replicates, {dt_all:.1f} s wall"} if ok_all else None
                return {
                    "all_cores": all_cores,
                    "value": S / dt, "unit": "replicates/s", "cores": 1, "kind": "reference",
                    "sample": f"{S} of the benchmark's replicates through the reference binary (.colate_mat hook), "
                              f"{dt:.1f} s wall incl. its start-up",
                    "coal_text_identical_to_gpu": bool(diff_per_epoch.sum() == 0),
                    "iterations_identical_to_gpu": bool(ref_iters == [int(x) for x in gpu_iters[:S]]),
                    "differing_tokens_per_epoch": {str(e): int(n) for e, n in enumerate(diff_per_epoch) if n},
                    "tokens_compared": int(S * E),
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


def roofline(pmc, hbm_algorithmic_gbs, kern_ms, kernel_name, variant, algorithmic_bytes, n_local, cus, rounds, crit_iters):
    """The bench line's `roofline` object.  SURVEY.md section 8(d)'s HBM line (algorithmic bytes / kernel time against the
    8 TB/s peak) is kept under `hbm`; the TOP-LEVEL bound is what limits the kernel: FP64 vector issue.  The kernel keeps
    its working set on chip (counter traffic is ~0.002 x the algorithmic bytes), so HBM is a label here, not a bound."""
    hbm = {
        "bound": "hbm", "achieved": hbm_algorithmic_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": hbm_algorithmic_gbs / HBM_PEAK_GBS,
        "algorithmic_bytes_per_launch": algorithmic_bytes,
        "achieved_from_counters": (pmc["hbm_bytes_per_launch"] / (kern_ms * 1e-3) / 1e9) if pmc else None,
        "note": "achieved = algorithmic bytes ((2*A*8 + A*8 + 3*E*8) B per replicate-iteration x E-steps per launch, SURVEY.md "
                "section 8d) / kernel time; achieved_from_counters = FETCH_SIZE/WRITE_SIZE bytes per launch (profiles/pmc.json) / "
                "kernel time: what the kernel really asks of HBM",
    }
    latency = {
        "what": "one workgroup per replicate runs its EM iterations back to back; an iteration is a chain of dependent FP64 "
                "instructions through the two role leaders and three LDS hand-overs, so a CU's vector units are busy only "
                "while one of its (few) waves has an instruction in flight",
        "workgroups": n_local, "cus": cus, "cus_occupied": min(n_local, cus), "workgroup_rounds": rounds,
        "em_iterations_on_critical_path": crit_iters * rounds,
        "ns_per_em_iteration": 1e6 * kern_ms / (crit_iters * rounds),
        "cycles_per_em_iteration_at_peak_clock": kern_ms * 1e-3 * PEAK_CLOCK_HZ / (crit_iters * rounds),
        "pmc": ({k: pmc[k] for k in pmc if k not in ("workload", "hbm_bytes_per_launch")} if pmc else None),
    }
    common = {"traffic": pmc["hbm_bytes_per_launch"] if pmc else None, "kernel": kernel_name, "kernel_build": variant,
              "kernel_ms": kern_ms, "hbm": hbm, "latency": latency}
    if not pmc or "sq_active_inst_valu" not in pmc or pmc.get("stale"):
        # no counter record for this shape under profiles/ -- or one taken on other kernel sources than this tree's (a kernel
        # edit without a PMC refresh must not be priced with the old counters): only the HBM line can be given
        why = ("the PMC record of this workload in profiles/pmc.json was taken on other kernel sources (kernel_source_sha16 "
               f"{pmc.get('kernel_source_sha16')}) than the library's: the FP64-VALU bound is not priced" if pmc else
               "no PMC record for this workload in profiles/pmc.json: the FP64-VALU bound is not priced; see `latency`")
        return dict(hbm, **common, valu_frac=None, note=why)
    # VALUBusy as rocprof defines it: quad-cycles with a VALU instruction executing, summed over waves, against the
    # SIMD-cycles of the whole chip during THIS run's kernel time
    busy_chip = 4.0 * pmc["sq_active_inst_valu"] / (kern_ms * 1e-3 * PEAK_CLOCK_HZ * 4 * cus)
    occupied = min(n_local, cus) / cus if rounds == 1 else 1.0
    return dict({
        "bound": "fp64-valu",
        "achieved": busy_chip * FP64_VALU_PEAK_TFLOPS, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": busy_chip,
        "valu_busy_frac_of_occupied_cus": busy_chip / occupied,
        "note": "frac = share of the chip's vector-issue cycles (256 CUs x 4 SIMDs x kernel time at the 2.4 GHz peak clock) in "
                "which a VALU instruction was executing (SQ_ACTIVE_INST_VALU of the committed PMC pass, profiles/pmc.json, over "
                "the kernel time measured live); achieved = frac x the FP64 vector peak, i.e. issue-equivalent TFLOP/s, not counted "
                "flops.  At B <= #CUs only B CUs are occupied by 4 live waves each (valu_busy_frac_of_occupied_cus) and the time "
                "is the latency of one EM iteration (`latency`); the HBM line SURVEY.md section 8(d) asks for is under `hbm`.",
    }, **common)


def emit(out):
    """ONE JSON line on stdout (NaN / inf are not JSON: they go out as null)."""
    def clean(x):
        if isinstance(x, dict):
            return {k: clean(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [clean(v) for v in x]
        if isinstance(x, float) and not np.isfinite(x):
            return None
        return x

    print(json.dumps(clean(out)), flush=True)


def write_colate_mat(path, grid, csh, cns):
    with open(path, "w") as f:
        f.write(" ".join("%.17g" % x for x in grid) + "\n")
        for b in range(csh.shape[0]):
            f.write(" ".join("%.17g" % x for x in csh[b]) + "\n")
            f.write(" ".join("%.17g" % x for x in cns[b]) + "\n")


def cxx_rccl_check(colate_amd, nranks, grid, csh, cns, epochs, bins):
    """The C++ form of the multi-GPU path on this box's GPUs: `Colate --ranks N` forks N processes (one per GPU),
    each runs its replicate range, ONE ncclAllGather (colate_comm.cpp) collects them, rank 0 writes the .coal.
    Checked against this process's own single-GPU run of the same replicates; bounded by a 120 s timeout."""
    cli = os.path.join(ROOT, "colate_amd", "bin", "Colate")
    S = min(csh.shape[0], 8 * nranks + 3)  # ragged on purpose when N > 1
    if not os.path.exists(cli):
        return {"ranks": nranks, "ok": False, "detail": "colate_amd/bin/Colate not built"}
    rates, iters, _, _ = colate_amd.em_batch(grid, csh[:S], cns[:S], epochs)
    want = ["0 %d " % b + " ".join("%g" % x for x in rates[b]) + " " for b in range(S)]
    with tempfile.TemporaryDirectory() as d:
        write_colate_mat(os.path.join(d, "OUT.colate_mat"), grid, csh[:S], cns[:S])
        cmd = ["timeout", "-k", "5", "120", cli, "--mode", "mut", "--mut", "dummy", "--bins", bins, "--num_bootstraps", str(S),
               "--seed", "1", "--ranks", str(nranks), "-o", "OUT"]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        if r.returncode != 0 or not os.path.exists(os.path.join(d, "OUT.coal")):
            return {"ranks": nranks, "ok": False, "seconds": dt, "detail": f"exit {r.returncode}: {r.stderr[-300:]}"}
        got = open(os.path.join(d, "OUT.coal")).read().split("\n")[2:2 + S]
        got_iters = [int(l.rsplit(" ", 1)[1]) for l in r.stderr.split("\n") if l.startswith("Bootstrap ")]
    return {"ranks": nranks, "replicates": S, "seconds": dt,
            "ok": bool(got == want and got_iters == [int(x) for x in iters]),
            "what": "Colate --ranks N (fork per GPU, one ncclAllGather in C++) vs this process's one-GPU run: .coal text and iteration counts"}


# --------------------------------------------------------------------------------------------- PMC records and their validity
KERNEL_SOURCES = ("em_kernel_impl.hpp", "em_math.hpp", "em_kernels.h", "em_kernels.hip", "em_kernels_ilp.hip", "em_kernels_big.hip",
                  "bootstrap_kernel.hip")


def kernel_source_sha16():
    """Fingerprint of what the kernels are built from (the sources travel with the library): a PMC record under profiles/ is
    only used for the VALU line when it was taken on exactly these sources."""
    import hashlib

    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "colate_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_record(workload):
    """The PMC entry of profiles/pmc.json for this workload, or None; `stale` is set when its fingerprint is not this tree's."""
    try:
        for entry in json.load(open(os.path.join(ROOT, "profiles", "pmc.json"))):
            if entry["workload"] == workload:
                entry = dict(entry)
                entry["stale"] = entry.get("kernel_source_sha16") != kernel_source_sha16()
                return entry
    except (OSError, KeyError, ValueError):
        pass
    return None


# --------------------------------------------------------------------------------------------- one rank's part of a workload
class Workload:
    """One rank's part of one workload, inputs resident in HBM, and `launch()` = the kernel launch(es) of one pass.

    kind "replicates": B_total bootstrap replicates of one genome's count tables, sharded over the ranks (or `replicates`
    per rank); a pass is ONE EM launch.  kind "pairs" (BASELINE configs[4]): G (target, reference) pairs x B replicates =
    G * B rows with per-row epochs, sharded over the ranks; a pass is the block bootstrap of this rank's rows
    (bootstrap_groups_kernel) + ONE EM launch, the count tables never leaving the device."""

    def __init__(self, torch, colate_amd, cd, workloads, dev, dry, world, rank, bins, replicates=0, total_replicates=0, pairs=0):
        self.torch, self.ca, self.dry, self.world, self.rank, self.bins = torch, colate_amd, dry, world, rank, bins
        grid = colate_amd.age_grid()
        epochs, _ = colate_amd.epochs_from_bins(bins)
        self.grid, self.epochs, self.E, self.A = grid, epochs, epochs.size, grid.size
        E, A = self.E, self.A
        self.kind = "pairs" if pairs > 0 else "replicates"
        self.strong = pairs > 0 or total_replicates > 0
        self.B_total = pairs * replicates if pairs > 0 else (total_replicates if total_replicates > 0 else world * replicates)
        self.lo, self.hi = cd.shard_bounds(self.B_total, world, rank)
        self.n_local = n = self.hi - self.lo
        f64 = dict(dtype=torch.float64, device=dev)
        self.d_grid = torch.tensor(grid, **f64)
        self.layout = cd.ShardLayout(self.B_total, E, world)
        self.d_out = self.layout.new_buffer(device=dev)  # the kernel writes its four outputs straight into the packed buffer
        self.v_rates, self.v_ll, self.v_iters, self.v_flags = (v[:n] for v in self.layout.views(self.d_out))
        self.stream = None if dry else torch.cuda.current_stream()
        self.csh = self.cns = None
        if self.kind == "replicates":
            if total_replicates > 0:  # one table of B_total replicates, every rank takes its contiguous range
                csh, cns = workloads.bootstrap_tables(grid, self.B_total, nb=115, scale=11.0, seed=12345)
                csh, cns = csh[self.lo:self.hi], cns[self.lo:self.hi]
            else:                     # every rank bootstraps its own replicates of the same genome
                csh, cns = workloads.bootstrap_tables(grid, n, nb=115, scale=11.0, seed=12345 + 1000 * rank)
            self.csh, self.cns = csh, cns
            self.d_sh, self.d_ns = torch.tensor(csh, **f64), torch.tensor(cns, **f64)
            self.d_ep = torch.tensor(epochs, **f64)
            self.d_init = torch.full((E,), colate_amd.DEFAULT_INIT_RATE, **f64)
            self.what = (f"whole-genome-like LBK-vs-Loschbour-shaped count tables (nb=115 blocks), "
                         + (f"num_bootstrap={self.B_total} sharded over {world} GPU(s)" if total_replicates > 0 else f"num_bootstrap={replicates} per GPU"))
        else:
            # G pairs over one genome: every pair its own block tables (another pairwise Ne, another depth), its own bootstrap
            # weights from its own std::mt19937; all modern samples (one epoch grid) but in the per-row layout of the launch
            G, B, nb = pairs, replicates, 115
            self.G, self.B, self.nb = G, B, nb
            g_lo, g_hi = (self.lo // B, (self.hi - 1) // B + 1) if n else (0, 0)
            self.g_lo, self.g_cnt = g_lo, g_hi - g_lo
            tabs, ws = [], []
            for g in range(g_lo, g_hi):
                sh, ns = workloads.block_tables(grid, nb=nb, scale=6.0 + (g % 10), ne2=8000.0 + 1500.0 * (g // 10 % 10), seed=777 + g)
                she, nse = np.zeros_like(sh), np.zeros_like(ns)
                she[:, 41:60], nse[:, 41:60] = 0.02 * sh[:, 41:60], 0.02 * ns[:, 41:60]  # (age_begin <= 0 mutations: the F redistribution)
                tabs.append((sh, ns, she, nse))
                ws.append(colate_amd.bootstrap_weights(colate_amd.Rng(4242 + g), B, nb))
            cat = lambda k: torch.tensor(np.concatenate([t[k] for t in tabs]) if tabs else np.zeros((1, A)), **f64)  # noqa: E731
            self.d_tabs = [cat(k) for k in range(4)]
            self.d_w = torch.tensor(np.concatenate([w.ravel() for w in ws]) if ws else np.zeros(1), **f64)
            i32, i64 = dict(dtype=torch.int32, device=dev), dict(dtype=torch.int64, device=dev)
            self.d_nb = torch.full((max(self.g_cnt, 1),), nb, **i32)
            self.d_boff = torch.arange(max(self.g_cnt, 1), **i64) * nb
            self.d_woff = torch.arange(max(self.g_cnt, 1), **i64) * (B * nb)
            self.d_age = torch.zeros(max(self.g_cnt, 1), **f64)
            self.d_status = torch.zeros(1, **i32)
            self.d_sh, self.d_ns = torch.empty((max(n, 1), A), **f64), torch.empty((max(n, 1), A), **f64)
            self.d_ep = torch.tensor(np.tile(epochs, (max(n, 1), 1)), **f64)       # per-row epochs
            self.d_init = torch.full((max(n, 1), E), colate_amd.DEFAULT_INIT_RATE, **f64)
            self.what = (f"batched all-pairs (BASELINE configs[4]): {G} (target, reference) pairs x {B} bootstrap replicates = {G * B} rows "
                         f"with per-row epochs, block tables of nb={nb} genome blocks per pair resident in HBM, rows sharded over {world} GPU(s)")

    def launch(self, events=None):
        ca = self.ca
        if self.dry:
            self.v_iters.fill_(1001)
            return
        if not self.n_local:
            return
        if self.kind == "pairs":
            ca.bootstrap_counts_groups_device(self.g_cnt, self.B, self.g_lo, self.lo, self.hi, self.d_grid, self.d_nb, self.d_boff, self.d_woff,
                                              self.d_age, self.d_w, *self.d_tabs, self.d_sh, self.d_ns, self.d_status, stream=self.stream)
            if events:
                events.record(self.stream)  # between the two kernels
        ca.em_batch_device(self.d_grid, self.d_sh[:self.n_local], self.d_ns[:self.n_local], self.d_ep if self.kind == "replicates" else self.d_ep[:self.n_local],
                           self.d_init if self.kind == "replicates" else self.d_init[:self.n_local], self.v_rates, self.v_iters, self.v_ll, self.v_flags,
                           stream=self.stream)


class Runner:
    """Passes of a workload: launch -> (N > 1: the one all-gather) -> results copied to host memory -> stream synchronised."""

    def __init__(self, torch, cd, dist, w, backend):
        self.torch, self.cd, self.dist, self.w = torch, cd, dist, w
        dry, world, layout = w.dry, w.world, w.layout
        self.h_all = torch.zeros(world * layout.nbytes, dtype=torch.uint8)
        if not dry:
            self.h_all = self.h_all.pin_memory()
        self.on_gpu = not dry and (world == 1 or dist.get_backend() == "nccl")
        self.d_all = torch.empty(world * layout.nbytes, dtype=torch.uint8, device=w.d_out.device) if (world > 1 and self.on_gpu) else None
        self.h_local = None if self.on_gpu else (layout.new_buffer() if dry else layout.new_buffer(pin_memory=True))

    def sync(self):
        if self.w.stream is not None:
            self.w.stream.synchronize()

    def one_pass(self, events=None):
        """inputs resident in HBM -> every rank holds all results in host memory"""
        w, cd, dist = self.w, self.cd, self.dist
        if events:
            events[0].record(w.stream)  # on the stream the kernels are launched on
        w.launch(events[3] if events else None)
        if events:
            events[1].record(w.stream)
        if w.world == 1:
            self.h_all.copy_(w.d_out, non_blocking=True)
        elif self.on_gpu:  # the one collective: RCCL all-gather of (8E+16) bytes per row, then one D2H copy
            cd.all_gather_shards(w.d_out, w.layout, dist, out=self.d_all)
            self.h_all.copy_(self.d_all, non_blocking=True)
        else:              # gloo rehearsal: staged through host memory
            self.h_local.copy_(w.d_out, non_blocking=True)
            self.sync()
            cd.all_gather_shards(self.h_local, w.layout, dist, out=self.h_all)
        if events:
            events[2].record(w.stream)
        self.sync()

    def fence(self):
        if not self.w.dry:
            self.torch.cuda.synchronize()
        if self.w.world > 1:
            self.dist.barrier()
        if not self.w.dry:
            self.torch.cuda.synchronize()

    def estimate(self):
        t_w = time.perf_counter()
        for _ in range(3):
            self.one_pass()
        return max((time.perf_counter() - t_w) / 3, 1e-5)

    def timed(self, n_pass):
        """n_pass passes between two fences; HIP events around the kernels of the first 256.  Returns (elapsed s -- the maximum over
        the ranks --, [EM-or-whole kernel ms, gather+copy ms, bootstrap kernel ms])."""
        torch, w, cd, dist = self.torch, self.w, self.cd, self.dist
        mk = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
        n_ev = min(n_pass, 256)
        ev = None if w.dry else [(mk(), mk(), mk(), mk()) for _ in range(n_ev)]
        self.fence()
        t0 = time.perf_counter()
        for k in range(n_pass):
            self.one_pass(ev[k] if (ev and k < n_ev) else None)
        self.fence()
        elapsed = time.perf_counter() - t0
        if w.world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=cd.collective_device(dist))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        nan = float("nan")
        if not (w.n_local and ev):
            return elapsed, [nan, float(np.mean([b.elapsed_time(c) for _, b, c, _ in ev])) if ev else nan, nan]
        if w.kind == "pairs":
            boot = float(np.mean([a.elapsed_time(m) for a, _, _, m in ev]))
            kern = float(np.mean([m.elapsed_time(b) for _, b, _, m in ev]))
        else:
            boot, kern = nan, float(np.mean([a.elapsed_time(b) for a, b, _, _ in ev]))
        return elapsed, [kern, float(np.mean([b.elapsed_time(c) for _, b, c, _ in ev])), boot]


def describe_kernel(colate_amd, torch, w):
    """(kernel instantiation name, build, #CUs) of the EM launch of this workload."""
    E, n = w.E, w.n_local
    try:
        variant = "dry-run" if w.dry else colate_amd.em_kernel_variant(n, E)
    except AttributeError:  # (an older library build under COLATE_AMD_LIB)
        variant = "n/a"
    nch = 1 if E <= 64 else (2 if E <= 128 else 4)
    rows = (1 if E <= 16 else (2 if E <= 32 else 4)) if nch == 1 else 4
    cus = 256 if w.dry else torch.cuda.get_device_properties(w.d_out.device).multi_processor_count
    # (the build without the three-waves-per-SIMD register cap runs where every workgroup has a CU to itself)
    wpe = ", 2" if (variant == "latency-ilp" and nch == 1 and n <= cus) else ", 0"
    return f"em_kernel<0, {nch}, {rows}, {'true' if variant == 'throughput' else 'false'}{wpe}>", variant, cus


def roofline_of(colate_amd, torch, w, iters, kern_ms, boot_ms):
    E, A, n = w.E, w.A, w.n_local
    esteps = int((iters.astype(np.int64) + 1).sum())  # E-steps executed per launch (this rank)
    bytes_per_rep_iter = 2 * A * 8 + A * 8 + 3 * E * 8  # 4992 B at E=23
    kernel_name, variant, cus = describe_kernel(colate_amd, torch, w)
    pmc = pmc_record({"replicates": n, "epochs": int(E), "age_bins": int(A)} if w.kind == "replicates" else
                     {"pairs": int(w.G), "replicates_per_pair": int(w.B), "rows": n, "epochs": int(E), "age_bins": int(A)})
    achieved = esteps * bytes_per_rep_iter / (kern_ms * 1e-3) / 1e9
    crit_iters = int(iters.max()) + 1 if n else 1  # the launch lasts as long as its slowest row (B <= #CUs: all run at once)
    r = roofline(pmc, achieved, kern_ms, kernel_name, variant, esteps * bytes_per_rep_iter, n, cus, -(-n // cus) if n else 1, crit_iters)
    r["kernel_source_sha16"] = kernel_source_sha16()
    if w.kind == "pairs":
        nb = w.nb
        r["bootstrap_kernel"] = {"kernel": "bootstrap_groups_kernel", "kernel_ms": boot_ms, "rows": n,
                                 "algorithmic_bytes_per_launch": n * (4 * nb * A * 8 + nb * 8) + 2 * n * A * 8,
                                 "note": "every row reads its pair's four block tables [nb][A] (out of L2: a pair's 20 rows share them) and its nb "
                                         "weights, and writes two rows of counts; launch-latency-bound at this size"}
    return r


# --------------------------------------------------------------------------------------------- one rank
def run_rank(args):
    import torch

    import colate_amd
    from colate_amd import distributed as cd
    from colate_amd import workloads

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # COLATE_BENCH_BACKEND=gloo: rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks
    # (ranks share devices, the gather goes through host memory); never the measured setup
    backend = os.environ.get("COLATE_BENCH_BACKEND", "nccl")
    # COLATE_BENCH_DRY=1 (tests/test_bench_launcher.py, no GPU): everything but the kernel -- rank spawning, rendezvous,
    # sharding, the packed all-gather (gloo), the JSON line; `value` is null and the line says so
    dry = os.environ.get("COLATE_BENCH_DRY") == "1"
    if dry:
        backend = "gloo"
    ndev = 0 if dry else torch.cuda.device_count()
    if ndev < 1 and not dry:
        sys.exit("bench.py: no GPU visible (colate_amd has no CPU path)")
    if world > 1 and backend == "nccl" and ndev < world:
        sys.exit(f"bench.py: --gpus {world} needs {world} GPUs, {ndev} visible (COLATE_BENCH_BACKEND=gloo rehearses the control flow)")
    if not dry:
        torch.cuda.set_device(local_rank % ndev)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank % ndev))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cpu") if dry else torch.device("cuda", torch.cuda.current_device())

    w = Workload(torch, colate_amd, cd, workloads, dev, dry, world, rank, args.bins, replicates=args.replicates,
                 total_replicates=args.total_replicates, pairs=args.pairs)
    run = Runner(torch, cd, dist, w, backend)
    E, A, n_local, B_total, lo, hi = w.E, w.A, w.n_local, w.B_total, w.lo, w.hi

    # A step = `passes` passes of the hot path over the batch, each complete (launch -> gather -> host copy -> sync).
    # One pass takes about a millisecond, so 20 one-pass steps would be a 20 ms timed region -- too short for anything
    # outside this process (the driver's GPU-busy sampling) to see; `passes` is chosen from the warm-up so that
    # steps x passes lasts >= TIMED_REGION_S, the same on every rank, and stated in config.passes_per_step.
    for _ in range(args.warmup):
        run.one_pass()
    run.fence()
    passes = args.passes_per_step
    if passes <= 0:
        est = run.estimate()
        passes = 1 if dry else max(1, int(np.ceil(1.05 * TIMED_REGION_S / (args.steps * est))))  # (5 % on top: the estimate includes first-touch effects)
        if world > 1:
            t = torch.tensor([passes], dtype=torch.int64, device=cd.collective_device(dist))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            passes = int(t.item())
    n_pass = args.steps * passes
    elapsed, (kern_ms, gather_ms, boot_ms) = run.timed(n_pass)
    # per-rank kernel and gather(+copy) times, so that an inefficiency at N > 1 can be attributed (rank 0 reports them)
    per_rank = [[kern_ms, gather_ms]]
    if world > 1:
        mine = torch.tensor([kern_ms, gather_ms], dtype=torch.float64, device=cd.collective_device(dist))
        allr = torch.empty(world * 2, dtype=torch.float64, device=mine.device)
        dist.all_gather_into_tensor(allr, mine)
        per_rank = allr.cpu().view(world, 2).tolist()
    rates_all, iters_all, ll_all, flags_all = w.layout.unpack(run.h_all)  # every rank has every row's results

    # launches enqueued back to back without the per-step copy and synchronisation (N = 1 only; not `value`)
    device_only = None
    if world == 1 and not dry:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_dev = min(n_pass, 2000)
        for _ in range(n_dev):
            w.launch()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        device_only = {"value": B_total * n_dev / dt, "unit": "replicates/s", "ms_per_pass": 1e3 * dt / n_dev,
                       "note": "launches enqueued back to back, no per-step copy of the rates to the host and no "
                               "per-step synchronisation (round 1's definition of `value`)"}

    out = None
    if rank == 0:
        iters = iters_all[lo:hi]
        status = colate_amd.status_flags(flags_all)
        unresolved = colate_amd.unresolved_epochs(flags_all)
        out = {
            "metric": "bootstrap replicates/sec to EM convergence, whole-genome SGDP mut, 20 epochs",
            "value": None if dry else B_total * n_pass / elapsed,
            "unit": "replicates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if w.strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "DRY RUN: control flow only, the EM kernel was not launched (COLATE_BENCH_DRY=1)" if dry else "synthetic",
            "config": {
                "workload": w.what + f", --bins {args.bins} (E={E} epochs), A={A} age bins, EM to the reference stop rule (min 1001 iterations)",
                "replicates_total": B_total, "replicates_rank0": n_local, "epochs": E, "age_bins": A,
                "parallelism": f"{'rows (pair, replicate)' if w.kind == 'pairs' else 'replicates'} sharded over {world} GPU(s), one "
                               + ("RCCL" if (world > 1 and run.on_gpu) else ("gloo (rehearsal)" if world > 1 else "(no)"))
                               + " all-gather of rates/loglik/iterations/flags per step",
                "step": f"{passes} passes, each: " + ("bootstrap kernel + " if w.kind == "pairs" else "") + "EM kernel launch -> all-gather (N > 1) -> results "
                        "copied to host memory -> stream synchronised, all timed",
                "passes_per_step": passes, "ms_per_pass": 1e3 * elapsed / n_pass, "timed_region_s": elapsed,
                "per_rank_ms": {"kernel": [x[0] for x in per_rank], "gather_and_copy": [x[1] for x in per_rank],
                                "note": "HIP events on the launch stream, mean over the timed passes, one entry per rank"},
                "em_iterations_mean": float(iters_all.mean()), "status_flags_nonzero": int((status != 0).sum()),
                "unresolved_epochs_max": int(unresolved.max()),
            },
            "roofline": roofline_of(colate_amd, torch, w, iters, kern_ms, boot_ms),
        }
        if w.kind == "pairs":
            out["config"]["pairs"] = w.G
            out["config"]["replicates_per_pair"] = w.B
        if device_only:
            out["device_only"] = device_only
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    if dry:
        emit(out)
        return
    if world == 1 and not args.no_other_configs:
        # the other single-GPU shapes of BASELINE.json, the same pass loop, about a second each
        shapes = [("configs[3]: --bins 2,7.95,0.05 (122 epochs), num_bootstrap=100", dict(bins="2,7.95,0.05", replicates=100)),
                  ("configs[2] on one GPU: num_bootstrap=1000", dict(bins=BINS, total_replicates=1000)),
                  ("configs[4] on one GPU: 100 pairs x 20 replicates, per-row epochs, bootstrap + EM per pass", dict(bins=BINS, pairs=100, replicates=20))]
        if w.kind == "pairs" or args.bins != BINS or args.total_replicates or args.replicates != B_PER_GPU:
            shapes.insert(0, ("configs[1]: --bins 3,7,0.2, num_bootstrap=100", dict(bins=BINS, replicates=100)))
        others = []
        for name, kw in shapes:
            if kw == dict(bins=args.bins, replicates=args.replicates, **({"pairs": args.pairs} if args.pairs else {})):
                continue
            w2 = Workload(torch, colate_amd, cd, workloads, dev, False, 1, 0, kw["bins"], replicates=kw.get("replicates", 0),
                          total_replicates=kw.get("total_replicates", 0), pairs=kw.get("pairs", 0))
            r2 = Runner(torch, cd, None, w2, backend)
            for _ in range(2):
                r2.one_pass()
            np2 = max(3, int(np.ceil(OTHER_CONFIG_S / r2.estimate())))
            el2, (k2, g2, b2) = r2.timed(np2)
            _, it2, _, fl2 = w2.layout.unpack(r2.h_all)
            others.append({"config": name, "value": w2.B_total * np2 / el2, "unit": "replicates/s", "passes": np2, "ms_per_pass": 1e3 * el2 / np2,
                           "timed_region_s": el2, "em_iterations_mean": float(it2.mean()),
                           "status_flags_nonzero": int((colate_amd.status_flags(fl2) != 0).sum()),
                           "unresolved_epochs_max": int(colate_amd.unresolved_epochs(fl2).max()),
                           "roofline": roofline_of(colate_amd, torch, w2, it2, k2, b2)})
            del w2, r2
        out["other_configs"] = others
    if world == 1 and not args.no_host_path and w.kind == "replicates":
        # PCIe-inclusive rate through the host-pointer entry point (staging copies + launch + copy back inside
        # the call, on the library's cached workspace); reported beside `value`, never as it
        colate_amd.em_batch(w.grid, w.csh, w.cns, w.epochs)
        t1 = time.perf_counter()
        for _ in range(5):
            colate_amd.em_batch(w.grid, w.csh, w.cns, w.epochs)
        out["host_path"] = {"value": 5 * n_local / (time.perf_counter() - t1), "unit": "replicates/s",
                            "note": "colate_em_batch with host buffers: PCIe-inclusive"}
    if not args.no_cxx_rccl_check and w.kind == "replicates":
        out["cxx_rccl"] = cxx_rccl_check(colate_amd, world, w.grid, w.csh, w.cns, w.epochs, args.bins)
    if world == 1 and not args.no_cpu_baseline and w.kind == "replicates":
        out["cpu_baseline"] = cpu_baseline(w.grid, w.csh, w.cns, w.epochs, rates_all[lo:hi], iters_all[lo:hi], args.bins)
    emit(out)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    run_rank(args)


if __name__ == "__main__":
    main()
