"""Replicate sharding for multi-GPU runs (one process per GPU, torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests).

Bootstrap replicates are independent once the count tables exist (coal.cpp:3675-3846 loops over
them sequentially), so the path shards with NO data-path collective: rank r runs the EM kernel on
the contiguous replicates [lo, hi) and the only exchange is ONE all-gather of the per-replicate
results at the end (B*E doubles in total: 18 KB at B=100, E=23)."""
import numpy as np


def shard_bounds(B, world, rank):
    """Contiguous, balanced split of B replicates over `world` ranks: returns (lo, hi)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_shard(B, world):
    return (B + world - 1) // world


def all_gather_replicates(local, B, dist, device=None):
    """Gather per-replicate rows (tensor [n_local, ...]) from all ranks into [B, ...] in replicate
    order, on every rank, with a single all_gather (rows padded to the largest shard)."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    n = max_shard(B, world)
    pad = torch.zeros((n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    pieces = []
    for r in range(world):
        lo, hi = shard_bounds(B, world, r)
        pieces.append(out[r * n: r * n + (hi - lo)])
    return torch.cat(pieces, dim=0)


def em_batch_sharded(run_local, age_grid, cnt_shared, cnt_notshared, epochs, dist, **kw):
    """Run `run_local(age_grid, cnt_sh_shard, cnt_ns_shard, epochs, **kw) -> (rates, iters, ll, flags)`
    (numpy in/out, e.g. colate_amd.em_batch) on this rank's shard and all-gather the results.
    Returns full-size numpy arrays on every rank."""
    import torch

    B = cnt_shared.shape[0]
    lo, hi = shard_bounds(B, dist.get_world_size(), dist.get_rank())
    E = np.asarray(epochs).size
    if hi > lo:
        rates, iters, ll, flags = run_local(age_grid, cnt_shared[lo:hi], cnt_notshared[lo:hi], epochs, **kw)
    else:
        rates, iters, ll, flags = np.zeros((0, E)), np.zeros(0, np.int32), np.zeros(0), np.zeros(0, np.int32)
    packed = np.concatenate([rates, iters[:, None].astype(np.float64), ll[:, None], flags[:, None].astype(np.float64)], axis=1)
    full = all_gather_replicates(torch.from_numpy(np.ascontiguousarray(packed)), B, dist).numpy()
    return full[:, :E], full[:, E].astype(np.int32), full[:, E + 1], full[:, E + 2].astype(np.int32)
