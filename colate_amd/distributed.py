"""Replicate sharding for multi-GPU runs (one process per GPU, torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests and rehearsals).

Bootstrap replicates are independent once the count tables exist (coal.cpp:3675-3846 loops over
them sequentially), so the path shards with NO data-path collective: rank r runs the EM kernel on
the contiguous replicates [lo, hi) and the only exchange is ONE all-gather of the per-replicate
results at the end: (8 E + 16) bytes per replicate (rates, log-likelihood, iterations, flags), 20 KB
in total at B = 100, E = 23.

Every rank packs its results into one byte buffer (`ShardLayout`), a single `all_gather_into_tensor`
moves the buffers, and the gathered bytes are unpacked in replicate order.  bench.py, `em_batch_sharded`
below and tests/test_distributed_cpu.py (gloo, world size 2) all go through the same three steps; the
C++ form of the same thing is colate_em_batch_allgather (colate_amd/csrc/colate_comm.cpp)."""
import numpy as np


def shard_bounds(B, world, rank):
    """Contiguous, balanced split of B replicates over `world` ranks: returns (lo, hi).
    (The same split as colate_shard_bounds in the C ABI.)"""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_shard(B, world):
    return (B + world - 1) // world


class ShardLayout:
    """Byte layout of one rank's results, sized for the largest shard so that all ranks send equally much:
        rates[n_max][E] f64 | loglik[n_max] f64 | iters[n_max] i32 | flags[n_max] i32      (+ padding to 8 bytes)
    """

    def __init__(self, B, E, world):
        self.B, self.E, self.world = int(B), int(E), int(world)
        self.n_max = max_shard(self.B, self.world)
        n = self.n_max
        self.off_rates = 0
        self.off_ll = n * self.E * 8
        self.off_iters = self.off_ll + n * 8
        self.off_flags = self.off_iters + n * 4
        self.nbytes = (self.off_flags + n * 4 + 7) // 8 * 8

    def views(self, buf):
        """Typed views (rates [n_max][E], loglik, iters, flags) into one rank's byte buffer `buf` (a 1-D uint8 torch
        tensor of nbytes, on any device): what the EM kernel writes into directly."""
        import torch

        n, E = self.n_max, self.E
        rates = buf[self.off_rates:self.off_ll].view(torch.float64).view(n, E)
        ll = buf[self.off_ll:self.off_iters].view(torch.float64)
        iters = buf[self.off_iters:self.off_flags].view(torch.int32)
        flags = buf[self.off_flags:self.off_flags + n * 4].view(torch.int32)
        return rates, ll, iters, flags

    def new_buffer(self, device="cpu", pin_memory=False):
        import torch

        if pin_memory:
            return torch.zeros(self.nbytes, dtype=torch.uint8).pin_memory()
        return torch.zeros(self.nbytes, dtype=torch.uint8, device=device)

    def unpack(self, gathered):
        """gathered: uint8 CPU tensor [world * nbytes] -> (rates[B][E], iters[B], loglik[B], flags[B]) as numpy arrays,
        in replicate order."""
        g = gathered.view(self.world, self.nbytes)
        R, I, L, F = [], [], [], []
        for r in range(self.world):
            lo, hi = shard_bounds(self.B, self.world, r)
            rates, ll, iters, flags = self.views(g[r])
            R.append(rates[: hi - lo].numpy()), L.append(ll[: hi - lo].numpy())
            I.append(iters[: hi - lo].numpy()), F.append(flags[: hi - lo].numpy())
        return np.concatenate(R, axis=0), np.concatenate(I), np.concatenate(L), np.concatenate(F)


def all_gather_shards(local_buf, layout, dist, out=None):
    """The ONE collective of the path: every rank's packed byte buffer to every rank.
    `local_buf` lives where the backend wants it (HBM for "nccl" = RCCL, host memory for "gloo");
    returns a uint8 tensor [world * nbytes] on the same device (`out` if given)."""
    import torch

    if out is None:
        out = torch.empty(layout.world * layout.nbytes, dtype=torch.uint8, device=local_buf.device)
    dist.all_gather_into_tensor(out, local_buf)
    return out


def collective_device(dist):
    """Where tensors handed to the collectives must live for the initialised backend."""
    import torch

    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def em_batch_sharded(run_local, age_grid, cnt_shared, cnt_notshared, epochs, dist, **kw):
    """Run `run_local(age_grid, cnt_sh_shard, cnt_ns_shard, epochs, **kw) -> (rates, iters, ll, flags)`
    (numpy in/out, e.g. colate_amd.em_batch) on this rank's shard and all-gather the results.
    Returns full-size numpy arrays on every rank."""
    import torch

    B = cnt_shared.shape[0]
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(B, world, rank)
    E = np.asarray(epochs).size
    layout = ShardLayout(B, E, world)
    buf = layout.new_buffer()
    if hi > lo:
        rates, iters, ll, flags = run_local(age_grid, cnt_shared[lo:hi], cnt_notshared[lo:hi], epochs, **kw)
        v_rates, v_ll, v_iters, v_flags = layout.views(buf)
        v_rates[: hi - lo] = torch.from_numpy(np.ascontiguousarray(rates, dtype=np.float64))
        v_ll[: hi - lo] = torch.from_numpy(np.ascontiguousarray(ll, dtype=np.float64))
        v_iters[: hi - lo] = torch.from_numpy(np.ascontiguousarray(iters, dtype=np.int32))
        v_flags[: hi - lo] = torch.from_numpy(np.ascontiguousarray(flags, dtype=np.int32))
    dev = collective_device(dist)  # RCCL moves device memory only
    gathered = all_gather_shards(buf.to(dev), layout, dist)
    return layout.unpack(gathered.cpu())
