"""Multi-GPU path on the CPU: replicate sharding + the single all-gather, world_size 2 over gloo.
The per-shard compute is the oracle here (no GPU in this container); on the GPU box bench.py runs the
same sharding with the HIP kernel and RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as ol
from colate_amd import distributed as cd
from colate_amd import workloads


def test_shard_bounds_cover_everything():
    for B in (0, 1, 5, 100, 1000, 1001):
        for world in (1, 2, 3, 8):
            spans = [cd.shard_bounds(B, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
            assert max(h - l for l, h in spans) <= cd.max_shard(B, world)


def _worker(rank, world, port, B, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grid = ol.age_grid()
    ep, _ = ol.epochs_from_bins("3,6,0.5")
    csh, cns = workloads.bootstrap_tables(grid, B, nb=9, scale=1.0)

    def run_local(g, s, n, e, **kw):
        return ol.em_batch(g, s, n, e, max_iter=40, min_iter=1000)

    rates, iters, ll, flags = cd.em_batch_sharded(run_local, grid, csh, cns, ep, dist)
    if rank == 0:
        np.save(out, np.concatenate([rates, iters[:, None], ll[:, None]], axis=1))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 8])
def test_sharded_em_equals_single_process(tmp_path, B):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r.npy")
    mp.spawn(_worker, args=(2, port, B, out), nprocs=2, join=True)
    got = np.load(out)
    grid = ol.age_grid()
    ep, _ = ol.epochs_from_bins("3,6,0.5")
    csh, cns = workloads.bootstrap_tables(grid, B, nb=9, scale=1.0)
    rates, iters, ll, _ = ol.em_batch(grid, csh, cns, ep, max_iter=40, min_iter=1000)
    assert np.array_equal(got[:, :-2], rates) and np.array_equal(got[:, -2], iters) and np.array_equal(got[:, -1], ll)
