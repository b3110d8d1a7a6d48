"""World-size-2 (and 3) gloo tests of the multi-GPU sharding/gather logic (ikflow_amd/dist.py) on CPU.
The per-rank compute is a stand-in row-wise function - the engine itself only runs on the GPU - so what is covered is
exactly what differs between N=1 and N>1: block bounds, order preservation, ragged blocks, the single all-gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ikflow_amd.dist import gather_rows, shard_bounds, sharded_rows


def test_shard_bounds_cover_and_order():
    for n in (0, 1, 7, 8, 4096, 1000003):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(n, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [h - l for l, h in edges]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)  # same full inputs on every rank
        poses = torch.randn(n, 7, generator=g)
        latent = torch.randn(n, 7, generator=g)

        def compute(p, l):  # row-wise stand-in for solver.generate_ik_solutions
            assert p.shape[0] == l.shape[0] == shard_bounds(n, world, rank)[1] - shard_bounds(n, world, rank)[0]
            return torch.tanh(p) + 0.5 * l

        full = sharded_rows(compute, poses, latent)
        expect = torch.tanh(poses) + 0.5 * latent
        ok = bool(torch.equal(full, expect))
        lo, hi = shard_bounds(n, world, rank)
        ok = ok and bool(torch.equal(gather_rows(expect[lo:hi].contiguous(), n), expect))
        q.put((rank, ok, tuple(full.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 4096), (2, 4097), (3, 10), (2, 1)])
def test_sharded_rows_gloo(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok, f"rank {rank} gathered a wrong tensor"
        assert shape == (n, 7)
