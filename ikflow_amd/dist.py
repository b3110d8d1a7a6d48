"""Row-sharded multi-GPU form of the hot path: one process per GPU, weights replicated, contiguous row
blocks, no collective inside the data path and ONE all-gather of the solutions at the end (RCCL over xGMI when the
process group's backend is "nccl"; the same code runs on "gloo" for the CPU tests of the sharding logic).

The reference has no distributed code at all (no NCCL/MPI call sites); this is the MI355X-native addition the
north star asks for ("huge pose batches shard embarrassingly across the 8 GPUs of one node with a single RCCL gather").
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

# RCCL exchanges peer memory handles when the group forms; this host driver supports the dmabuf form only.  Harmless when the
# launcher already exported it; it has to be in the environment before the HIP runtime initialises.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced row block of `rank`: the first n % world ranks get one extra row. Output order = input order."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather row blocks produced under `shard_bounds` into the full [n_total x C] tensor on every rank.
    Blocks are padded to the largest block so a single fixed-size all_gather_into_tensor moves everything."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(n_total, world, rank)
    assert local.shape[0] == hi - lo, f"rank {rank}: local block has {local.shape[0]} rows, expected {hi - lo}"
    rows_max = (n_total + world - 1) // world
    cols = local.shape[1:]
    pad = torch.zeros((rows_max,) + tuple(cols), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * rows_max,) + tuple(cols), dtype=local.dtype, device=local.device)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # gloo stages device tensors through the host on its own streams; on this ROCm build that staging copy was seen to start before
        # the work queued on the current stream had finished (stale rows in 1 of 8 runs of the two-ranks-on-one-GPU test) - with gloo
        # (tests only: RCCL orders its kernels behind the current stream itself) the producer stream is drained first
        torch.cuda.current_stream(local.device).synchronize()
    dist.all_gather_into_tensor(out, pad, group=group)
    if n_total == world * rows_max:
        return out
    pieces = []
    for r in range(world):
        l, h = shard_bounds(n_total, world, r)
        pieces.append(out[r * rows_max : r * rows_max + (h - l)])
    return torch.cat(pieces, dim=0)


def sharded_rows(
    compute: Callable[[torch.Tensor, Optional[torch.Tensor]], torch.Tensor],
    target_poses: torch.Tensor,
    latent: Optional[torch.Tensor] = None,
    group=None,
) -> torch.Tensor:
    """Run `compute(poses_block, latent_block)` on this rank's row block and all-gather the results.
    Every rank passes the same full `target_poses` (and `latent`); only its block is touched."""
    n = target_poses.shape[0]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(n, world, rank)
    lat = None if latent is None else latent[lo:hi].contiguous()
    local = compute(target_poses[lo:hi].contiguous(), lat)
    return gather_rows(local, n, group)


def sharded_generate_ik_solutions(solver, target_poses: torch.Tensor, latent: Optional[torch.Tensor] = None, group=None, **kw):
    """generate_ik_solutions over rows sharded across the ranks of `group`; returns the full [n x ndof] on every rank.

    When `latent` is None the FULL [n x D] latent is drawn on every rank with the reference's own call
    (draw_latent -> torch's global generator on the pose device, ikflow_solver.py:16-29,341) and each rank keeps its
    block: with the usual identical seed on every rank the result equals the single-process call on the same seed, and
    no two shards ever see the same latent block."""
    if latent is None:
        from ikflow_amd.ikflow_solver import draw_latent

        latent = draw_latent(kw.get("latent_distribution", "gaussian"), kw.get("latent_scale", 1.0),
                             (target_poses.shape[0], solver.network_width), target_poses.device)

    def compute(p, l):
        return solver.generate_ik_solutions(p, n=(1 if p.shape[0] == 1 else None), latent=l, **kw)

    return sharded_rows(compute, target_poses, latent, group)


def sharded_generate_exact_ik_solutions(solver, target_poses: torch.Tensor, group=None, **kw):
    """generate_exact_ik_solutions sharded by target pose (retry rounds are per pose, so shards never interact)."""
    n = target_poses.shape[0]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(n, world, rank)
    sol, valid = solver.generate_exact_ik_solutions(target_poses[lo:hi].contiguous(), **kw)
    packed = torch.cat([sol, valid.to(sol.dtype)[:, None]], dim=1)  # one collective for both outputs
    full = gather_rows(packed, n, group)
    return full[:, :-1].contiguous(), full[:, -1] > 0.5
