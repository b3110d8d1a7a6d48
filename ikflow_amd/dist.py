"""Row-sharded multi-GPU form of the hot path: one process per GPU, weights replicated, contiguous row
blocks, no collective inside the data path and ONE all-gather of the solutions at the end (RCCL over xGMI when the
process group's backend is "nccl"; the same code runs on "gloo" for the CPU tests of the sharding logic).

The reference has no distributed code at all (no NCCL/MPI call sites); this is the MI355X-native addition the
north star asks for ("huge pose batches shard embarrassingly across the 8 GPUs of one node with a single RCCL gather").

Two families of entry points:
  * full-tensor forms (`sharded_rows`, `sharded_generate_ik_solutions`, `sharded_generate_exact_ik_solutions`): every rank passes the SAME
    full [n x 7] pose tensor and only touches its block - the form parity runs use (identical seeds give the single-process result);
  * shard-in forms (`*_from_shard`, `ShardedStepper`): a rank holds ONLY its own block - poses generated or loaded in place, latents drawn
    in place from `base_seed + rank` (SURVEY 8(e) "Determinism") - so nothing of size O(n) is replicated except the gathered result
    itself.  `bench.py` (weak, --global-batch, --million) steps through `ShardedStepper`.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Sequence, Tuple

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
    dist.all_gather_into_tensor(out, pad, group=group)
    if n_total == world * rows_max:
        return out
    pieces = []
    for r in range(world):
        l, h = shard_bounds(n_total, world, r)
        pieces.append(out[r * rows_max : r * rows_max + (h - l)])
    return torch.cat(pieces, dim=0)


def gather_blocks(local: torch.Tensor, group=None, counts: Optional[Sequence[int]] = None) -> torch.Tensor:
    """All-gather row blocks of ANY sizes, rank order = row order, into the full tensor on every rank.  `counts` (rows of every rank's
    block) when the caller knows them; otherwise they are exchanged first (one all-gather of `world` int64 - the only extra collective of
    the shard-in forms).  The payload itself moves in one padded all_gather_into_tensor."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if counts is None:
        mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        every = torch.empty(world, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(every, mine, group=group)
        counts = [int(c) for c in every.tolist()]
    counts = [int(c) for c in counts]
    assert len(counts) == world and counts[rank] == local.shape[0], f"rank {rank}: block of {local.shape[0]} rows, counts say {counts}"
    rows_max = max(counts) if counts else 0
    cols = tuple(local.shape[1:])
    if rows_max == 0:
        return torch.empty((0,) + cols, dtype=local.dtype, device=local.device)
    if local.shape[0] == rows_max:
        pad = local.contiguous()
    else:
        pad = torch.zeros((rows_max,) + cols, dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    out = torch.empty((world * rows_max,) + cols, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    if all(c == rows_max for c in counts):
        return out
    return torch.cat([out[r * rows_max : r * rows_max + counts[r]] for r in range(world)], dim=0)


def draw_latent_shard(rows: int, dim: int, device, base_seed: int, rank: int, latent_distribution: str = "gaussian",
                      latent_scale: float = 1.0) -> torch.Tensor:
    """This rank's [rows x dim] latent block, drawn in place from its own generator seeded `base_seed + rank`: no rank
    draws - or holds - another rank's block.  Same distributions as the reference's draw_latent (ikflow_solver.py:16-29)."""
    assert latent_distribution in ("gaussian", "uniform")
    g = torch.Generator(device=device)
    g.manual_seed(int(base_seed) + int(rank))
    if latent_distribution == "gaussian":
        return latent_scale * torch.randn((rows, dim), generator=g, device=device)
    return 2 * latent_scale * torch.rand((rows, dim), generator=g, device=device) - latent_scale


def sharded_generate_ik_solutions_from_shard(solver, poses_shard: torch.Tensor, latent_shard: Optional[torch.Tensor] = None, *,
                                             base_seed: int = 0, counts: Optional[Sequence[int]] = None, gather: bool = True,
                                             group=None, **kw):
    """generate_ik_solutions where every rank passes ONLY its own pose block (rank order = row order of the gathered result).
    `latent_shard` None: drawn in place from `base_seed + rank`.  Returns the full [n x ndof] on every rank (`gather=False`: this
    rank's block alone - no collective at all)."""
    rank = dist.get_rank(group)
    if latent_shard is None:
        latent_shard = draw_latent_shard(poses_shard.shape[0], solver.network_width, poses_shard.device, base_seed, rank,
                                         kw.get("latent_distribution", "gaussian"), kw.get("latent_scale", 1.0))
    if poses_shard.shape[0] == 0:   # (a rank without rows still takes part in the collective)
        ndof = getattr(solver, "ndof", None) or getattr(getattr(solver, "robot", None), "ndof", None) or latent_shard.shape[1]
        local = torch.empty((0, int(ndof)), dtype=torch.float32, device=poses_shard.device)
    else:
        local = solver.generate_ik_solutions(poses_shard, n=(1 if poses_shard.shape[0] == 1 else None), latent=latent_shard, **kw)
    return gather_blocks(local, group, counts) if gather else local


def sharded_generate_exact_ik_solutions_from_shard(solver, poses_shard: torch.Tensor, *, counts: Optional[Sequence[int]] = None,
                                                   group=None, **kw):
    """generate_exact_ik_solutions on this rank's own pose block; solutions and valid flags of every rank in one collective."""
    if poses_shard.shape[0] == 0:   # (a rank without rows still takes part in the collective)
        ndof = getattr(solver, "ndof", None) or getattr(getattr(solver, "robot", None), "ndof", None)
        sol = torch.empty((0, int(ndof)), dtype=torch.float32, device=poses_shard.device)
        valid = torch.empty((0,), dtype=torch.bool, device=poses_shard.device)
    else:
        sol, valid = solver.generate_exact_ik_solutions(poses_shard, **kw)
    full = gather_blocks(torch.cat([sol, valid.to(sol.dtype)[:, None]], dim=1), group, counts)
    return full[:, :-1].contiguous(), full[:, -1] > 0.5


class ShardedStepper:
    """Repeated steps over a fixed shard: one step = `compute()` on this rank's row block, then the path's one collective -
    all_gather_into_tensor of the [rows x cols] result into a preallocated, double-buffered [world * rows x cols] tensor.  On the GPU the
    gather runs on its own stream behind an event, so the gather of step i overlaps the flow of step i + 1; `fence()` drains both streams
    and barriers.  On CPU tensors (gloo) the same calls run inline."""

    def __init__(self, compute: Callable[[], torch.Tensor], world: int, rank: int, rows: int, cols: int, device, use_dist: bool,
                 n_buf: int = 2, group=None):
        self.compute, self.world, self.rank, self.rows, self.device, self.use_dist = compute, world, rank, rows, device, use_dist
        self.group = group
        self.cuda = torch.device(device).type == "cuda"
        self.n_buf = n_buf
        self.i = 0
        self.keep = [None] * n_buf
        self.gathered = [torch.empty((world * rows, cols), dtype=torch.float32, device=device) for _ in range(n_buf)] if use_dist else None
        self.comm_stream = torch.cuda.Stream(device) if (use_dist and self.cuda) else None

    def step(self) -> torch.Tensor:
        sol = self.compute()
        if self.use_dist:
            k = self.i % self.n_buf
            self.i += 1
            if self.comm_stream is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self.comm_stream.wait_event(ev)
                with torch.cuda.stream(self.comm_stream):
                    dist.all_gather_into_tensor(self.gathered[k], sol, group=self.group)
                sol.record_stream(self.comm_stream)
            else:
                dist.all_gather_into_tensor(self.gathered[k], sol, group=self.group)
            self.keep[k] = sol
        return sol

    def fence(self) -> None:
        if self.use_dist:
            if self.comm_stream is not None:
                torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
            dist.barrier(group=self.group)
        if self.cuda:
            torch.cuda.synchronize(self.device)

    def last_gathered(self) -> Optional[torch.Tensor]:
        """The gathered result of the most recent step - None before the first one (or without a process group).  With a side stream the
        gather may still be in flight: call fence() first."""
        if not self.use_dist or self.i == 0:
            return None
        return self.gathered[(self.i - 1) % self.n_buf]


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
