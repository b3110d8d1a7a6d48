"""Root-cause probe for the red driver test of round 5 (tests/test_dist_gloo.py, two gloo ranks on the one GPU, shard-in clause).

Two spawned ranks share cuda:0 over gloo and loop the failing clause `iters` times.  Every iteration separates the three candidate causes:
  (1) gloo's device staging: `out` is snapshotted right after all_gather_into_tensor and again after a full device synchronize - a
      difference is a copy-back that had not landed when the collective returned (mode "device"), impossible in mode "host" where only
      CPU tensors are ever handed to gloo;
  (2) the local recomputation: every block is recomputed twice - a difference is the solver itself under two tenants;
  (3) the peer's rows: per-block max |gathered - recomputed|, own block and peer block reported separately.
Writes one JSON line per failing iteration and one summary line per rank and mode.

    python tools/two_rank_gather_repro.py --iters 200 --out gpurun_out/r06/gather_repro.jsonl
"""
import argparse
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, iters, mode, q):
    from helpers import latents, reachable_poses, tiny_model
    from ikflow_amd.dist import draw_latent_shard, shard_bounds
    from ikflow_amd.ikflow_solver import IKFlowSolver

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = "cuda:0"
        robot, hp, lay, sd = tiny_model(seed=4)
        s = IKFlowSolver(hp, robot)
        s.load_state_dict_tensors(sd)
        _, poses = reachable_poses(robot, n, 5)
        poses = poses.to(dev)
        lo, hi = shard_bounds(n, world, rank)
        counts = [shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0] for r in range(world)]
        rows_max = max(counts)
        kernel = s.engine(dev).dominant_kernel_name(hi - lo)
        fails, stale, unstable, worst = [], 0, 0, 0.0
        for it in range(iters):
            lat = draw_latent_shard(hi - lo, lay.dim, dev, 31, rank)
            local = s.generate_ik_solutions(poses[lo:hi].contiguous(), latent=lat)
            pad = torch.zeros((rows_max, local.shape[1]), dtype=local.dtype, device=dev)
            pad[: hi - lo] = local
            if mode == "device":      # what round 5 did: gloo handed device tensors (its own host staging on its own streams)
                out = torch.empty((world * rows_max, local.shape[1]), dtype=local.dtype, device=dev)
                torch.cuda.current_stream().synchronize()
                dist.all_gather_into_tensor(out, pad)
                snap1 = out.clone()
                torch.cuda.synchronize()
                snap2 = out.clone()
            else:                     # gloo sees CPU tensors only
                host = torch.empty((world * rows_max, local.shape[1]), dtype=local.dtype)
                dist.all_gather_into_tensor(host, pad.cpu())
                snap1 = host.to(dev)
                torch.cuda.synchronize()
                snap2 = host.to(dev)
            got = torch.cat([snap1[r * rows_max : r * rows_max + counts[r]] for r in range(world)], dim=0)
            blocks, blocks2 = [], []
            for r in range(world):
                l, h = shard_bounds(n, world, r)
                for dst in (blocks, blocks2):
                    dst.append(s.generate_ik_solutions(poses[l:h].contiguous(), latent=draw_latent_shard(h - l, lay.dim, dev, 31, r)))
            ref, ref2 = torch.cat(blocks, dim=0), torch.cat(blocks2, dim=0)
            d_stale = float((snap1 - snap2).abs().max())
            d_unst = float((ref - ref2).abs().max())
            per_block = []
            for r in range(world):
                l, h = shard_bounds(n, world, r)
                per_block.append(float((got[l:h] - ref[l:h]).abs().max()))
            stale += d_stale > 0
            unstable += d_unst > 0
            worst = max(worst, max(per_block))
            if max(per_block) > 1e-5 or d_stale > 0 or d_unst > 1e-5:
                rec = {"rank": rank, "mode": mode, "iter": it, "per_block_max_abs": per_block, "own_block": rank,
                       "snap_right_after_vs_after_device_sync": d_stale, "recomputed_twice_max_abs": d_unst}
                if d_stale > 0:
                    bad = (snap1 != snap2).any(dim=1).nonzero().flatten()
                    rec["stale_rows_first_last_count"] = [int(bad[0]), int(bad[-1]), int(bad.numel())]
                fails.append(rec)
        q.put({"rank": rank, "mode": mode, "iters": iters, "n": n, "kernel": kernel, "failing_iters": len(fails), "stale_snapshots": stale,
               "unstable_recomputations": unstable, "worst_block_abs": worst, "fails": fails[:20]})
    finally:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--n", type=int, default=1501)
    ap.add_argument("--modes", default="device,host")
    ap.add_argument("--out", default="gpurun_out/r06/gather_repro.jsonl")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    ctx = mp.get_context("spawn")
    with open(a.out, "w") as f:
        for mode in a.modes.split(","):
            q, port, world = ctx.Queue(), _free_port(), 2
            procs = [ctx.Process(target=_worker, args=(r, world, port, a.n, a.iters, mode, q)) for r in range(world)]
            for p in procs:
                p.start()
            res = [q.get(timeout=1200) for _ in range(world)]
            for p in procs:
                p.join(timeout=60)
            for r in sorted(res, key=lambda x: x["rank"]):
                line = json.dumps(r)
                f.write(line + "\n")
                print(line[:600])


if __name__ == "__main__":
    main()
