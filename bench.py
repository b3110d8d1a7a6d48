#!/usr/bin/env python
"""bench.py - IK solutions/sec of the MI355X engine on BASELINE.json's headline configuration.

  python bench.py --gpus 1 --steps 50 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one pass of the hot path (IKFlowSolver.generate_ik_solutions: 12-block conditional-flow inverse pass, slice,
clamp) over one batch of B=4096 synthetic target poses per GPU, inputs resident in HBM.  Weights are seeded random
(nn.Linear default init) of the released architecture `panda__full__lp191_5.25m` - the released weight file is a
remote URL and there is no network.  With N > 1 every rank processes its own 4096-row shard (weak scaling) and the
step ends with the one RCCL all-gather of the [4096 x 7] solutions.

Rank 0 prints ONE JSON line: metric/value/unit per BASELINE.json, plus `roofline` for the dominant kernel
(k_flow_gemm: the [B x 1024].[1024 x 1024]^T fp32-MFMA contraction) and `cpu_baseline` (the torch-CPU oracle timed on
this host's cores on a bounded sample; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

MODEL = "panda__full__lp191_5.25m"
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, 2.4 GHz


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--batch", type=int, default=4096, help="target poses per GPU per step")
    p.add_argument("--model", type=str, default=MODEL)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="bound on the CPU-baseline sample")
    p.add_argument("--gemm-variant", type=int, default=-1)
    p.add_argument("--precision", type=str, default="f32", choices=["f32", "f16x3"],
                   help="arithmetic of the hidden contractions: exact-f32 MFMA, or the error-compensated 3x f16 MFMA split")
    p.add_argument("--no-split-extra", action="store_true", help="skip the extra f16x3 measurement")
    p.add_argument("--extras", action="store_true", help="also time B=512 approx and exact IK (reported under `extra`)")
    p.add_argument("--million", action="store_true",
                   help="BASELINE config 5: 1,000,000 target poses per step sharded over the ranks (batch = 1e6 / world), "
                        "one all-gather per step")
    return p.parse_args()


def cpu_baseline(sd, layout, limits, poses_cpu, latent_cpu, budget_s):
    """The oracle (op-for-op the reference's PyTorch-CPU path) on this host, same batch, bounded wall time.
    torch's default thread count (= all logical cores, what the reference would use) is timed first; because MKL scales
    badly past ~32 threads on [4096x1024] GEMMs, 16/32/64 threads are tried too and the best rate is reported with the
    thread count it used."""
    from oracle import flow_oracle as fo

    n = poses_cpu.shape[0]
    default_threads = torch.get_num_threads()
    cands = sorted({default_threads} | {t for t in (16, 32, 64) if t < default_threads})
    per = max(1.0, budget_s / (len(cands) + 1))
    best = None
    notes = []
    for th in cands:
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        fo.generate_ik_solutions_torch(sd, layout, limits, poses_cpu, latent_cpu)  # warm-up pass
        t_warm = time.perf_counter() - t0
        reps = max(1, min(20, int(per / max(t_warm, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(reps):
            fo.generate_ik_solutions_torch(sd, layout, limits, poses_cpu, latent_cpu)
        dt = time.perf_counter() - t0
        rate = n * reps / dt
        notes.append(f"{th} thr: {rate:.0f}/s ({reps} passes, {dt:.1f} s)")
        if best is None or rate > best[0]:
            best = (rate, th)
    torch.set_num_threads(default_threads)
    return {
        "value": best[0],
        "unit": "IK solutions/s",
        "cores": best[1],
        "kind": "port",
        "sample": f"passes of the same B={n} batch through oracle/flow_oracle.py (torch-CPU fp32); " + "; ".join(notes)
                  + f"; host has {os.cpu_count()} logical cores",
    }


def pmc_traffic_per_launch():
    """HBM bytes per dominant-kernel launch from the newest committed rocprofv3 PMC summary (profiles/rNN_pmc_summary.json:
    separate FETCH_SIZE / WRITE_SIZE passes over this same command, gfx950 x2 read correction applied). PMC counters
    cannot be collected from inside the process, so this is the recorded figure, or None if no summary is present."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            return json.load(f).get("dominant_kernel_traffic_bytes_per_launch"), os.path.basename(files[-1])
    except Exception:
        return None, None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the engine has no CPU path)"
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)  # launched by torch.distributed.run
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from ikflow_amd.ikflow_solver import IKFlowSolver
    from ikflow_amd.model import hparams_for, layout_from, random_state_dict
    from ikflow_amd.robots import get_robot
    from ikflow_amd.model import MODEL_DESCRIPTIONS

    robot = get_robot(MODEL_DESCRIPTIONS[args.model]["robot_name"])
    hp = hparams_for(args.model)
    layout = layout_from(hp, robot)
    sd = random_state_dict(layout, robot, seed=0)
    solver = IKFlowSolver(hp, robot)
    solver.load_state_dict_tensors(sd)
    eng = solver.engine(dev)
    if args.gemm_variant >= 0:
        eng.set_gemm_variant(args.gemm_variant)
    if args.precision != "f32":
        solver.set_precision(args.precision)

    if args.million:
        args.batch = (1_000_000 + world - 1) // world
    B = args.batch
    # poses = FK(q), q ~ U(lo+eps, hi-eps), numpy default_rng(seed); latents N(0,1)
    q = torch.tensor(robot.sample_joint_angles(B, 0.004363323129985824, np.random.default_rng(rank)), device=dev)
    poses = robot.forward_kinematics(q)
    latent = torch.randn(B, layout.dim, generator=torch.Generator().manual_seed(1 + rank)).to(dev)
    eng.reserve(B)
    # the one collective of the path: every rank's [B x ndof] solutions are all-gathered over RCCL/xGMI.  It runs on its
    # own HIP stream behind an event, so the gather of step i overlaps the flow of step i+1; the timed region ends with
    # both streams drained.
    n_buf = 2
    gathered = [torch.empty((world * B, layout.ndof), dtype=torch.float32, device=dev) for _ in range(n_buf)] if use_dist else None
    comm_stream = torch.cuda.Stream(dev) if use_dist else None
    state = {"i": 0, "keep": [None] * n_buf}

    def step():
        sol = solver.generate_ik_solutions(poses, latent=latent)
        if use_dist:
            k = state["i"] % n_buf
            state["i"] += 1
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            comm_stream.wait_event(ev)
            with torch.cuda.stream(comm_stream):
                dist.all_gather_into_tensor(gathered[k], sol)
            sol.record_stream(comm_stream)
            state["keep"][k] = sol
        return sol

    def fence():
        if use_dist:
            torch.cuda.current_stream(dev).wait_stream(comm_stream)
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sol = step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        last = gathered[(state["i"] - 1) % n_buf]
        assert torch.equal(last[rank * B : (rank + 1) * B], sol)  # own shard sits at its rank offset
    assert bool(torch.isfinite(sol).all())

    # dominant kernel: per-launch HIP-event timing (on the engine's stream) of every hidden-Linear contraction inside
    # a few more, otherwise identical, steps
    eng.profile_begin()
    gemm_layers = 2 * layout.nb_nodes * (layout.n_hidden - 1)
    chunks = (B + 16383) // 16384  # the engine processes a call in chunks of <= 16384 rows
    prof_steps = max(1, min(10, max(2, args.steps), 8000 // (gemm_layers * chunks)))  # the event pool holds 8192 pairs
    for _ in range(prof_steps):
        step()
    n_launch, tot_ms = eng.profile_end()
    gemm_ms = tot_ms / max(n_launch, 1)
    # every hidden layer processes all B rows of a step, in one launch (B <= 16384) or in 16384-row chunks
    flop_per_launch = prof_steps * gemm_layers * 2.0 * B * layout.width * layout.width / max(n_launch, 1)
    achieved = flop_per_launch / (gemm_ms * 1e-3) / 1e12
    value = world * B * args.steps / elapsed
    flow_tflops = value / world * layout.flops_per_solution() / 1e12

    traffic, traffic_src = pmc_traffic_per_launch() if args.batch == 4096 else (None, None)
    extra = {"flow_tflops_per_gpu": round(flow_tflops, 2), "gemm_ms": round(gemm_ms, 5),
             "gemm_launches_per_step": 2 * layout.nb_nodes * (layout.n_hidden - 1)}
    if args.precision == "f32" and not args.no_split_extra:
        # the same workload with the hidden contractions on the error-compensated 3x f16 MFMA split (opt-in precision mode;
        # measured closer to the fp64 twin than the f32 MFMA path - tests/test_gpu_parity.py::test_flow_f16_split_*)
        solver.set_precision("f16x3")
        for _ in range(5):
            step()
        fence()
        t1 = time.perf_counter()
        n2 = max(5, args.steps)
        for _ in range(n2):
            sol2 = step()
        fence()
        dt2 = time.perf_counter() - t1
        if use_dist:
            tt = torch.tensor([dt2], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt2 = float(tt.item())
        eng.profile_begin()
        for _ in range(5):
            step()
        nl2, ms2 = eng.profile_end()
        solver.set_precision("f32")
        extra["f16x3_split"] = {
            "value": world * B * n2 / dt2, "unit": "IK solutions/s", "ms_per_step": 1000.0 * dt2 / n2,
            "kernel": "k_split_gemm_dma", "avg_launch_ms": ms2 / max(nl2, 1),
            "max_abs_diff_vs_f32_path": float((sol2 - sol).abs().max().item()),
            "note": "opt-in IKFlowSolver.set_precision('f16x3'): a = hi + lo/2048 operand split, 3 v_mfma_f32_32x32x16_f16 per 16 k, fp32 accumulate",
        }
    if args.extras and rank == 0:
        extra.update(run_extras(solver, eng, robot, layout, dev))

    out = {
        "metric": f"IK solutions/sec (approx), {'Panda 12-node' if 'panda' in args.model else args.model} flow, batch {B} per GPU",
        "value": value,
        "unit": "IK solutions/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.precision == "f32" else "f16x3 (error-compensated f16 MFMA, f32 accumulate)",
        "data": "synthetic (seeded random weights of the released architecture; poses = FK(uniform q); N(0,1) latents)",
        "config": {"workload": f"{args.model} generate_ik_solutions, B={B} poses per GPU per step, clamp_to_joint_limits",
                   "global_batch": world * B, "parallelism": f"rows sharded x{world}, weights replicated, 1 all-gather/step"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                     "traffic_source": traffic_src, "kernel": eng.dominant_kernel_name(),
                     "flop_per_launch": flop_per_launch, "avg_launch_ms": gemm_ms,
                     "timing": f"hipEvent pair per launch on the engine stream, {n_launch} launches over extra steps; an empty pair "
                               f"({eng.last_event_overhead_ms * 1e3:.2f} us, calibrated on the same stream) is subtracted"},
        "extra": extra,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, layout, robot.actuated_joints_limits, poses.cpu(), latent.cpu(), args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def run_extras(solver, eng, robot, layout, dev):
    """Secondary numbers (not the headline): B=512 approx and exact IK at B=4096."""
    res = {}
    for b in (512,):
        q = torch.tensor(robot.sample_joint_angles(b, 0.004363323129985824, np.random.default_rng(5)), device=dev)
        poses = robot.forward_kinematics(q)
        lat = torch.randn(b, layout.dim, device=dev)
        for _ in range(5):
            solver.generate_ik_solutions(poses, latent=lat)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        k = 50
        for _ in range(k):
            solver.generate_ik_solutions(poses, latent=lat)
        torch.cuda.synchronize(dev)
        res[f"approx_B{b}_solutions_per_s"] = b * k / (time.perf_counter() - t0)
    b = 4096
    q = torch.tensor(robot.sample_joint_angles(b, 0.004363323129985824, np.random.default_rng(6)), device=dev)
    poses = robot.forward_kinematics(q)
    for _ in range(2):
        solver.generate_exact_ik_solutions(poses, pos_error_threshold=1e-3, rot_error_threshold=0.01)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    k = 5
    nvalid = 0
    for _ in range(k):
        _, valid = solver.generate_exact_ik_solutions(poses, pos_error_threshold=1e-3, rot_error_threshold=0.01)
        nvalid += int(valid.sum().item())
    dt = time.perf_counter() - t0
    res["exact_B4096_target_poses_per_s"] = b * k / dt
    res["exact_B4096_valid_per_s"] = nvalid / dt
    res["exact_note"] = "random weights: flow seeds are uninformative, so valid/s measures arithmetic cost, not convergence"
    return res


if __name__ == "__main__":
    main()
