#!/usr/bin/env python
"""Why does the torch-CPU baseline get SLOWER with more threads?  (VERDICT r02 weak #10.)

Times, per intra-op thread count, on this host:
  gemm      one [4096 x 1024].[1024 x 1024]^T addmm (the layer that is 99 % of the path's FLOPs), best of several repeats
  chain     the whole approximate-IK pass of the oracle (oracle/flow_oracle.py) over a B=4096 Panda batch
  small     the ~30 small ops between two GEMMs (cat / slice / atan / exp / mul on [4096 x <=16] tensors), per coupling block
and records the host topology (lscpu: sockets, cores per socket, NUMA nodes).  Writes one JSON document.

  python tools/cpu_thread_scaling.py [out.json]
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch


def best_of(fn, reps, inner=1):
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(inner):
            fn()
        best = min(best, (time.perf_counter() - t0) / inner)
    return best


def main():
    from helpers import latents, panda_model, reachable_poses
    from oracle import flow_oracle as fo

    robot, hp, lay, sd = panda_model()
    n = 4096
    _, poses = reachable_poses(robot, n, 0)
    lat = latents(n, lay.dim, 1)
    A = torch.randn(n, 1024)
    W = torch.randn(1024, 1024)
    b = torch.randn(1024)
    x = torch.randn(n, 7)
    c = torch.randn(n, 8)

    def small_ops():  # the non-GEMM part of one coupling half, as the reference graph issues it
        x1, x2 = x[:, :3], x[:, 3:]
        u = torch.cat([x1, c], dim=1)
        a = u[:, :8]
        s_, t_ = a[:, :4], a[:, 4:]
        s_ = 2.5 * 0.636 * torch.atan(s_)
        y2 = (x2 - t_) * torch.exp(-s_)
        return torch.cat([x1, y2], dim=1), s_.sum(dim=1)

    topo = {}
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=20).stdout
        for line in out.splitlines():
            k = line.split(":")[0].strip()
            if k in ("CPU(s)", "Thread(s) per core", "Core(s) per socket", "Socket(s)", "NUMA node(s)", "Model name", "L3 cache", "L2 cache"):
                topo[k] = line.split(":", 1)[1].strip()
    except Exception as e:
        topo["error"] = repr(e)
    default_threads = torch.get_num_threads()
    rows = []
    for th in [t for t in (1, 2, 4, 8, 16, 32, 64, 128, 256) if t <= max(default_threads, 1)] + ([default_threads] if default_threads not in (1, 2, 4, 8, 16, 32, 64, 128, 256) else []):
        torch.set_num_threads(th)
        t_gemm = best_of(lambda: torch.addmm(b, A, W.t()), 5 if th > 1 else 2, 3 if th > 1 else 1)
        t_small = best_of(small_ops, 5, 20)
        sub = slice(0, n if th > 1 else 512)
        t0 = time.perf_counter()
        fo.generate_ik_solutions_torch(sd, lay, robot, poses[sub], lat[sub])
        t_chain = time.perf_counter() - t0
        if th > 1:
            t_chain = best_of(lambda: fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat), 2)
        nr = sub.stop
        rows.append({"threads": th, "gemm_ms": 1e3 * t_gemm, "gemm_gflops": 2 * n * 1024 * 1024 / t_gemm / 1e9,
                     "small_ops_us_per_coupling_half": 1e6 * t_small, "chain_rows": nr, "chain_s": t_chain,
                     "chain_solutions_per_s": nr / t_chain,
                     "chain_share_of_48_gemms": 48 * t_gemm * (nr / n) / t_chain})
        print(json.dumps(rows[-1]), flush=True)
    # where the time goes at the best and at the default thread count: torch.profiler, self CPU time per operator over one pass
    ops = {}
    from torch.profiler import ProfilerActivity, profile

    for th in sorted({16, default_threads}):
        torch.set_num_threads(th)
        fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat)
        t0 = time.perf_counter()
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat)
        wall = time.perf_counter() - t0
        rows_p = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)[:8]
        ops[str(th)] = {"wall_s": wall, "top_self_cpu": [{"op": e.key, "calls": e.count, "self_ms": e.self_cpu_time_total / 1e3} for e in rows_p]}
        print(th, "threads:", wall, [(e.key, e.count, round(e.self_cpu_time_total / 1e3, 1)) for e in rows_p[:6]], flush=True)
    torch.set_num_threads(default_threads)
    doc = {"host": topo, "profiler_top_ops": ops, "torch_default_threads": default_threads, "torch_parallel_info": torch.__config__.parallel_info().splitlines()[:8],
           "rows": rows,
           "reading": "gemm_gflops peaks at a fraction of the cores and falls beyond it; chain_share_of_48_gemms shows how much of the "
                      "pass is GEMM time at that thread count - the remainder is the per-op overhead of the ~390 small ops, which grows "
                      "with the pool size"}
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "cpu_thread_scaling.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
