"""Can an approximate-IK call be captured into a HIP graph (torch.cuda.CUDAGraph) after ikf_reserve, and does the replay give the call's bits?
Sizes: 128 / 512 / 4096 / 5000 rows (cluster32, cluster8, row-owner, row-owner + tail).  Prints one JSON line per size."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from helpers import latents, panda_model, reachable_poses
from ikflow_amd.ikflow_solver import IKFlowSolver

robot, hp, lay, sd = panda_model()
s = IKFlowSolver(hp, robot)
s.load_state_dict_tensors(sd)
eng = s.engine("cuda:0")
eng.reserve(8192)
for n in (128, 512, 4096, 5000):
    _, poses = reachable_poses(robot, n, 1)
    P, L = poses.to("cuda:0"), latents(n, lay.dim, 2).to("cuda:0")
    ref = s.generate_ik_solutions(P, latent=L).clone()
    rec = {"rows": n, "plan": eng.plan(n)}
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                s.generate_ik_solutions(P, latent=L)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = s.generate_ik_solutions(P, latent=L)
        with torch.inference_mode():
            out.zero_()
        g.replay()
        torch.cuda.synchronize()
        rec["replay_equals_call"] = bool(torch.equal(out, ref))
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            g.replay()
        torch.cuda.synchronize()
        rec["replay_ms"] = round((time.perf_counter() - t0) / 200 * 1e3, 4)
        for _ in range(20):
            s.generate_ik_solutions(P, latent=L)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            s.generate_ik_solutions(P, latent=L)
        torch.cuda.synchronize()
        rec["call_ms"] = round((time.perf_counter() - t0) / 200 * 1e3, 4)
        rec["cluster_repairs"] = eng.cluster_repairs
    except Exception as e:  # noqa: BLE001
        rec["error"] = f"{type(e).__name__}: {str(e)[:300]}"
    print(json.dumps(rec), flush=True)
