"""How long a cluster launch waits for a peer that never arrives (ikf_set_gemm_variant 188: the launch runs one workgroup short): wall time of the
call that gives up - the wait + the repair launch (2.8 ms) - per form.  usage: python tools/cluster_wait_probe.py [out.json]"""
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
res = {}
for n, variants in ((100, ()), (256, ()), (512, ()), (1024, ()), (2048, ()), (512, (192,)), (512, (189,))):
    s = IKFlowSolver(hp, robot)
    s.load_state_dict_tensors(sd)
    eng = s.engine("cuda:0")
    for v in variants:
        eng.set_gemm_variant(v)
    _, poses = reachable_poses(robot, n, 1)
    P, L = poses.to("cuda:0"), latents(n, lay.dim, 2).to("cuda:0")
    for _ in range(20):
        s.generate_ik_solutions(P, latent=L)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.generate_ik_solutions(P, latent=L)
    torch.cuda.synchronize()
    clean = (time.perf_counter() - t0) * 1e3
    plan = eng.plan(n)
    eng.set_gemm_variant(188)
    t0 = time.perf_counter()
    s.generate_ik_solutions(P, latent=L)
    torch.cuda.synchronize()
    gave_up = (time.perf_counter() - t0) * 1e3
    res[f"{n}{'_v' + '_'.join(map(str, variants)) if variants else ''}"] = {"plan": plan, "clean_ms": round(clean, 3), "call_that_gave_up_ms": round(gave_up, 3), "repairs": eng.cluster_repairs}
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
