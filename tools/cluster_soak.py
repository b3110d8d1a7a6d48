"""Randomised soak of the cluster form's default hand-over (parity-tagged payload, no drain / epoch words / memset): thousands of calls of random
sizes 129 .. 3300 queued with hardly any synchronisation, every result compared with the same rows through the row-owner launch (no
inter-workgroup hand-over at all), plus a give-up injected now and then (ikf_set_gemm_variant 188).  usage: python tools/cluster_soak.py [calls=3000] [give_up_every=500]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import latents, panda_model, reachable_poses  # noqa: E402

from ikflow_amd.ikflow_solver import IKFlowSolver  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device("cuda:0")
robot, hp, lay, sd = panda_model(gain=2.0)
s = IKFlowSolver(hp, robot)
s.load_state_dict_tensors(sd)
eng = s.engine(dev)
rng = np.random.default_rng(0)
N = 3300
_, poses = reachable_poses(robot, N, 1)
P, L = poses.to(dev), latents(N, lay.dim, 2).to(dev)
eng.set_gemm_variant(182)
ref = s.generate_ik_solutions(P, latent=L).clone()     # rows are independent: any sub-batch must reproduce its rows of this
eng.set_gemm_variant(181)
torch.cuda.synchronize()
worst, checked, injected, t0 = 0.0, 0, 0, time.perf_counter()
pending = []
for c in range(calls):
    n = int(rng.integers(129, N + 1))
    lo = int(rng.integers(0, N - n + 1))
    if c % every == every // 2:
        eng.set_gemm_variant(188)      # the next cluster launch runs a workgroup short: waits run out, repair launch, pause, buffers re-created
        injected += 1
    out = s.generate_ik_solutions(P[lo:lo + n], latent=L[lo:lo + n])
    pending.append((out, lo, n))
    if len(pending) >= 64:
        for o, l, m in pending:
            worst = max(worst, float((o - ref[l:l + m]).abs().max()))
            checked += 1
        pending = []
for o, l, m in pending:
    worst = max(worst, float((o - ref[l:l + m]).abs().max()))
    checked += 1
torch.cuda.synchronize()
res = {"calls": calls, "checked": checked, "max_abs_diff_vs_row_owner_form": worst, "give_ups_injected": injected, "cluster_repairs": eng.cluster_repairs,
       "seconds": round(time.perf_counter() - t0, 1), "ok": worst <= 1e-5}
print(json.dumps(res))
sys.exit(0 if res["ok"] else 1)
