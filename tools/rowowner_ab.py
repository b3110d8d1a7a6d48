"""Row-owner launch (ikf_set_gemm_variant 182) and cluster form (187) against the per-layer kernels (180 + 185) and the default split by batch size.
usage: PYTHONPATH=. python tools/rowowner_ab.py [rows,rows,...] [out.jsonl]
Prints ms per call for each form - where the last partial round of a batch should switch from the per-layer kernels to the row-owner launch."""
import json
import sys
import time

import numpy as np
import torch

from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import MODEL_DESCRIPTIONS, hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import get_robot

MODEL = "panda__full__lp191_5.25m"
rows_list = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1024, 2048, 2560, 3072, 3328, 3584, 3840, 4096, 4352, 6144, 7168, 7680, 8192, 12288, 16384, 65536]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else None
dev = torch.device("cuda", 0)
robot = get_robot(MODEL_DESCRIPTIONS[MODEL]["robot_name"])
hp = hparams_for(MODEL)
layout = layout_from(hp, robot)
solver = IKFlowSolver(hp, robot)
solver.load_state_dict_tensors(random_state_dict(layout, robot, seed=0))
eng = solver.engine(dev)
for rows in rows_list:
    q = torch.tensor(robot.sample_joint_angles(rows, 0.004, np.random.default_rng(0)), device=dev)
    p = robot.forward_kinematics(q)
    l = torch.randn(rows, layout.dim, generator=torch.Generator().manual_seed(1)).to(dev)
    rec = {"rows": rows}
    for name, variants in (("per_layer", (180, 185)), ("row_owner", (182, 185)), ("cluster", (180, 187)), ("default", (181, 186))):
        if name == "cluster" and rows > 2048:
            continue
        if name == "row_owner" and rows < 2048:
            continue
        for variant in variants:
            eng.set_gemm_variant(variant)
        reps = max(5, min(50, int(200e3 / rows)))
        for _ in range(3):
            solver.generate_ik_solutions(p, n=(1 if rows == 1 else None), latent=l)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            solver.generate_ik_solutions(p, n=(1 if rows == 1 else None), latent=l)
        torch.cuda.synchronize(dev)
        rec[name + "_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
    eng.set_gemm_variant(181)
    eng.set_gemm_variant(186)
    rec["default_Msol_per_s"] = round(rows / rec["default_ms"] * 1e-3, 4)
    print(json.dumps(rec), flush=True)
    if out:
        out.write(json.dumps(rec) + "\n")
