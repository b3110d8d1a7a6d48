"""(r06 debugging aid) One row of tools/lm_precision_report.py taken apart on the GPU box: the kernel's fp32 Jacobian and FK against the oracle's fp64 ones, and the LM
step of both arithmetics.  usage: python tools/lm_row_debug.py ROW [ROW ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from helpers import panda_model, reachable_poses
from ikflow_amd.ikflow_solver import IKFlowSolver
from oracle import kinematics_oracle as ko

rows = [int(a) for a in sys.argv[1:]] or [72]
robot, hp, lay, sd = panda_model()
eng = IKFlowSolver(hp, robot).engine("cuda:0")
n = 4096
q_true, poses = reachable_poses(robot, n, 0)
seeds = ko.clamp_to_joint_limits(robot, q_true + 0.05 * torch.randn(q_true.shape, generator=torch.Generator().manual_seed(3)))
J64 = ko.jacobian(robot, seeds.double())
J32o = ko.jacobian(robot, seeds)
Jg = eng.jacobian(seeds.to("cuda:0")).cpu()
fk64 = ko.forward_kinematics(robot, seeds.double())
fkg = eng.forward_kinematics(seeds.to("cuda:0")).cpu()
e64 = ko.pose_error_vector(robot, poses.double(), seeds.double())
out = {}
for mode in ("f32", "f64"):
    eng.set_lm_precision(mode)
    out[mode] = eng.lm_step(poses.to("cuda:0"), seeds.to("cuda:0")).cpu()
eng.set_lm_precision("f64")
ref64 = ko.lm_step(robot, poses.double(), seeds.double())
ref32 = ko.lm_step(robot, poses, seeds)
for i in rows:
    A = J64[i].T @ J64[i] + 1e-4 * torch.eye(7, dtype=torch.float64)
    print(json.dumps({"row": i, "seed": seeds[i].tolist(), "eig": torch.linalg.eigvalsh(A).tolist(),
                      "J_gpu_minus_J64_maxabs": float((Jg[i].double() - J64[i]).abs().max()), "J_oracle32_minus_J64_maxabs": float((J32o[i].double() - J64[i]).abs().max()),
                      "fk_gpu_minus_fk64": (fkg[i].double() - fk64[i]).tolist(), "e64": e64[i].tolist(),
                      "step_hip32_minus_truth": (out["f32"][i].double() - ref64[i]).tolist(), "step_hip64_minus_truth": (out["f64"][i].double() - ref64[i]).tolist(),
                      "step_oracle32_minus_truth": (ref32[i].double() - ref64[i]).tolist()}))
