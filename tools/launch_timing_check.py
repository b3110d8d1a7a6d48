"""Per-launch duration of the B = 4096 row-owner launch after an idle gap (one stream synchronisation) against steady state: is the first
launch behind a gap slower (clock ramp), and for how many launches?  usage: PYTHONPATH=. python tools/launch_timing_check.py"""
import time

import numpy as np
import torch

from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import get_robot

MODEL = "panda__full__lp191_5.25m"
dev = torch.device("cuda", 0)
robot = get_robot("panda")
hp = hparams_for(MODEL)
layout = layout_from(hp, robot)
solver = IKFlowSolver(hp, robot)
solver.load_state_dict_tensors(random_state_dict(layout, robot, seed=0))
B = 4096
q = torch.tensor(robot.sample_joint_angles(B, 0.004, np.random.default_rng(0)), device=dev)
p = robot.forward_kinematics(q)
l = torch.randn(B, layout.dim, generator=torch.Generator().manual_seed(1)).to(dev)


def pairs(n):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record()
        solver.generate_ik_solutions(p, latent=l)
        b.record()
    torch.cuda.synchronize()
    return [round(a.elapsed_time(b), 3) for a, b in evs]


for _ in range(10):
    solver.generate_ik_solutions(p, latent=l)
torch.cuda.synchronize()
for gap_ms in (0, 1, 10, 100, 1000):
    time.sleep(gap_ms * 1e-3)
    print(f"after a {gap_ms} ms idle gap:", pairs(12))
