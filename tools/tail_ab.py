"""A/B of the in-launch entry phase (TailSync, ikf_set_gemm_variant 121) against entry launches (120): ms per approximate-IK
call of the Panda model at the batch sizes that take it.  `python tools/tail_ab.py [sizes]`"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import Panda

dev = torch.device("cuda:0")
robot = Panda()
hp = hparams_for("panda__full__lp191_5.25m")
lay = layout_from(hp, robot)
s = IKFlowSolver(hp, robot)
s.library_flavour = "probes"  # the forms measured here live in lib/libikflow_amd_probes.so (python -m ikflow_amd.build --probes)
s.load_state_dict_tensors(random_state_dict(lay, robot, 0))
eng = s.engine(dev)


def t(B, variant, steps):
    eng.set_gemm_variant(variant)
    poses = torch.randn(B, 7, device=dev)
    poses[:, 3:] /= poses[:, 3:].norm(dim=1, keepdim=True)
    lat = torch.randn(B, 7, device=dev)
    for _ in range(10):
        eng.generate_approx(poses, lat, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.generate_approx(poses, lat, True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [288, 384, 512, 4096]
for rep in range(3):
    for B in sizes:
        steps = 200 if B <= 512 else 60
        a, b = t(B, 120, steps), t(B, 121, steps)
        print(f"B={B}: entry launches {a:.4f} ms   in-launch entry phase {b:.4f} ms   ({100 * (b / a - 1):+.1f} %)", flush=True)
