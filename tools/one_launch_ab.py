import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import Panda
dev = torch.device("cuda:0")
robot = Panda(); hp = hparams_for("panda__full__lp191_5.25m"); lay = layout_from(hp, robot)
s = IKFlowSolver(hp, robot); s.load_state_dict_tensors(random_state_dict(lay, robot, 0)); eng = s.engine(dev)
def t(B, variant, steps=100):
    eng.set_gemm_variant(variant)
    poses = torch.randn(B, 7, device=dev); poses[:, 3:] /= poses[:, 3:].norm(dim=1, keepdim=True)
    lat = torch.randn(B, 7, device=dev)
    for _ in range(10): eng.generate_approx(poses, lat, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): eng.generate_approx(poses, lat, True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
for rep in range(2):
    for B in (288, 320, 384, 448, 512):
        print(f"B={B}: two-launch {t(B,110):.4f} ms   one-launch forced {t(B,112):.4f} ms", flush=True)
