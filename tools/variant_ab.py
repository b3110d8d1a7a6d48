"""A/B of engine variants (ikf_set_gemm_variant codes) on the Panda model: ms per approximate-IK call.
  python tools/variant_ab.py 130,131,132,133 512,1024,2048,4096   (a+b = several codes at once; the defaults are restored before every measurement)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import Panda
dev = torch.device("cuda:0")
robot = Panda(); hp = hparams_for("panda__full__lp191_5.25m"); lay = layout_from(hp, robot)
s = IKFlowSolver(hp, robot); s.library_flavour = "probes"; s.load_state_dict_tensors(random_state_dict(lay, robot, 0)); eng = s.engine(dev)
codes = [[int(y) for y in x.split("+")] for x in sys.argv[1].split(",")]  # "160+110" = both codes set
sizes = [int(x) for x in sys.argv[2].split(",")]
def t(B, variant, steps):
    for c in (100, 111, 134, 120, 151, 153, 159, 162, 170):  # defaults
        eng.set_gemm_variant(c)
    for c in variant:
        eng.set_gemm_variant(c)
    poses = torch.randn(B, 7, device=dev); poses[:, 3:] /= poses[:, 3:].norm(dim=1, keepdim=True)
    lat = torch.randn(B, 7, device=dev)
    for _ in range(10): eng.generate_approx(poses, lat, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): eng.generate_approx(poses, lat, True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
for rep in range(3):
    for B in sizes:
        steps = 200 if B <= 512 else 60
        print(f"B={B}: " + "   ".join(f"[{'+'.join(str(x) for x in c)}] {t(B, c, steps):.4f} ms" for c in codes), flush=True)
