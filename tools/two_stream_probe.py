"""Does running two independent half batches on two streams (two handles) hide each other's prologues / epilogues / entry launches?
  python tools/two_stream_probe.py [B=4096]
Prints ms for: one handle B rows; two handles B/2 rows each on two streams (total rows B); one handle B/2 rows alone."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import Panda
dev = torch.device("cuda:0")
robot = Panda(); hp = hparams_for("panda__full__lp191_5.25m"); lay = layout_from(hp, robot)
sd = random_state_dict(lay, robot, 0)
def mk():
    s = IKFlowSolver(hp, robot); s.load_state_dict_tensors(sd); return s.engine(dev)
e0, e1 = mk(), mk()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
splits = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0.5]
poses = torch.randn(B, 7, device=dev); poses[:, 3:] /= poses[:, 3:].norm(dim=1, keepdim=True)
lat = torch.randn(B, 7, device=dev)
s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def one(n, steps=40):
    p, l = poses[:n], lat[:n]
    for _ in range(5): e0.generate_approx(p, l, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): e0.generate_approx(p, l, True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
def two(n0, steps=40):
    pa, la, pb, lb = poses[:n0], lat[:n0], poses[n0:], lat[n0:]
    def step():
        with torch.cuda.stream(s0): e0.generate_approx(pa, la, True)
        with torch.cuda.stream(s1): e1.generate_approx(pb, lb, True)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
for rep in range(3):
    print(f"B={B}: one handle {one(B):.4f} ms   half alone {one(B // 2):.4f} ms   " +
          "   ".join(f"two streams split {f:.2f}: {two(int(B * f) // 128 * 128):.4f} ms" for f in splits), flush=True)
