"""Does a hipGraph replay of the approx-IK launch chain beat stream launches?  (probe; GPU box only)

Captures Engine.generate_approx on a side stream with torch.cuda.CUDAGraph (the library only enqueues kernels on the
caller's stream once scratch is reserved) and times direct launches against graph replays."""
import sys, time
import torch
from ikflow_amd.model import random_state_dict
from ikflow_amd.robots import get_robot

from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import MODEL_DESCRIPTIONS, hparams_for, layout_from

name = "panda__full__lp191_5.25m"
robot = get_robot(MODEL_DESCRIPTIONS[name]["robot_name"])
hp = hparams_for(name)
layout = layout_from(hp, robot)
solver = IKFlowSolver(hp, robot)
solver.load_state_dict_tensors(random_state_dict(layout, robot, seed=0))
eng = solver.engine(torch.device("cuda", 0))
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
eng.set_precision(prec)
for B in (1, 128, 256, 512, 1024, 4096):
    eng.reserve(B)
    poses = torch.randn(B, 7, device="cuda"); poses[:, 3:] /= poses[:, 3:].norm(dim=1, keepdim=True)
    lat = torch.randn(B, layout.dim, device="cuda")
    def run():
        return eng.generate_approx(poses, lat, True)
    for _ in range(5): out = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): out = run()
    torch.cuda.synchronize()
    direct = (time.perf_counter() - t0) / 50 * 1e3
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            gout = run()
    torch.cuda.synchronize()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 50 * 1e3
    ok = torch.equal(gout, out)
    print(f"{prec} B={B:5d}: direct {direct:.3f} ms  graph {graph:.3f} ms  same={ok}", flush=True)
