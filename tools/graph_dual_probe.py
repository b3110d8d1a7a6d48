"""Experiment: hipGraph with S parallel branches, each the launch chain of one sub-batch (host launch cost out of the way)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import Panda

dev = torch.device("cuda:0")
robot = Panda(); hp = hparams_for("panda__full__lp191_5.25m"); lay = layout_from(hp, robot)
sd = random_state_dict(lay, robot, 0)
engs = []
for _ in range(4):
    s = IKFlowSolver(hp, robot); s.load_state_dict_tensors(sd); engs.append(s.engine(dev))
side = [torch.cuda.Stream(dev) for _ in range(4)]

def bench(B, S):
    poses = torch.randn(B, 7, device=dev); poses[:, 3:] /= poses[:, 3:].norm(dim=1, keepdim=True)
    lat = torch.randn(B, 7, device=dev)
    h = B // S
    ps = [poses[i*h:(i+1)*h].contiguous() for i in range(S)]
    ls = [lat[i*h:(i+1)*h].contiguous() for i in range(S)]
    for e in engs[:S]: e.reserve(h)
    main = side[0]
    def run():
        outs = [None] * S
        for i in range(1, S):
            side[i].wait_stream(main)
            with torch.cuda.stream(side[i]):
                outs[i] = engs[i].generate_approx(ps[i], ls[i], True)
        with torch.cuda.stream(main):
            outs[0] = engs[0].generate_approx(ps[0], ls[0], True)
        for i in range(1, S):
            main.wait_stream(side[i])
        return outs
    main.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(main):
        for _ in range(3): run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main):
        outs = run()
    torch.cuda.synchronize()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 50 * 1e3

for B in (256, 512, 1024, 2048, 4096):
    row = [f"B={B:5d}:"]
    for S in (1, 2, 4):
        if B // S >= 64:
            try:
                row.append(f"S={S}: {bench(B, S):.3f} ms")
            except Exception as e:
                row.append(f"S={S}: ERR {type(e).__name__} {str(e)[:80]}")
    print("  ".join(row), flush=True)
