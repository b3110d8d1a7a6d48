"""Experiment: a small batch as S independent sub-batches on S HIP streams (S engine handles), so that one chain's kernel
boundaries / prologues / epilogues overlap another chain's MFMA work on the same CUs.  usage: dual_stream_small.py [f32|f16x3]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import Panda

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
dev = torch.device("cuda:0")
robot = Panda(); hp = hparams_for("panda__full__lp191_5.25m"); lay = layout_from(hp, robot)
sd = random_state_dict(lay, robot, 0)
solvers = []
for _ in range(4):
    s = IKFlowSolver(hp, robot); s.load_state_dict_tensors(sd); s.set_precision(prec); s.engine(dev); solvers.append(s)
streams = [torch.cuda.Stream(dev) for _ in range(4)]

def run(B, S, steps=40):
    q = torch.tensor(robot.sample_joint_angles(B, 0.0043, np.random.default_rng(0)), device=dev)
    poses = robot.forward_kinematics(q); lat = torch.randn(B, 7, device=dev)
    h = B // S
    ps = [poses[i*h:(i+1)*h].contiguous() for i in range(S)]
    ls = [lat[i*h:(i+1)*h].contiguous() for i in range(S)]
    def step():
        if S == 1:
            return solvers[0].generate_ik_solutions(ps[0], latent=ls[0])
        for i in range(S):
            with torch.cuda.stream(streams[i]):
                solvers[i].generate_ik_solutions(ps[i], latent=ls[i])
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    return dt * 1e3

for B in (256, 512, 1024, 2048, 4096):
    row = [f"{prec} B={B:5d}:"]
    for S in (1, 2, 4):
        if B // S >= 64:
            row.append(f"S={S}: {run(B, S):.3f} ms")
    print("  ".join(row), flush=True)
