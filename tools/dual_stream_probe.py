"""Experiment: one B=4096 step as two independent half-batches on two HIP streams (two engine handles), so that one
chain's kernel boundaries / prologues / epilogues overlap the other chain's MFMA work on the same CUs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import Panda

dev = torch.device("cuda:0")
robot = Panda(); hp = hparams_for("panda__full__lp191_5.25m"); lay = layout_from(hp, robot)
sd = random_state_dict(lay, robot, 0)
B = 4096
q = torch.tensor(robot.sample_joint_angles(B, 0.0043, np.random.default_rng(0)), device=dev)
poses = robot.forward_kinematics(q); lat = torch.randn(B, 7, device=dev)

def make(variant):
    s = IKFlowSolver(hp, robot); s.load_state_dict_tensors(sd); e = s.engine(dev)
    if variant >= 0: e.set_gemm_variant(variant)
    return s

def run_single(variant, steps=30):
    s = make(variant)
    for _ in range(5): s.generate_ik_solutions(poses, latent=lat)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): s.generate_ik_solutions(poses, latent=lat)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return B * steps / dt, dt / steps * 1e3

def run_dual(variant, nsplit=2, steps=30):
    ss = [make(variant) for _ in range(nsplit)]
    streams = [torch.cuda.Stream(dev) for _ in range(nsplit)]
    h = B // nsplit
    ps = [poses[i*h:(i+1)*h].contiguous() for i in range(nsplit)]
    ls = [lat[i*h:(i+1)*h].contiguous() for i in range(nsplit)]
    def step():
        for i in range(nsplit):
            with torch.cuda.stream(streams[i]):
                ss[i].generate_ik_solutions(ps[i], latent=ls[i])
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return B * steps / dt, dt / steps * 1e3

for v in [int(a) for a in sys.argv[1:]] or [0, 1, 6, 7]:
    print(f"single stream  variant {v}: %.0f sol/s  %.3f ms/step" % run_single(v), flush=True)
for v, ns in [(5, 2), (4, 2), (2, 2), (8, 2), (5, 4), (4, 4)]:
    print(f"{ns} streams x {B//ns} rows variant {v}: %.0f sol/s  %.3f ms/step" % run_dual(v, ns), flush=True)
