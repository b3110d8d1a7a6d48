"""Does a launch wait for where its weights come from?  Marginal cost per coupling block of 1 / 2 / 3 / 12-node models (17 ... 203 MB of weights:
resident in the L2s from call to call, or streamed from the Infinity Cache / HBM) at 1 ... 512 rows.   python tools/l2_resident_check.py"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch
from helpers import custom_model
from ikflow_amd.ikflow_solver import IKFlowSolver
dev = torch.device("cuda:0")
res = {}
for nb in (1, 2, 3, 12):
    robot, hp, lay, sd = custom_model(seed=1, gain=1.0, nb_nodes=nb, dim=7, n_hidden=3, width=1024)
    s = IKFlowSolver(hp, robot); s.load_state_dict_tensors(sd); eng = s.engine(dev)
    for B in (1, 128, 256, 512):
        poses = torch.randn(B, 7, device=dev); poses[:, 3:] /= poses[:, 3:].norm(dim=1, keepdim=True)
        lat = torch.randn(B, 7, device=dev)
        for _ in range(50): eng.generate_approx(poses, lat, True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(500): eng.generate_approx(poses, lat, True)
        torch.cuda.synchronize(); res[(nb, B)] = (time.perf_counter() - t0) / 500 * 1e6
for B in (1, 128, 256, 512):
    print(f"B={B}: us per call: " + "  ".join(f"nb={nb}: {res[(nb, B)]:.1f}" for nb in (1, 2, 3, 12)) +
          f"   per subnet: nb=1 {res[(1, B)] / 2:.2f}, nb=2 {res[(2, B)] / 4:.2f}, nb=3 {res[(3,B)]/6:.2f}, marginal 3->12 {(res[(12, B)] - res[(3, B)]) / 18:.2f}")
