"""Latency of ONE synchronised call at small batch: IKFlowSolver.generate_ik_solutions (Python shim) against Engine.generate_approx against
the C-ABI call alone, and the CPU time the shim spends before the first kernel is enqueued.   python tools/shim_overhead.py"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import torch
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import Panda
dev = torch.device("cuda:0")
robot = Panda(); hp = hparams_for("panda__full__lp191_5.25m"); lay = layout_from(hp, robot)
s = IKFlowSolver(hp, robot); s.load_state_dict_tensors(random_state_dict(lay, robot, 0)); eng = s.engine(dev)
def lat_sync(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def lat_async(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e3, (time.perf_counter() - t0) / n * 1e3
for B in (1, 16, 128):
    poses = torch.randn(B, 7, device=dev); poses[:, 3:] /= poses[:, 3:].norm(dim=1, keepdim=True)
    lat = torch.randn(B, 7, device=dev)
    f_solver_lat = lambda: s.generate_ik_solutions(poses, n=(1 if B == 1 else None), latent=lat)
    f_solver = lambda: s.generate_ik_solutions(poses if B > 1 else poses[0], n=B)
    f_eng = lambda: eng.generate_approx(poses, lat, True)
    print(f"B={B}: synchronised call: solver (latent drawn) {lat_sync(f_solver):.4f} ms   solver (latent given) {lat_sync(f_solver_lat):.4f} ms   engine {lat_sync(f_eng):.4f} ms")
    a = lat_async(f_solver); b = lat_async(f_solver_lat); c = lat_async(f_eng)
    print(f"       back to back: CPU enqueue time / wall per call: solver (drawn) {a[0]:.4f} / {a[1]:.4f}   solver (given) {b[0]:.4f} / {b[1]:.4f}   engine {c[0]:.4f} / {c[1]:.4f}")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): s.generate_ik_solutions(poses[0], n=16)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
