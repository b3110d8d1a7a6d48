"""Repeat identical calls and require identical bits: approximate IK at row counts on every tile path (both precisions), seeded exact IK,
and calls interleaved across sizes (buffers are reused between sizes).  python tools/determinism_soak.py [repeats]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from helpers import panda_model, reachable_poses, latents
from ikflow_amd.ikflow_solver import IKFlowSolver

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = "cuda:0"
robot, hp, lay, sd = panda_model()
s = IKFlowSolver(hp, robot); s.load_state_dict_tensors(sd)
sizes = [1, 31, 128, 256, 300, 512, 700, 1024, 1500, 2048, 3000, 4096, 5000, 20000]
inputs = {}
for n in sizes:
    _, p = reachable_poses(robot, n, n); inputs[n] = (p.to(dev), latents(n, lay.dim, n + 1).to(dev))
bad = 0
for prec in ("f32", "f16x3"):
    s.set_precision(prec)
    first = {n: s.generate_ik_solutions(inputs[n][0], n=(1 if n == 1 else None), latent=inputs[n][1]).clone() for n in sizes}
    for r in range(reps):
        for n in (sizes if r % 2 == 0 else sizes[::-1]):
            out = s.generate_ik_solutions(inputs[n][0], n=(1 if n == 1 else None), latent=inputs[n][1])
            if not torch.equal(out, first[n]):
                bad += 1
                print("MISMATCH", prec, n, r, float((out - first[n]).abs().max()))
    print(prec, "approx:", reps, "x", len(sizes), "sizes identical" if bad == 0 else "MISMATCHES")
s.set_precision("f32")
eng = s.engine(dev)
n = 3000
q_true, poses = reachable_poses(robot, n, 5)
g = torch.Generator().manual_seed(6)
lat = [torch.randn(n * r, lay.dim, generator=g).to(dev) for r in (1, 3, 10)]
ref = None
for r in range(max(4, reps // 5)):
    sol, valid = eng.generate_exact(poses.to(dev), (1, 3, 10), 5e-2, 0.5, latents=lat)
    if ref is None: ref = (sol.clone(), valid.clone())
    elif not (torch.equal(sol, ref[0]) and torch.equal(valid, ref[1])):
        bad += 1; print("EXACT MISMATCH", r)
print("exact IK valid", int(ref[1].sum()), "of", n, "- identical" if bad == 0 else "- MISMATCHES")
sys.exit(1 if bad else 0)
