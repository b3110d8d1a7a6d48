"""Same-box A/B of builds of the library (box-to-box variation is ~2 %, more than many of the changes being judged):
  python tools/lib_ab.py 256,512,1024 tools/bin/ab/lib_a.so tools/bin/ab/lib_b.so ...     # "-" = the in-tree library
alternates the libraries in fresh processes, three rounds; prints ms per approximate-IK call (Panda) for each.  (tools/bin is git-ignored but
travels to the GPU box.)"""
import os, subprocess, sys
CHILD = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import ikflow_amd.build as B
if sys.argv[1] != "-":
    _p = os.path.abspath(sys.argv[1]); _orig = B.lib_path
    B.lib_path = lambda flavour="": _p if flavour == "" else _orig(flavour)
    B.is_stale = lambda flavour="": False
import torch
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import Panda
dev = torch.device("cuda:0")
robot = Panda(); hp = hparams_for("panda__full__lp191_5.25m"); lay = layout_from(hp, robot)
s = IKFlowSolver(hp, robot); s.load_state_dict_tensors(random_state_dict(lay, robot, 0)); eng = s.engine(dev)
out = []
for Bn in [int(x) for x in sys.argv[2].split(",")]:
    poses = torch.randn(Bn, 7, device=dev); poses[:, 3:] /= poses[:, 3:].norm(dim=1, keepdim=True)
    lat = torch.randn(Bn, 7, device=dev)
    steps = 400 if Bn <= 2048 else 60
    for _ in range(60): eng.generate_approx(poses, lat, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): eng.generate_approx(poses, lat, True)
    torch.cuda.synchronize(); out.append(f"{Bn}: {(time.perf_counter() - t0) / steps * 1e3:.4f}")
print("   ".join(out))
'''
sizes, libs = sys.argv[1], sys.argv[2:]
for rnd in range(3):
    for path in libs:
        r = subprocess.run([sys.executable, "-c", CHILD, path, sizes], capture_output=True, text=True)
        print(f"{os.path.basename(path):>16}", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-600:], flush=True)
