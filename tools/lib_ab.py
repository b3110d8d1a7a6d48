"""Same-box A/B of two builds of the library (box-to-box variation is ~2 %, more than many of the changes being judged):
  cp ikflow_amd/lib/libikflow_amd.so tools/bin/lib_base.so   # before the change; tools/bin is git-ignored but travels to the GPU box
  python tools/lib_ab.py tools/bin/lib_base.so 1,16,64,128   # alternates base / current in fresh processes, three rounds
Prints ms per approximate-IK call (Panda) for each."""
import os, subprocess, sys
CHILD = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import ikflow_amd._lib as L
if sys.argv[1] != "-": L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import Panda
dev = torch.device("cuda:0")
robot = Panda(); hp = hparams_for("panda__full__lp191_5.25m"); lay = layout_from(hp, robot)
s = IKFlowSolver(hp, robot); s.load_state_dict_tensors(random_state_dict(lay, robot, 0)); eng = s.engine(dev)
out = []
for B in [int(x) for x in sys.argv[2].split(",")]:
    poses = torch.randn(B, 7, device=dev); poses[:, 3:] /= poses[:, 3:].norm(dim=1, keepdim=True)
    lat = torch.randn(B, 7, device=dev)
    steps = 300 if B <= 512 else 60
    for _ in range(20): eng.generate_approx(poses, lat, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): eng.generate_approx(poses, lat, True)
    torch.cuda.synchronize(); out.append(f"{B}: {(time.perf_counter() - t0) / steps * 1e3:.4f}")
print("   ".join(out))
'''
base, sizes = sys.argv[1], sys.argv[2]
for rnd in range(3):
    for name, path in (("base", base), ("new ", "-")):
        r = subprocess.run([sys.executable, "-c", CHILD, path, sizes], capture_output=True, text=True)
        print(name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
