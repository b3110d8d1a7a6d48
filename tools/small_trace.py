"""In-kernel timeline of the 16-row small-batch kernels (probe build: IKF_HIPCC_FLAGS_FLOW_FUSED=-DIKF_TRACE python -m ikflow_amd.build --force).
Stamps: contraction 10 start, 11 K loop done, 12 tail done; head 20 start, 21 loads issued, 22 pending coupling done, 23 first Linear done,
24 K loop done, 25 tail done.  Shader-clock cycles (2.39 GHz), median over the workgroups of the LAST launch of each kind in a call."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ikflow_amd import _lib
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import Panda
dev = torch.device("cuda:0")
robot = Panda(); hp = hparams_for("panda__full__lp191_5.25m"); lay = layout_from(hp, robot)
s = IKFlowSolver(hp, robot); s.library_flavour = "probes"; s.load_state_dict_tensors(random_state_dict(lay, robot, 0)); eng = s.engine(dev)
lib = _lib.load("probes")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for code in (sys.argv[2].split(",") if len(sys.argv) > 2 else []):  # engine variants, e.g. 171 = the one-launch chain (stamps of its LAST subnet)
    eng.set_gemm_variant(int(code))
nb = 4096
buf = torch.zeros(nb * 64, dtype=torch.int64, device=dev)
poses = torch.randn(B, 7, device=dev); poses[:, 3:] /= poses[:, 3:].norm(dim=1, keepdim=True)
lat = torch.randn(B, 7, device=dev)
for _ in range(3): eng.generate_approx(poses, lat, True)
torch.cuda.synchronize()
assert lib.ikf_debug_set_trace(ctypes.c_void_p(buf.data_ptr())) == 0
eng.generate_approx(poses, lat, True)
torch.cuda.synchronize()
r = buf.cpu().numpy().reshape(nb, 64).astype(np.int64)
def show(name, keys, labels):
    nz = np.nonzero(r[:, keys[-1]])[0]
    print(f"B={B} {name}: {len(nz)} workgroups")
    for a, b, lab in zip(keys[:-1], keys[1:], labels):
        d = r[nz, b] - r[nz, a]
        print(f"   {lab:28s} {np.median(d):8.0f} cycles   (min {d.min()}, max {d.max()})")
    d = r[nz, keys[-1]] - r[nz, keys[0]]
    print(f"   {'total in kernel':28s} {np.median(d):8.0f} cycles = {np.median(d) / 2390:.2f} us")
show("head", [20, 21, 22, 23, 24, 25], ["issue loads", "pending coupling", "inputs + first Linear", "K loop", "barrier + tail"])
show("contraction", [10, 11, 12], ["operands + K loop", "tail"])
