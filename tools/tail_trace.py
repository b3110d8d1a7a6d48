"""In-kernel timeline of the fused tail (probe build: IKF_HIPCC_FLAGS_FLOW_FUSED=-DIKF_TRACE python -m ikflow_amd.build --force).
Stamps (shader clock, thread 0 of each workgroup): 40 K loop done, 42 partial-sum stores issued, 43/44 drain + arrive,
45 prefetches issued, 46 siblings arrived, 47 pending coupling done, 48 input rows built, 49 first-Linear stores issued, 50 drained."""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = 4096
buf = torch.zeros(nb * 64, dtype=torch.int64, device=dev)
poses = torch.randn(B, 7, device=dev); poses[:, 3:] /= poses[:, 3:].norm(dim=1, keepdim=True)
lat = torch.randn(B, 7, device=dev)
eng.set_gemm_variant(121)
for _ in range(3): eng.generate_approx(poses, lat, True)
torch.cuda.synchronize()
assert lib.ikf_debug_set_trace(ctypes.c_void_p(buf.data_ptr())) == 0
eng.generate_approx(poses, lat, True)
torch.cuda.synchronize()
r = buf.cpu().numpy().reshape(nb, 64)
nz = np.nonzero(r[:, 49])[0]
print(f"B={B}: {len(nz)} workgroups stamped the tail")
names = {42: "P stores issued", 43: "tail entered", 44: "drained+arrived", 45: "prefetch issued", 46: "siblings arrived", 47: "coupling done", 48: "inputs built", 49: "h stores issued", 50: "h stores drained"}
base = r[nz, 43].astype(np.int64)
for k in (44, 45, 46, 47, 48, 49, 50):
    d = r[nz, k].astype(np.int64) - base
    print(f"  {names[k]:18s} +{np.median(d):8.0f} cycles median   (min {d.min()}, max {d.max()})")
