"""In-kernel timeline of the large-tile contraction k_flow_gemm (probe build: IKF_HIPCC_FLAGS_FLOW_FUSED=-DIKF_TRACE python -m ikflow_amd.build --force).
Stamps (shader clock, thread 0 of each workgroup, the LAST contraction of the call): 0 kernel start, 1 prologue done (tile 0 in LDS),
2 + i after K tile 4 i + 3, 40 K loop done, 41 epilogue done.   python tools/gemm_trace.py [B=4096]"""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ikflow_amd import _lib
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import Panda
dev = torch.device("cuda:0")
robot = Panda(); hp = hparams_for("panda__full__lp191_5.25m"); lay = layout_from(hp, robot)
s = IKFlowSolver(hp, robot); s.load_state_dict_tensors(random_state_dict(lay, robot, 0)); eng = s.engine(dev)
lib = ctypes.CDLL(_lib.LIB_PATH)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
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
nz = np.nonzero(r[:, 41])[0]
print(f"B={B}: {len(nz)} workgroups")
def seg(a, b, lab):
    d = r[nz, b] - r[nz, a]
    print(f"   {lab:30s} {np.median(d):8.0f} cycles   (min {d.min()}, max {d.max()})")
seg(0, 1, "prologue")
last = 1
for i in range(8):
    if (r[nz, 2 + i] != 0).all():
        seg(last, 2 + i, f"K tiles .. {4 * i + 3}")
        last = 2 + i
seg(last, 40, "rest of the K loop")
seg(40, 41, "epilogue")
seg(0, 41, "total")
t0 = r[nz, 0]
print(f"   start skew over the workgroups: {t0.max() - t0.min()} cycles; end skew {r[nz, 41].max() - r[nz, 41].min()}")
