"""In-chain step time of generate_approx for every tile configuration x batch size x precision (GPU box only).
The picker rules in flow_fused.hip / flow_split.hip (fused_pick_cfg, split_pick_cfg) are derived from this table."""
import sys, time, json
import torch
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import MODEL_DESCRIPTIONS, hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import get_robot

name = "panda__full__lp191_5.25m"
robot = get_robot(MODEL_DESCRIPTIONS[name]["robot_name"])
hp = hparams_for(name)
layout = layout_from(hp, robot)
solver = IKFlowSolver(hp, robot)
solver.load_state_dict_tensors(random_state_dict(layout, robot, seed=0))
eng = solver.engine(torch.device("cuda", 0))
batches = [int(b) for b in sys.argv[1].split(",")] if len(sys.argv) > 1 else [128, 256, 512, 768, 1024, 1536, 2048, 2560, 3072, 4096, 6144, 8192]
eng.reserve(max(batches))
rows = []
for prec in ("f32", "f16x3"):
    eng.set_precision(prec)
    for B in batches:
        poses = torch.randn(B, 7, device="cuda"); poses[:, 3:] /= poses[:, 3:].norm(dim=1, keepdim=True)
        lat = torch.randn(B, layout.dim, device="cuda")
        res = {}
        for variant in (-1, 101, 102, 103, 104, 105, 107):  # 101 + tile config (ikf_set_gemm_variant)
            if variant in (105, 107) and B > 1024:
                continue  # configs 4 / 6 = small-batch kernels (32x64 / 32x32 tiles)
            if variant == 104 and B > 2048:
                continue
            eng.set_gemm_variant(variant)
            for _ in range(3): eng.generate_approx(poses, lat, True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): eng.generate_approx(poses, lat, True)
            torch.cuda.synchronize()
            res["auto" if variant < 0 else f"cfg{variant - 101}"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
        eng.set_gemm_variant(-1)
        best = min((v, k) for k, v in res.items() if k != "auto")
        print(f"{prec:6s} B={B:5d}  " + "  ".join(f"{k}={v:.3f}" for k, v in res.items()) + f"   best={best[1]}", flush=True)
        rows.append({"precision": prec, "batch": B, "ms": res, "best": best[1]})
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        for r in rows: f.write(json.dumps(r) + "\n")
