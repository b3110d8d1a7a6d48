"""Accuracy evidence for both contraction precisions: HIP f32-MFMA path, HIP f16x3 split path and the torch-CPU fp32 oracle,
each against the numpy fp64 twin, on the full Panda and FetchArm architectures (seeded weights; `gain` scales the last
Linear of every subnet so the coupling coefficients reach O(1) like a trained model's).  Prints one JSON line per case."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import fetch_arm_model, latents, panda_model, reachable_poses
from ikflow_amd.ikflow_solver import IKFlowSolver
from oracle import flow_oracle as fo

dev = "cuda:0"
for name, model, gain, n in (("panda", panda_model, 1.0, 2048), ("panda", panda_model, 2.0, 2048), ("fetch_arm", fetch_arm_model, 1.0, 1024)):
    robot, hp, lay, sd = model(seed=7, gain=gain)
    _, poses = reachable_poses(robot, n, 70)
    lat = latents(n, lay.dim, 71)
    cond = torch.cat([poses, torch.zeros(n, 1)], 1)
    ref64 = fo.flow_inverse_f64(sd, lay, lat.numpy(), cond.numpy())[:, : lay.ndof]
    cpu32 = fo.flow_inverse_torch(sd, lay, lat, cond).numpy()[:, : lay.ndof]
    s = IKFlowSolver(hp, robot); s.load_state_dict_tensors(sd)
    out = {}
    for prec in ("f32", "f16x3"):
        s.set_precision(prec)
        out[prec] = s.generate_ik_solutions(poses.to(dev), latent=lat.to(dev), clamp_to_joint_limits=False).cpu().numpy()
    scale = np.maximum(1.0, np.abs(ref64))
    def err(x): e = np.abs(x - ref64) / scale; return {"max": float(e.max()), "rms": float(np.sqrt((e ** 2).mean()))}
    print(json.dumps({"model": name, "output_gain": gain, "rows": n, "joint_value_range": [float(ref64.min()), float(ref64.max())],
                      "err_vs_fp64_twin(relative to max(1,|q|))": {"hip_f32_mfma": err(out["f32"]), "hip_f16x3_split": err(out["f16x3"]), "torch_cpu_fp32_oracle": err(cpu32)},
                      "max_abs_hip_f32_vs_cpu32": float(np.abs(out["f32"] - cpu32).max()), "max_abs_hip_f16x3_vs_cpu32": float(np.abs(out["f16x3"] - cpu32).max())}))
