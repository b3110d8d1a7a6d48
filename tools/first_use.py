"""What first use costs on a fresh handle (one MI355X): ikf_load_weights (packing on the host, upload, the row-owner stream's pack
launches, the placement census, exchange buffers), the first call of each headline size, the small-batch per-layer image when a call
reaches that path.  `python tools/first_use.py [model]` -> one JSON line (profiles/r05_first_use.jsonl)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import fetch_arm_model, latents, panda_model, reachable_poses  # noqa: E402

from ikflow_amd.ikflow_solver import IKFlowSolver  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "panda"
    robot, hp, lay, sd = panda_model() if which == "panda" else fetch_arm_model()
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    out = {"model": which, "nb_nodes": lay.nb_nodes, "dim": lay.dim}
    s = IKFlowSolver(hp, robot)
    t0 = time.perf_counter()
    s.load_state_dict_tensors(sd)
    eng = s.engine(dev)
    torch.cuda.synchronize()
    out["load_state_dict_tensors_to_engine_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    out["ikf_load_weights_ms"] = round(eng.load_time_ms, 2)
    for n in (4096, 512, 16):
        _, poses = reachable_poses(robot, n, 3)
        lat = latents(n, lay.dim, 4)
        P, L = poses.to(dev), lat.to(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.generate_ik_solutions(P, latent=L)
        torch.cuda.synchronize()
        first = (time.perf_counter() - t0) * 1e3
        for _ in range(20):
            s.generate_ik_solutions(P, latent=L)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            s.generate_ik_solutions(P, latent=L)
        torch.cuda.synchronize()
        out[f"first_call_{n}_ms"] = round(first, 3)
        out[f"steady_call_{n}_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
        out[f"plan_{n}"] = eng.plan(n)
    out["frag_image_ms_default_plan"] = round(eng.frag_image_time_ms, 2)
    eng.set_gemm_variant(180)
    eng.set_gemm_variant(185)
    _, poses = reachable_poses(robot, 256, 5)
    lat = latents(256, lay.dim, 6)
    t0 = time.perf_counter()
    s.generate_ik_solutions(poses.to(dev), latent=lat.to(dev))
    torch.cuda.synchronize()
    out["first_per_layer_call_256_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    out["frag_image_ms"] = round(eng.frag_image_time_ms, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
