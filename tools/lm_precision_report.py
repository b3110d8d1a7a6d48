"""The LM step in the reference's arithmetic (ikf_set_lm_precision 0: fp32 throughout, LU with partial pivoting) and in the default (fp64
inside), each against the oracle's fp32 step (torch CPU: the reference's literal arithmetic, ikflow_solver.py:205,208 -> jrl) and against an
fp64 evaluation of the same step, stratified by cond(J^T J + 1e-4 I).

    python tools/lm_precision_report.py --n 4096 --out profiles/r06_lm_precision.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch


def strat(cond, err, edges=(0, 1e2, 1e3, 1e4, 1e5, 1e6, float("inf"))):
    out = []
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (cond >= lo) & (cond < hi)
        if int(m.sum()) == 0:
            continue
        e = err[m]
        out.append({"cond_from": lo, "cond_to": hi if np.isfinite(hi) else None, "rows": int(m.sum()), "max": float(e.max()), "p99": float(np.quantile(e, 0.99)),
                    "median": float(np.median(e))})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--sigma", type=float, default=0.05, help="seed = q_true + N(0, sigma^2) (SURVEY 8(d) config 3, perturbed-truth mode)")
    ap.add_argument("--out", default="gpurun_out/r06/lm_precision.json")
    a = ap.parse_args()
    from helpers import panda_model, reachable_poses
    from ikflow_amd.ikflow_solver import IKFlowSolver
    from oracle import kinematics_oracle as ko

    robot, hp, lay, sd = panda_model()
    s = IKFlowSolver(hp, robot)
    eng = s.engine("cuda:0")
    q_true, poses = reachable_poses(robot, a.n, 0)
    g = torch.Generator().manual_seed(3)
    seeds = ko.clamp_to_joint_limits(robot, q_true + a.sigma * torch.randn(q_true.shape, generator=g))
    J = ko.jacobian(robot, seeds.double())
    A = J.transpose(1, 2) @ J + 1e-4 * torch.eye(7, dtype=torch.float64)
    cond = torch.linalg.cond(A).numpy()
    ref32 = ko.lm_step(robot, poses, seeds)
    ref64 = ko.lm_step(robot, poses.double(), seeds.double())
    step = (ref64 - seeds.double()).abs().max(dim=1).values.numpy()          # |dq| of the (clamped) step, fp64
    unit = cond * 2.0 ** -24 * np.maximum(step, 1e-3)                        # the noise an fp32 evaluation of the step carries: cond x eps x |dq|
    res = {"n": a.n, "sigma": a.sigma, "cond_quantiles": {k: float(np.quantile(cond, v)) for k, v in (("p50", 0.5), ("p90", 0.9), ("p99", 0.99), ("max", 1.0))}}
    dev_p, dev_s = poses.to("cuda:0"), seeds.to("cuda:0")
    for mode in ("f32", "f64"):
        eng.set_lm_precision(mode)
        got = eng.lm_step(dev_p, dev_s).cpu()
        e32 = (got - ref32).abs().max(dim=1).values.numpy()
        e64 = (got.double() - ref64).abs().max(dim=1).values.numpy()
        res["hip_" + mode] = {"vs_oracle_f32_by_cond": strat(cond, e32), "vs_f64_truth_by_cond": strat(cond, e64),
                              "vs_oracle_f32_max": float(e32.max()), "vs_f64_truth_max": float(e64.max()),
                              "vs_f64_truth_in_units_of_cond_eps_step": {k: float(np.quantile(e64 / unit, v)) for k, v in (("p50", 0.5), ("p99", 0.99), ("max", 1.0))},
                              "vs_oracle_f32_in_units_of_cond_eps_step": {k: float(np.quantile(e32 / unit, v)) for k, v in (("p50", 0.5), ("p99", 0.99), ("max", 1.0))}}
        top = np.argsort(-(e64 / unit))[:5]
        res["hip_" + mode]["worst_rows_vs_truth"] = [{"row": int(i), "units": float(e64[i] / unit[i]), "abs": float(e64[i]), "cond": float(cond[i]), "step": float(step[i]),
                                                      "q_hip": [float(v) for v in got[i]], "q_truth": [float(v) for v in ref64[i]], "q_oracle32": [float(v) for v in ref32[i]]} for i in top]
    eo = (ref32.double() - ref64).abs().max(dim=1).values.numpy()
    res["oracle_f32"] = {"vs_f64_truth_by_cond": strat(cond, eo), "vs_f64_truth_max": float(eo.max()),
                         "vs_f64_truth_in_units_of_cond_eps_step": {k: float(np.quantile(eo / unit, v)) for k, v in (("p50", 0.5), ("p99", 0.99), ("max", 1.0))}}
    res["step_quantiles"] = {k: float(np.quantile(step, v)) for k, v in (("p50", 0.5), ("p99", 0.99), ("max", 1.0))}
    eng.set_lm_precision("f64")
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res)[:3000])


if __name__ == "__main__":
    main()
