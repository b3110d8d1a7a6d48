"""The reference's runtime ladders on the MI355X engine (what scripts/benchmark_runtime.py:80-97 and
scripts/benchmark_generate_exact_solutions.py:88-139 time): for each batch size, k repeats of
``generate_ik_solutions`` ("ikflow - NOT EXACT") and ``generate_exact_ik_solutions`` (1 mm / 0.01 rad), plus the three
developer-note cases of ikflow/ikflow_solver.py:135-157 (n = 500 / 1000 / 5000, default thresholds).

Differences, on purpose: the device is synchronised around every timed call (the reference reads ``time()`` without a
sync); the Klampt baselines are not reproduced; with ``--model_name`` pointing at a model whose weight file is not on
disk (there is no network) seeded random weights of that architecture are used - the runtime of the approximate path does
not depend on the weights, the exact path then runs its worst case (all three retry rounds, 14 flow rows per pose).

  python scripts/benchmark_runtime.py --model_name=panda__full__lp191_5.25m [--k 3] [--precision f16x3] [--out table.jsonl]
"""
import argparse
import json
import os
import sys
from time import perf_counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import MODEL_DESCRIPTIONS, hparams_for, layout_from, random_state_dict
from ikflow_amd.model_loading import get_ik_solver
from ikflow_amd.robots import get_robot

POS_ERROR_THRESHOLD = 0.001
ROT_ERROR_THRESHOLD = 0.01


def fn_mean_std(fn, k):
    runtimes = []
    for _ in range(k):
        torch.cuda.synchronize()
        t0 = perf_counter()
        fn()
        torch.cuda.synchronize()
        runtimes.append(1000 * (perf_counter() - t0))
    return float(np.mean(runtimes)), float(np.std(runtimes))


def load_solver(model_name: str):
    try:
        solver, _ = get_ik_solver(model_name)
        return solver, "released weights"
    except Exception:
        robot = get_robot(MODEL_DESCRIPTIONS[model_name]["robot_name"])
        hp = hparams_for(model_name)
        solver = IKFlowSolver(hp, robot)
        solver.load_state_dict_tensors(random_state_dict(layout_from(hp, robot), robot, seed=0))
        return solver, "seeded random weights (weight file not on disk)"


if __name__ == "__main__":
    parser = argparse.ArgumentParser(prog="benchmark_runtime.py - the reference's runtime ladders on the MI355X engine")
    parser.add_argument("--model_name", type=str, default="panda__full__lp191_5.25m")
    parser.add_argument("--k", type=int, default=3, help="Number of repeats per batch size")
    parser.add_argument("--precision", type=str, default="f32", choices=["f32", "f16x3"])
    parser.add_argument("--out", type=str, default=None)
    args = parser.parse_args()
    dev = torch.device("cuda:0")
    solver, weights = load_solver(args.model_name)
    solver.set_precision(args.precision)
    robot = solver.robot
    rng = np.random.default_rng(0)
    rows = []

    def poses_for(n):
        q = torch.tensor(robot.sample_joint_angles(n, 0.004363323129985824, rng), device=dev)
        return robot.forward_kinematics(q)

    solver.generate_ik_solutions(poses_for(8))  # engine + scratch creation outside the timed region
    for batch_size in sorted({1, 2, 5, 10, 50, 100, 500, 1000, 2500, 5000}):
        target_poses = poses_for(batch_size)
        for name, fn in (
            ("ikflow - NOT EXACT", lambda: solver.generate_ik_solutions(target_poses.clone(), n=1 if batch_size == 1 else None)),  # a [1 x 7] y is the single-pose form (ikflow_solver.py:313-315)
            ("ikflow with levenberg-marquardt (1mm / 0.01rad)",
             lambda: solver.generate_exact_ik_solutions(target_poses.clone(), pos_error_threshold=POS_ERROR_THRESHOLD,
                                                        rot_error_threshold=ROT_ERROR_THRESHOLD)),
            ("ikflow with levenberg-marquardt (defaults: 1mm / 0.1rad)", lambda: solver.generate_exact_ik_solutions(target_poses.clone())),
        ):
            fn()
            mean_ms, std_ms = fn_mean_std(fn, args.k)
            row = {"method": name, "number of solutions": batch_size, "total runtime (ms)": round(mean_ms, 4),
                   "runtime std": round(std_ms, 4), "runtime per solution (ms)": round(mean_ms / batch_size, 6),
                   "precision": args.precision, "weights": weights}
            rows.append(row)
            print(json.dumps(row), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")
