"""Model evaluation on the MI355X engine: what the reference's scripts/evaluate.py:42-108 measures - mean positional and
rotational error of `solutions_per_pose` solutions for each of `testset_size` target poses (gaussian latent, scale 0.75,
`:34-35`), percentage of solutions outside the joint limits, and the runtime for `n_solutions_for_runtime` solutions.

MI355X-first instead of the per-pose Python loop: all testset_size x solutions_per_pose rows go through ONE
`generate_ik_solutions` call (poses tiled on the device) and one `evaluate_solutions` call.  The self-collision column
needs Klampt (evaluation_utils.py:115-126) and is reported as n/a.  Without the released weight file on disk (no
network) seeded random weights of the architecture are used and the error columns only exercise the plumbing.

  python scripts/evaluate.py --testset_size=500 --model_name=panda__full__lp191_5.25m --solutions_per_pose=50 [--do_refinement]
"""
import argparse
import os
import sys
from collections import namedtuple
from time import perf_counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from benchmark_runtime import load_solver
from ikflow_amd.evaluation_utils import evaluate_solutions

_K_FOR_RUNTIME_STATS = 5
_DEFAULT_LATENT_DISTRIBUTION = "gaussian"
_DEFAULT_LATENT_SCALE = 0.75
ErrorStats = namedtuple("ErrorStats", "mean_l2_error_mm mean_angular_error_deg pct_joint_limits_exceeded pct_self_colliding")
RuntimeStats = namedtuple("RuntimeStats", "mean_runtime_ms runtime_std nb_solutions")


def boolean_string(s):
    if isinstance(s, bool):
        return s
    if s.lower() not in {"false", "true"}:
        raise ValueError("Not a valid boolean string")
    return s.lower() == "true"


def calculate_error_stats(ik_solver, robot, testset: torch.Tensor, latent_distribution, latent_scale, solutions_per_pose,
                          refine_solutions, clamp_to_joint_limits) -> ErrorStats:
    """testset [m x 7] on the device.  Row i*solutions_per_pose + j = solution j of pose i."""
    tiled = testset.repeat_interleave(solutions_per_pose, dim=0)
    if refine_solutions:
        solutions, _ = ik_solver.generate_exact_ik_solutions(tiled)
    else:
        solutions = ik_solver.generate_ik_solutions(tiled, latent_distribution=latent_distribution, latent_scale=latent_scale,
                                                    clamp_to_joint_limits=clamp_to_joint_limits)
    pos_errs, rot_errs, joint_limits_exceeded, _ = evaluate_solutions(robot, tiled, solutions)
    return ErrorStats(1000 * pos_errs.mean().item(), float(np.rad2deg(rot_errs.mean().item())),
                      100 * joint_limits_exceeded.float().mean().item(), float("nan"))


def calculate_runtime_stats(ik_solver, n_solutions: int, k: int, rng) -> RuntimeStats:
    robot = ik_solver.robot
    dev = torch.device("cuda:0")
    q = torch.tensor(robot.sample_joint_angles(n_solutions * k, 0.0, rng), device=dev)
    poses = robot.forward_kinematics(q)
    ik_solver.generate_ik_solutions(poses[: max(n_solutions, 2)])
    sample_times = []
    for k_i in range(k):
        target_poses = poses[k_i * n_solutions : (k_i + 1) * n_solutions]
        torch.cuda.synchronize()
        t0 = perf_counter()
        ik_solver.generate_ik_solutions(target_poses, n=(1 if n_solutions == 1 else None))
        torch.cuda.synchronize()
        sample_times.append(perf_counter() - t0)
    return RuntimeStats(float(np.mean(sample_times)) * 1000, float(np.std(sample_times)), n_solutions)


if __name__ == "__main__":
    parser = argparse.ArgumentParser(prog="evaluate.py - evaluates IK models on the MI355X engine")
    parser.add_argument("--solutions_per_pose", default=50, type=int)
    parser.add_argument("--n_solutions_for_runtime", default=100, type=int)
    parser.add_argument("--testset_size", default=500, type=int)
    parser.add_argument("--model_name", type=str, default="panda__full__lp191_5.25m")
    parser.add_argument("--do_refinement", action="store_true")
    parser.add_argument("--clamp_to_joint_limits", type=str, default="true")
    parser.add_argument("--precision", type=str, default="f32", choices=["f32", "f16x3"])
    args = parser.parse_args()
    args.clamp_to_joint_limits = boolean_string(args.clamp_to_joint_limits)

    solver, weights = load_solver(args.model_name)
    solver.set_precision(args.precision)
    robot = solver.robot
    rng = np.random.default_rng(0)
    dev = torch.device("cuda:0")
    testset = robot.forward_kinematics(torch.tensor(robot.sample_joint_angles(args.testset_size, 0.0, rng), device=dev))
    error_stats = calculate_error_stats(solver, robot, testset, _DEFAULT_LATENT_DISTRIBUTION, _DEFAULT_LATENT_SCALE,
                                        args.solutions_per_pose, args.do_refinement, args.clamp_to_joint_limits)
    runtime_stats = calculate_runtime_stats(solver, args.n_solutions_for_runtime, _K_FOR_RUNTIME_STATS, rng)
    print("\n----------------------------------------")
    print(f"> Results for {args.model_name.upper()}   [{weights}; precision {args.precision}]")
    print(f"\n  solutions clamped to joint limits: {args.clamp_to_joint_limits}")
    print(f"  solutions refined:                 {args.do_refinement}")
    print(f"\n  Average positional error:      {round(error_stats.mean_l2_error_mm, 4)} mm")
    print(f"  Average rotational error:      {round(error_stats.mean_angular_error_deg, 4)} deg")
    print(f"  Percent joint limits exceeded: {round(error_stats.pct_joint_limits_exceeded, 4)} %")
    print("  Percent self-colliding:        n/a (Klampt check not available)")
    print(f"  Average runtime:               {round(runtime_stats.mean_runtime_ms, 4)} +/- {round(runtime_stats.runtime_std * 1000, 4)} ms"
          f" for {runtime_stats.nb_solutions} solutions")
    print(f"                                 {round(runtime_stats.mean_runtime_ms / runtime_stats.nb_solutions, 5)} ms per solution")
