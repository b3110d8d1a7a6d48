"""Runs BASELINE.json's remaining single-GPU configurations once and prints one JSON line each (records for profiles/):
  config 3: Panda generate_exact_ik_solutions, B=4096, 1 mm / 0.01 rad, repeat_counts (1,3,10)
  config 4: FetchArm fetch_arm__large__mh186_9.25m, B=8192 approximate
  config 5 (1 GPU share): Panda, 1,000,000 target poses in one call (chunked inside the engine), and its 125k-per-GPU shard
Weights are seeded random (no network), so exact-IK valid counts measure cost, not convergence; a second exact run seeds LM
from perturbed true configurations through the public LM/pose-error entry points to show convergence."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import MODEL_DESCRIPTIONS, hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import get_robot

dev = torch.device("cuda:0")
EPS = 0.004363323129985824


def make(model, precision="f32"):
    robot = get_robot(MODEL_DESCRIPTIONS[model]["robot_name"]); hp = hparams_for(model); lay = layout_from(hp, robot)
    s = IKFlowSolver(hp, robot); s.load_state_dict_tensors(random_state_dict(lay, robot, 0)); s.set_precision(precision)
    return robot, lay, s


def timeit(fn, warm, reps):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps, out


for prec in ("f32", "f16x3"):
    robot, lay, s = make("panda__full__lp191_5.25m", prec)
    q = torch.tensor(robot.sample_joint_angles(4096, EPS, np.random.default_rng(0)), device=dev); poses = robot.forward_kinematics(q)
    dt, (sol, valid) = timeit(lambda: s.generate_exact_ik_solutions(poses, pos_error_threshold=1e-3, rot_error_threshold=0.01), 2, 5)
    eng = s.engine(dev); _, _, stats = eng.generate_exact(poses, (1, 3, 10), 1e-3, 0.01, return_stats=True)
    print(json.dumps({"config": 3, "precision": prec, "workload": "Panda exact IK B=4096 (1mm/0.01rad, (1,3,10))", "target_poses_per_s": 4096 / dt,
                      "ms_per_call": dt * 1e3, "valid": int(valid.sum()), "flow_rows_per_call": int(stats[:, 1].sum()), "lm_row_iters_per_call": int(stats[:, 2].sum()),
                      "flow_rows_per_s": float(stats[:, 1].sum()) / dt, "note": "random weights: seeds uninformative"}))
    # LM convergence from perturbed truth (public kernels): 3 steps
    q0 = robot.clamp_to_joint_limits(q + 0.05 * torch.randn_like(q))
    def lm3():
        x = q0
        for _ in range(3): x = robot.inverse_kinematics_step_levenburg_marquardt(poses, x)
        return eng.pose_error(x, poses)
    dt, (pe, re) = timeit(lm3, 3, 20)
    print(json.dumps({"config": "3b", "workload": "3 LM steps + pose error from q_true + N(0,0.05^2), B=4096", "ms": dt * 1e3,
                      "lm_rows_per_s": 3 * 4096 / dt, "converged_frac": float(((pe < 1e-3) & (re < 0.01)).float().mean())}))
    robot, lay, s = make("fetch_arm__large__mh186_9.25m", prec)
    q = torch.tensor(robot.sample_joint_angles(8192, EPS, np.random.default_rng(1)), device=dev); poses = robot.forward_kinematics(q)
    lat = torch.randn(8192, lay.dim, device=dev)
    dt, _ = timeit(lambda: s.generate_ik_solutions(poses, latent=lat), 5, 20)
    print(json.dumps({"config": 4, "precision": prec, "workload": "FetchArm 16-block flow B=8192 approx", "solutions_per_s": 8192 / dt, "ms_per_step": dt * 1e3,
                      "tflops": 8192 / dt * lay.flops_per_solution() / 1e12}))
    robot, lay, s = make("panda__full__lp191_5.25m", prec)
    for n in (125000, 1000000):
        q = torch.tensor(robot.sample_joint_angles(n, EPS, np.random.default_rng(2)), device=dev); poses = robot.forward_kinematics(q)
        lat = torch.randn(n, lay.dim, device=dev)
        dt, sol = timeit(lambda: s.generate_ik_solutions(poses, latent=lat), 1, 3)
        print(json.dumps({"config": 5, "precision": prec, "workload": f"Panda approx, {n} poses in one call on 1 GPU (16384-row chunks)", "solutions_per_s": n / dt, "ms_per_call": dt * 1e3}))
