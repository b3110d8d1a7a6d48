"""ORACLE (test infrastructure - never imported by the product path in ikflow_amd/; imports nothing from ikflow_amd).

CPU restatement (torch, dtype-generic: float32 = the reference's CPU arithmetic, float64 = arbitration twin)
of the kinematics the exact-IK path calls.  The arithmetic lives in jrl @ git 2ba7c3995b36b32886a8aa021a00c73b2cd55b2c
(pyproject.toml:22, uv.lock:875-877) which is NOT in /root/reference and not installed.  Call sites followed:

  robot.forward_kinematics(q)                                   ikflow/ikflow_solver.py:114, evaluation_utils.py:86
  geodesic_distance_between_quaternions(q_target, q_realized)   ikflow/ikflow_solver.py:116, evaluation_utils.py:47-49,96
  robot.inverse_kinematics_step_levenburg_marquardt(poses, q)   ikflow/ikflow_solver.py:205,208  (defaults only)
  robot.clamp_to_joint_limits(q)                                ikflow/ikflow_solver.py:101-102
  _calculate_pose_error                                          ikflow/ikflow_solver.py:112-117
  _generate_exact_ik_solutions / generate_exact_ik_solutions     ikflow/ikflow_solver.py:119-247, 345-411

Restated jrl algorithm (published code of jrl/robot.py, jrl/math_utils.py):
  FK      : T = I; for joint on chain: T = T @ T_fixed(origin_xyz, origin_rpy); revolute: T = T @ Rot(axis, q_i)
            (Rodrigues), prismatic: T = T @ Trans(axis*q_i); pose = [T[:3,3], quat_wxyz(T[:3,:3])]
            quaternion by the largest-component ("pytorch3d matrix_to_quaternion") rule
  geodesic: d = 2*acos(clamp(sum(q1*q2), -1+1e-7, 1-1e-7)); d = |((d + pi) mod 2pi) - pi|
  LM step : J [n,6,ndof] rows = [angular; linear]; e = [rpy(q_target * conj(q_cur)); p_target - p_cur];
            dq = solve(J^T J + 1e-4 I, J^T e); q <- clamp_to_limits(q + 1.0*dq)

PARITY STATUS: pinned by the reference's tests ONLY at Panda FK(q=0) and the pi geodesic / L2 known answers
(tests/evaluation_utils_test.py:18-32) and the Panda limits (tests/model_test.py:27-44) - all checked in
tests/test_oracle_golden.py.  The LM step (row order, rpy parametrisation, lambda, alpha) and every FetchArm
number are "parity unpinned": restated from memory of jrl's published code, no vector to check against - until jrl is importable:
tests/test_thirdparty_pin.py then compares FK, the LM step, the geodesic distance, limits and clamp with jrl's own (armed, dormant here).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from oracle.robot_tables import FIXED as JOINT_FIXED
from oracle.robot_tables import PRISMATIC as JOINT_PRISMATIC
from oracle.robot_tables import REVOLUTE as JOINT_REVOLUTE
from oracle.robot_tables import OracleRobot
from oracle.robot_tables import robot as _robot_by_name
from oracle.robot_tables import rpy_matrix as rpy_to_matrix


def _R(robot) -> OracleRobot:
    """The oracle computes on ITS OWN chain tables (oracle/robot_tables.py): of a product Robot only the name is used."""
    return robot if isinstance(robot, OracleRobot) else _robot_by_name(robot if isinstance(robot, str) else robot.name)

Robot = OracleRobot

LM_LAMBDA = 1e-4
LM_ALPHA = 1.0
ACOS_EPS = 1e-7


# ---------------------------------------------------------------------------------------------------
# rotations / quaternions (w, x, y, z)
# ---------------------------------------------------------------------------------------------------
def axis_angle_to_matrix(axis: Sequence[float], angle: torch.Tensor) -> torch.Tensor:
    """Rodrigues rotation about a fixed unit axis by a batch of angles -> [n,3,3]."""
    ax = np.asarray(axis, dtype=np.float64)
    ax = ax / np.linalg.norm(ax)
    x, y, z = (float(v) for v in ax)
    c, s = torch.cos(angle), torch.sin(angle)
    t = 1.0 - c
    R = torch.stack(
        [
            t * x * x + c, t * x * y - s * z, t * x * z + s * y,
            t * x * y + s * z, t * y * y + c, t * y * z - s * x,
            t * x * z - s * y, t * y * z + s * x, t * z * z + c,
        ],
        dim=1,
    )
    return R.reshape(-1, 3, 3)


def matrix_to_quaternion(R: torch.Tensor) -> torch.Tensor:
    """[n,3,3] -> [n,4] (w,x,y,z): candidate built from the largest of |w|,|x|,|y|,|z| (that component > 0)."""
    m00, m01, m02 = R[:, 0, 0], R[:, 0, 1], R[:, 0, 2]
    m10, m11, m12 = R[:, 1, 0], R[:, 1, 1], R[:, 1, 2]
    m20, m21, m22 = R[:, 2, 0], R[:, 2, 1], R[:, 2, 2]
    zero = torch.zeros_like(m00)
    q_abs = torch.sqrt(
        torch.maximum(
            torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=1),
            zero[:, None],
        )
    )
    cands = torch.stack(
        [
            torch.stack([q_abs[:, 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=1),
            torch.stack([m21 - m12, q_abs[:, 1] ** 2, m10 + m01, m02 + m20], dim=1),
            torch.stack([m02 - m20, m10 + m01, q_abs[:, 2] ** 2, m12 + m21], dim=1),
            torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[:, 3] ** 2], dim=1),
        ],
        dim=1,
    )  # [n, 4(cand), 4]
    floor = torch.tensor(0.1, dtype=R.dtype)
    cands = cands / (2.0 * torch.maximum(q_abs[:, :, None], floor))
    best = torch.argmax(q_abs, dim=1)
    return cands[torch.arange(R.shape[0]), best, :]


def quaternion_conjugate(q: torch.Tensor) -> torch.Tensor:
    return torch.cat([q[:, 0:1], -q[:, 1:4]], dim=1)


def quaternion_product(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Hamilton product a*b, (w,x,y,z)."""
    w1, x1, y1, z1 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    w2, x2, y2, z2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    return torch.stack(
        [
            w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
            w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
            w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
            w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
        ],
        dim=1,
    )


def quaternion_to_rpy(q: torch.Tensor) -> torch.Tensor:
    """(w,x,y,z) -> fixed-axis roll, pitch, yaw."""
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    roll = torch.atan2(2.0 * (w * x + y * z), 1.0 - 2.0 * (x * x + y * y))
    pitch = torch.asin(torch.clamp(2.0 * (w * y - z * x), -1.0, 1.0))
    yaw = torch.atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))
    return torch.stack([roll, pitch, yaw], dim=1)


def geodesic_distance_between_quaternions(q1: torch.Tensor, q2: torch.Tensor, acos_epsilon: Optional[float] = None):
    eps = ACOS_EPS if acos_epsilon is None else acos_epsilon
    dot = torch.clip(torch.sum(q1 * q2, dim=1), -1.0 + eps, 1.0 - eps)
    d = 2.0 * torch.acos(dot)
    return torch.abs(torch.remainder(d + math.pi, 2.0 * math.pi) - math.pi)


# ---------------------------------------------------------------------------------------------------
# FK / Jacobian
# ---------------------------------------------------------------------------------------------------
def _fixed_T(joint, dtype) -> torch.Tensor:
    T = np.eye(4)
    T[:3, :3] = rpy_to_matrix(joint.origin_rpy)
    T[:3, 3] = joint.origin_xyz
    return torch.tensor(T, dtype=dtype)


def _chain_transforms(robot: Robot, q: torch.Tensor):
    """Walk the chain; returns final T [n,4,4] and, per actuated joint, (kind, axis_world [n,3], origin_world [n,3])."""
    robot = _R(robot)
    n, dtype = q.shape[0], q.dtype
    T = torch.eye(4, dtype=dtype).expand(n, 4, 4).contiguous()
    per_joint = []
    qi = 0
    for joint in robot.joints:
        T = T.bmm(_fixed_T(joint, dtype).expand(n, 4, 4))
        if joint.kind == JOINT_FIXED:
            continue
        ax = np.asarray(joint.axis, dtype=np.float64)
        ax = ax / np.linalg.norm(ax)
        ax_t = torch.tensor(ax, dtype=dtype)
        axis_world = T[:, :3, :3] @ ax_t
        origin_world = T[:, :3, 3].clone()
        per_joint.append((joint.kind, axis_world, origin_world))
        M = torch.eye(4, dtype=dtype).expand(n, 4, 4).contiguous()
        if joint.kind == JOINT_REVOLUTE:
            M[:, :3, :3] = axis_angle_to_matrix(joint.axis, q[:, qi])
        elif joint.kind == JOINT_PRISMATIC:
            M[:, :3, 3] = ax_t[None, :] * q[:, qi, None]
        T = T.bmm(M)
        qi += 1
    assert qi == robot.ndof
    return T, per_joint


def forward_kinematics(robot: Robot, q: torch.Tensor) -> torch.Tensor:
    """[n x ndof] -> [n x 7] = (x, y, z, qw, qx, qy, qz)."""
    T, _ = _chain_transforms(robot, q)
    return torch.cat([T[:, :3, 3], matrix_to_quaternion(T[:, :3, :3])], dim=1)


def jacobian(robot: Robot, q: torch.Tensor) -> torch.Tensor:
    """[n x 6 x ndof], rows 0-2 angular, rows 3-5 linear (world frame, end-effector origin)."""
    robot = _R(robot)
    T, per_joint = _chain_transforms(robot, q)
    p_ee = T[:, :3, 3]
    J = torch.zeros(q.shape[0], 6, robot.ndof, dtype=q.dtype)
    for i, (kind, axis_w, origin_w) in enumerate(per_joint):
        if kind == JOINT_REVOLUTE:
            J[:, 0:3, i] = axis_w
            J[:, 3:6, i] = torch.cross(axis_w, p_ee - origin_w, dim=1)
        else:
            J[:, 3:6, i] = axis_w
    return J


def clamp_to_joint_limits(robot: Robot, q: torch.Tensor) -> torch.Tensor:
    robot = _R(robot)
    lo = torch.tensor([l[0] for l in robot.actuated_joints_limits], dtype=q.dtype)
    hi = torch.tensor([l[1] for l in robot.actuated_joints_limits], dtype=q.dtype)
    return torch.max(torch.min(q, hi), lo)


def pose_error_vector(robot: Robot, target_poses: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    """[n x 6] = [rpy(q_target * conj(q_current)), p_target - p_current]."""
    cur = forward_kinematics(robot, q)
    rot_err = quaternion_to_rpy(quaternion_product(target_poses[:, 3:7], quaternion_conjugate(cur[:, 3:7])))
    return torch.cat([rot_err, target_poses[:, 0:3] - cur[:, 0:3]], dim=1)


def lm_step(robot: Robot, target_poses: torch.Tensor, q: torch.Tensor, lambd: float = LM_LAMBDA, alpha: float = LM_ALPHA) -> torch.Tensor:
    robot = _R(robot)
    J = jacobian(robot, q)
    e = pose_error_vector(robot, target_poses, q)[:, :, None]
    Jt = J.transpose(1, 2)
    A = Jt.bmm(J) + lambd * torch.eye(robot.ndof, dtype=q.dtype)[None]
    dq = torch.linalg.solve(A, Jt.bmm(e))[:, :, 0]
    return clamp_to_joint_limits(robot, q + alpha * dq)


def calculate_pose_error(robot: Robot, q: torch.Tensor, target_poses: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """ikflow_solver.py:112-117."""
    realized = forward_kinematics(robot, q)
    pos = torch.norm(realized[:, 0:3] - target_poses[:, 0:3], dim=1)
    rot = geodesic_distance_between_quaternions(target_poses[:, 3:], realized[:, 3:])
    return pos, rot


def calculate_joint_limits_exceeded(configs: torch.Tensor, joint_limits) -> torch.Tensor:
    """evaluation_utils.py:100-112 (strict inequalities)."""
    toolarge = configs > torch.tensor([x[1] for x in joint_limits], dtype=torch.float32)
    toosmall = configs < torch.tensor([x[0] for x in joint_limits], dtype=torch.float32)
    return torch.logical_or(toolarge, toosmall).any(dim=1)


# ---------------------------------------------------------------------------------------------------
# exact-IK control loop (ikflow_solver.py:119-247, 345-411), flow seeds supplied by a callback so the same
# loop can be driven by the torch oracle flow (CPU parity) or by recorded seeds.
# ---------------------------------------------------------------------------------------------------
def exact_round(robot, seeds_q, target_poses, repeat_count, pos_thr, rot_thr, n_opt_steps_max=3, lm_dtype=torch.float32,
                margins=None, q_ulps=0):
    """One call of _generate_exact_ik_solutions given the clamped flow seeds q [n*R x ndof] (tile-major).
    lm_dtype=float64 evaluates each LM step in double and rounds q back to float32 (what the HIP kernel does).
    ``q_ulps`` (tests): every q that leaves an LM step is moved by that many fp32 ulps - the twin run that measures, pose by pose, what a
    last-bit difference in an intermediate iterate does to the result (the step map amplifies along the arm's self-motion by up to
    |e| |d2x/dq2| / lambda).
    ``margins`` (optional, [n, 2] float tensor, updated in place): per pose the smallest |pos_err - pos_thr| and
    |rot_err - rot_thr| seen over every (iteration, repeat) evaluated for it - lets a test exclude poses whose validity
    flag hangs on rounding."""
    robot = _R(robot)
    n = target_poses.shape[0]
    active = torch.arange(n)
    q = seeds_q.clone()
    poses_tiled = target_poses.repeat((repeat_count, 1))
    final_solutions = torch.zeros(n, robot.ndof, dtype=torch.float32)
    final_valids = torch.zeros(n, dtype=torch.bool)
    n_invalid = n
    for _ in range(n_opt_steps_max):
        assert len(q) == n_invalid * repeat_count
        q = lm_step(robot, poses_tiled.to(lm_dtype), q.to(lm_dtype)).to(torch.float32)
        for _u in range(abs(int(q_ulps))):
            q = torch.nextafter(q, torch.full_like(q, float("inf") if q_ulps > 0 else -float("inf")))
        q = clamp_to_joint_limits(robot, q)
        pos_err, rot_err = calculate_pose_error(robot, q, poses_tiled)
        valids_tiled = torch.logical_and(pos_err < pos_thr, rot_err < rot_thr)
        if margins is not None:
            pm = (pos_err - pos_thr).abs().reshape(repeat_count, n_invalid).min(0).values
            rm = (rot_err - rot_thr).abs().reshape(repeat_count, n_invalid).min(0).values
            margins[active, 0] = torch.minimum(margins[active, 0], pm.to(margins.dtype))
            margins[active, 1] = torch.minimum(margins[active, 1], rm.to(margins.dtype))
        valids_i = torch.zeros(n_invalid, dtype=torch.bool)
        sols_i = torch.zeros((n_invalid, robot.ndof), dtype=torch.float32)
        for idx in torch.nonzero(valids_tiled)[:, 0].tolist():  # ascending: highest valid repeat wins (:217-222)
            sols_i[idx % n_invalid, :] = q[idx, :]
            valids_i[idx % n_invalid] = True
        not_valid = torch.logical_not(final_valids)
        final_solutions[not_valid] = sols_i
        final_valids[not_valid] = valids_i
        if final_valids.all():
            return final_solutions, final_valids
        keep = torch.logical_not(valids_i).repeat((repeat_count))
        active = active[torch.logical_not(valids_i)]
        q = q[keep, :]
        poses_tiled = poses_tiled[keep, :]
        n_invalid = n - int(final_valids.sum().item())
    return final_solutions, final_valids


def generate_exact_ik_solutions(robot, flow_fn, target_poses, latents: List[torch.Tensor], repeat_counts=(1, 3, 10), pos_thr=1e-3, rot_thr=0.1,
                                lm_dtype=torch.float32):
    """Retry schedule of ikflow_solver.py:345-411.  ``flow_fn(latent, poses_tiled) -> clamped q`` ;
    ``latents[r]`` is the [n_r*R_r x D] latent the reference would have drawn in round r."""
    robot = _R(robot)
    n = target_poses.shape[0]
    R0 = repeat_counts[0]
    seeds = flow_fn(latents[0], target_poses.repeat((R0, 1)))
    solutions, valids = exact_round(robot, seeds, target_poses, R0, pos_thr, rot_thr, lm_dtype=lm_dtype)
    if valids.all():
        return solutions, valids
    for r in range(1, len(repeat_counts)):
        R = repeat_counts[r]
        missing = target_poses[torch.logical_not(valids), :]
        if missing.shape[0] == 0:
            break
        seeds = flow_fn(latents[r][: missing.shape[0] * R], missing.repeat((R, 1)))
        new_sol, new_valid = exact_round(robot, seeds, missing, R, pos_thr, rot_thr, lm_dtype=lm_dtype)
        not_valid = torch.logical_not(valids)
        solutions[not_valid, :] = new_sol
        valids[not_valid] = new_valid
        if new_sol.all():  # quirk Q3 (ikflow_solver.py:402)
            return solutions, valids
    return solutions, valids


def generate_exact_ik_solutions_seeded(robot, seed_fn, target_poses, repeat_counts=(1, 3, 10), pos_thr=1e-3, rot_thr=0.1,
                                       lm_dtype=torch.float32, return_margins=False, q_ulps=0):
    """The same retry schedule (ikflow_solver.py:345-411) with the flow taken out: ``seed_fn(round, pose_indices) ->
    [len(pose_indices) * R_round x ndof]`` supplies the (clamped) seeds of the still-invalid poses, tile-major
    (row = r * n_active + j, as ``conditional.repeat((R, 1))`` lays them out, :185).  Drives exactly the part of the path
    that follows ``self._run_inference`` (:188): LM iterations, validity, "highest valid repeat wins", slot order,
    compaction, rounds.  Optionally returns per-pose threshold margins (see exact_round)."""
    robot = _R(robot)
    n = target_poses.shape[0]
    margins = torch.full((n, 2), float("inf"), dtype=torch.float64)
    solutions = torch.zeros(n, robot.ndof, dtype=torch.float32)
    valids = torch.zeros(n, dtype=torch.bool)
    for r, R in enumerate(repeat_counts):
        idx = torch.nonzero(torch.logical_not(valids))[:, 0]
        if idx.numel() == 0:
            break
        missing = target_poses[idx, :]
        seeds = seed_fn(r, idx)
        assert seeds.shape == (idx.numel() * R, robot.ndof), (seeds.shape, idx.numel(), R)
        sub = torch.full((idx.numel(), 2), float("inf"), dtype=torch.float64)
        new_sol, new_valid = exact_round(robot, seeds, missing, R, pos_thr, rot_thr, lm_dtype=lm_dtype, margins=sub, q_ulps=q_ulps)
        margins[idx] = torch.minimum(margins[idx], sub)
        solutions[idx, :] = new_sol
        valids[idx] = new_valid
        if new_sol.all():  # quirk Q3 (ikflow_solver.py:402)
            break
    return (solutions, valids, margins) if return_margins else (solutions, valids)


# ---------------------------------------------------------------------------------------------------
# capsule self-collision (checker of ikf_self_collision; float64 numpy, brute-force closest points on a parameter grid
# refined by the closed form - no shared code with the kernel)
# ---------------------------------------------------------------------------------------------------
def _link_frames(robot: Robot, q: torch.Tensor):
    """World transforms [n x 4 x 4] of the base (index 0) and of the frame that follows each actuated joint (1..ndof),
    walking the URDF joints one by one (fixed joints included) in float64."""
    robot = _R(robot)
    qd = q.double()
    n = qd.shape[0]
    T = torch.eye(4, dtype=torch.float64).repeat(n, 1, 1)
    frames = [T.clone()]
    k = 0
    per_joint = []
    for j in robot.joints:
        T = T @ _fixed_T(j, torch.float64)
        if j.actuated:
            M = torch.eye(4, dtype=torch.float64).repeat(n, 1, 1)
            ax = torch.tensor(j.axis, dtype=torch.float64)
            ax = ax / ax.norm()
            if j.kind == 1:
                M[:, :3, :3] = axis_angle_to_matrix(ax.tolist(), qd[:, k])
            else:
                M[:, :3, 3] = ax[None, :] * qd[:, k : k + 1]
            T = T @ M
            k += 1
        per_joint.append(T.clone())
    return per_joint


def capsule_clearance(robot: Robot, capsules, ignored_pairs, q: torch.Tensor) -> torch.Tensor:
    """capsules in URDF link frames, as given to Robot.set_collision_capsules.  Returns [n] float64: min over tested pairs
    of (segment distance - r_a - r_b)."""
    robot = _R(robot)
    names = [j.name for j in robot.joints]
    per_joint = _link_frames(robot, q)
    n = q.shape[0]
    ends, radii, moving = [], [], []
    for after, p0, p1, r in capsules:
        T = torch.eye(4, dtype=torch.float64).repeat(n, 1, 1) if after is None else per_joint[names.index(after)]
        a = (T @ torch.tensor([*p0, 1.0], dtype=torch.float64))[:, :3]
        b = (T @ torch.tensor([*p1, 1.0], dtype=torch.float64))[:, :3]
        ends.append((a, b))
        radii.append(float(r))
        moving.append(0 if after is None else sum(1 for j in robot.joints[: names.index(after) + 1] if j.actuated))
    ignored = {tuple(sorted(p)) for p in ignored_pairs}
    best = torch.full((n,), float("inf"), dtype=torch.float64)
    grid = torch.linspace(0.0, 1.0, 201, dtype=torch.float64)
    for i in range(len(capsules)):
        for k in range(i + 1, len(capsules)):
            if moving[i] == moving[k] or (i, k) in ignored:
                continue
            (a0, a1), (b0, b1) = ends[i], ends[k]
            # brute force over s on a grid, exact t for each s (point-to-segment), then polish s by golden section
            def dist_for_s(sv):
                pa = a0[:, None, :] + (a1 - a0)[:, None, :] * sv[..., None]
                d2 = b1 - b0
                e = (d2 * d2).sum(-1).clamp_min(1e-300)
                tt = (((pa - b0[:, None, :]) * d2[:, None, :]).sum(-1) / e[:, None]).clamp(0.0, 1.0)
                pb = b0[:, None, :] + d2[:, None, :] * tt[..., None]
                return (pa - pb).norm(dim=-1)
            d = dist_for_s(grid[None, :].expand(n, -1))
            i0 = d.argmin(dim=1)
            lo = grid[(i0 - 1).clamp_min(0)]
            hi = grid[(i0 + 1).clamp_max(200)]
            for _ in range(60):
                m1 = lo + (hi - lo) * 0.381966011250105
                m2 = lo + (hi - lo) * 0.618033988749895
                f1 = dist_for_s(m1[:, None])[:, 0]
                f2 = dist_for_s(m2[:, None])[:, 0]
                take = f1 < f2
                hi = torch.where(take, m2, hi)
                lo = torch.where(take, lo, m1)
            dmin = torch.minimum(d.min(dim=1).values, dist_for_s(((lo + hi) / 2)[:, None])[:, 0])
            best = torch.minimum(best, dmin - radii[i] - radii[k])
    return best
