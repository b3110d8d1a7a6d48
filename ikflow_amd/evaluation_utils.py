"""Mirror of ``ikflow/evaluation_utils.py`` (the consumers of the hot path's output: SURVEY 8 f-2) on the MI355X engine.

Same function names, argument meaning and return shapes as the reference; the arithmetic runs in libikflow_amd
(``ikf_pose_distance``, ``ikf_pose_error``, ``ikf_limits_exceeded``).  numpy inputs are accepted where the reference
accepts them (``PT_NP_TYPE``) and are returned as numpy; torch inputs come back on the device they arrived on.
There is no CPU path: without a GPU every function raises ``EngineError``.

``calculate_self_collisions`` (evaluation_utils.py:115-126 calls Klampt through jrl, neither is available)
runs the engine's capsule test when the robot carries a capsule model (``Robot.set_collision_capsules`` - no geometry
ships with the built-in robots); otherwise it raises ``NotImplementedError`` and ``evaluate_solutions`` returns ``None``
in that slot.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from ikflow_amd import _lib
from ikflow_amd.engine import EngineError, _check, kinematics_engine_for
from ikflow_amd.robots import Robot

PT_NP_TYPE = Union[np.ndarray, torch.Tensor]


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise EngineError("no GPU visible: ikflow_amd.evaluation_utils runs on the MI355X only (there is no CPU path)")
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev(x: PT_NP_TYPE, dev: torch.device) -> torch.Tensor:
    t = torch.as_tensor(x)
    return t.to(device=dev, dtype=torch.float32).contiguous()


def _back(t: torch.Tensor, like: PT_NP_TYPE) -> PT_NP_TYPE:
    if isinstance(like, np.ndarray):
        return t.cpu().numpy()
    return t.to(like.device)


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _get_target_pose_batch(target_pose: PT_NP_TYPE, n_solutions: int) -> PT_NP_TYPE:
    """[7] -> tiled [n x 7]; [n x 7] -> unchanged (evaluation_utils.py:22-34: the test is ``shape[0] == 7``)."""
    if target_pose.shape[0] == 7:
        if isinstance(target_pose, torch.Tensor):
            return target_pose.repeat(n_solutions, 1)
        return np.tile(target_pose, (n_solutions, 1))
    return target_pose


def pose_errors(
    poses_1: PT_NP_TYPE, poses_2: PT_NP_TYPE, acos_epsilon: Optional[float] = None
) -> Tuple[PT_NP_TYPE, PT_NP_TYPE]:
    """L2 positional and geodesic angular error between two [n x 7] pose batches (evaluation_utils.py:37-51)."""
    assert poses_1.shape == poses_2.shape, f"Poses are of different shape: {poses_1.shape} != {poses_2.shape}"
    dev = poses_1.device if isinstance(poses_1, torch.Tensor) and poses_1.is_cuda else _device()
    a, b = _to_dev(poses_1, dev), _to_dev(poses_2, dev)
    assert a.ndim == 2 and a.shape[1] == 7, f"poses must be [n x 7], got {tuple(a.shape)}"
    n = a.shape[0]
    pe = torch.empty(n, dtype=torch.float32, device=dev)
    re = torch.empty_like(pe)
    with torch.cuda.device(dev):
        _check(_lib.load().ikf_pose_distance(a.data_ptr(), b.data_ptr(), n, -1.0 if acos_epsilon is None else float(acos_epsilon),
                                             pe.data_ptr(), re.data_ptr(), _stream()))
    return _back(pe, poses_1), _back(re, poses_1)


def pose_errors_cm_deg(
    poses_1: PT_NP_TYPE, poses_2: PT_NP_TYPE, acos_epsilon: Optional[float] = None
) -> Tuple[PT_NP_TYPE, PT_NP_TYPE]:
    """Same in centimetres and degrees (evaluation_utils.py:54-62)."""
    assert poses_1.shape == poses_2.shape, f"Poses are of different shape: {poses_1.shape} != {poses_2.shape}"
    l2, ang = pose_errors(poses_1, poses_2, acos_epsilon=acos_epsilon)
    if isinstance(poses_1, torch.Tensor):
        return 100 * l2, torch.rad2deg(ang)
    return 100 * l2, np.rad2deg(ang)


def solution_pose_errors(robot: Robot, solutions: torch.Tensor, target_poses: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """FK of every solution against its target pose ([7] or [n x 7]) (evaluation_utils.py:65-97)."""
    assert isinstance(target_poses, torch.Tensor), f"target_poses must be a torch.Tensor (got {type(target_poses)})"
    assert isinstance(solutions, torch.Tensor), f"solutions must be a torch.Tensor (got {type(solutions)})"
    target_poses = _get_target_pose_batch(target_poses, solutions.shape[0])
    dev = solutions.device if solutions.is_cuda else _device()
    eng = kinematics_engine_for(robot, dev)
    pe, re = eng.pose_error(solutions[:, 0 : robot.ndof].to(dev), target_poses.to(dev))
    return pe.to(solutions.device), re.to(solutions.device)


def calculate_joint_limits_exceeded(configs: torch.Tensor, joint_limits: List[Tuple[float, float]]) -> torch.Tensor:
    """[batch] bools: any joint strictly outside its (lower, upper) (evaluation_utils.py:100-112)."""
    assert configs.ndim == 2 and configs.shape[1] == len(joint_limits), (
        f"configs must be [batch x {len(joint_limits)}], got {tuple(configs.shape)}"
    )
    dev = configs.device if configs.is_cuda else _device()
    q = _to_dev(configs, dev)
    n, ncols = q.shape
    lo = (C.c_float * ncols)(*[float(np.float32(x[0])) for x in joint_limits])
    hi = (C.c_float * ncols)(*[float(np.float32(x[1])) for x in joint_limits])
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _check(_lib.load().ikf_limits_exceeded(q.data_ptr(), n, ncols, C.cast(lo, C.c_void_p), C.cast(hi, C.c_void_p),
                                               out.data_ptr(), _stream()))
    return out.to(torch.bool).to(configs.device)


def calculate_self_collisions(robot: Robot, configs: torch.Tensor) -> torch.Tensor:
    """[batch] bools from the robot's capsule model (evaluation_utils.py:115-126 asks jrl / Klampt per configuration).
    The built-in robots carry no collision geometry: attach one with ``Robot.set_collision_capsules`` first."""
    if not robot.has_collision_model:
        raise NotImplementedError(
            "self-collision checking needs a collision model (jrl's is not available here): Robot.set_collision_capsules(...)"
        )
    dev = configs.device if configs.is_cuda else _device()
    return robot.config_self_collides(configs.to(dev)).to(configs.device)


def evaluate_solutions(robot: Robot, target_poses: PT_NP_TYPE, solutions: torch.Tensor):
    """(l2_errors, angular_errors, joint_limits_exceeded, self_collisions) - evaluation_utils.py:130-147; the
    self-collision slot is ``None`` (see module docstring)."""
    assert isinstance(target_poses, torch.Tensor), f"target_poses must be a torch.Tensor (got {type(target_poses)})"
    assert isinstance(solutions, torch.Tensor), f"solutions must be a torch.Tensor (got {type(solutions)})"
    target_poses = _get_target_pose_batch(target_poses, solutions.shape[0])
    l2_errors, angular_errors = solution_pose_errors(robot, solutions, target_poses)
    joint_limits_exceeded = calculate_joint_limits_exceeded(solutions, robot.actuated_joints_limits)
    self_collisions = calculate_self_collisions(robot, solutions) if robot.has_collision_model else None
    return l2_errors, angular_errors, joint_limits_exceeded, self_collisions
