"""The per-row arithmetic of the kinematics kernels (ikflow_amd/csrc/kin_math.h: chain walk, quaternion, pose error, the LM step in both of its
arithmetics) compiled for the HOST with g++ and held against the oracle - the kernels' own source, checked without a GPU.  The GPU tests check the
same code where it ships (tests/test_gpu_parity.py); this one pins its arithmetic on every CPU run, and is how round 6 told a compiler effect
(fused multiply-adds in the fp32 LU elimination on the device) from an algorithmic one.  Test infrastructure: nothing in ikflow_amd/ loads it."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from helpers import reachable_poses
from ikflow_amd.engine import fold_chain
from ikflow_amd.robots import get_robot
from oracle import kinematics_oracle as ko

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("kin_math") / "libkin_math_host.so"
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-Wno-unknown-pragmas", os.path.join(ROOT, "tests", "kin_math_host.cpp"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = C.CDLL(str(out))
    lib.kin_math_host.restype = C.c_int
    lib.kin_math_host.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]
    return lib


def _chain_bytes(robot, lib):
    joints, tool = fold_chain(robot)
    b = struct.pack("i", robot.ndof)
    for j in range(8):
        if j < len(joints):
            kind, ax, pre = joints[j]
            b += struct.pack("i", kind) + np.asarray(ax, dtype=np.float32).tobytes() + np.asarray(pre, dtype=np.float32).reshape(-1).tobytes()
        else:
            b += bytes(4 + 12 + 48)
    b += np.asarray(tool, dtype=np.float32).reshape(-1).tobytes()
    lo, hi = np.zeros(8, np.float32), np.zeros(8, np.float32)
    for i, (l, h) in enumerate(robot.actuated_joints_limits):
        lo[i], hi[i] = l, h
    b += lo.tobytes() + hi.tobytes()
    assert len(b) == lib.kin_math_chain_bytes()
    return C.create_string_buffer(b, len(b))


def _call(lib, chain, what, tgt, q, out_cols, two=False):
    n = q.shape[0]
    q = np.ascontiguousarray(q, dtype=np.float32)
    tgt = np.ascontiguousarray(tgt, dtype=np.float32) if tgt is not None else np.zeros((n, 7), np.float32)
    out = np.zeros((n, out_cols) if out_cols > 1 else (n,), np.float32)
    out2 = np.zeros(n, np.float32)
    assert lib.kin_math_host(chain, what, tgt.ctypes.data, q.ctypes.data, n, out.ctypes.data, out2.ctypes.data) == 0
    return (out, out2) if two else out


@pytest.mark.parametrize("which", ["panda", "fetch", "fetch_arm"])
def test_fk_and_pose_error_of_the_kernel_source_on_the_host(host_lib, which):
    robot = get_robot(which)
    chain = _chain_bytes(robot, host_lib)
    n = 2000
    q_true, poses = reachable_poses(robot, n, 11)
    fk = _call(host_lib, chain, 0, None, q_true.numpy(), 7)
    ref = ko.forward_kinematics(robot, q_true.double()).numpy()
    sign = np.sign((fk[:, 3:] * ref[:, 3:]).sum(1))[:, None]     # (q and -q are the same rotation)
    assert np.abs(fk[:, :3] - ref[:, :3]).max() <= 2e-6 and np.abs(fk[:, 3:] * sign - ref[:, 3:]).max() <= 2e-6
    seeds = ko.clamp_to_joint_limits(robot, q_true + 0.05 * torch.randn(q_true.shape, generator=torch.Generator().manual_seed(2)))
    pe, re = _call(host_lib, chain, 1, poses.numpy(), seeds.numpy(), 1, two=True)
    rpe, rre = ko.calculate_pose_error(robot, seeds, poses)
    assert np.abs(pe - rpe.numpy()).max() <= 2e-6 and np.abs(re - rre.numpy()).max() <= 3e-5


def test_lm_step_of_the_kernel_source_on_the_host_in_both_arithmetics(host_lib):
    """fp64 inside: within 5e-6 of the oracle's fp64 step on every row.  fp32 (the reference's arithmetic): the reference's own noise - in units of
    cond x 2^-24 x |dq| as close to the fp64 truth as the oracle's fp32 step (torch: bmm + sgesv), the rows next to a singularity included."""
    robot = get_robot("panda")
    chain = _chain_bytes(robot, host_lib)
    n = 4096
    q_true, poses = reachable_poses(robot, n, 0)
    seeds = ko.clamp_to_joint_limits(robot, q_true + 0.05 * torch.randn(q_true.shape, generator=torch.Generator().manual_seed(3)))
    ref64 = ko.lm_step(robot, poses.double(), seeds.double()).numpy()
    ref32 = ko.lm_step(robot, poses, seeds).numpy()
    J = ko.jacobian(robot, seeds.double())
    cond = torch.linalg.cond(J.transpose(1, 2) @ J + 1e-4 * torch.eye(7, dtype=torch.float64)).numpy()
    unit = cond * 2.0 ** -24 * np.maximum(np.abs(ref64 - seeds.numpy()).max(1), 1e-3)
    got64 = _call(host_lib, chain, 3, poses.numpy(), seeds.numpy(), 7)
    got32 = _call(host_lib, chain, 2, poses.numpy(), seeds.numpy(), 7)
    assert np.abs(got64 - ref64).max() <= 5e-6
    e32, eo = np.abs(got32 - ref64).max(1) / unit, np.abs(ref32 - ref64).max(1) / unit
    q = lambda e: (float(np.median(e)), float(np.quantile(e, 0.99)), float(e.max()))
    print("host build of the kernel source, fp32 step vs truth (median, p99, max in units of cond eps |dq|):", q(e32), " oracle fp32:", q(eo))
    assert cond.min() >= 1e3                                   # (rank-6 J^T J: no well-conditioned pose exists on a 7-joint arm)
    assert q(e32)[0] <= 1.5 * q(eo)[0] and q(e32)[1] <= 1.5 * q(eo)[1] and q(e32)[2] <= 4.0 and q(eo)[2] <= 4.0
