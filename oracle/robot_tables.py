"""ORACLE (test infrastructure - never imported by the product path in ikflow_amd/; imports nothing from ikflow_amd).

Kinematic chains of the robots the released models use, written down a second time and in a different form from the
product's tables (ikflow_amd/robots.py): here they are the <joint> elements of the robots' public URDFs, kept as XML text
and parsed with xml.etree.  A typing error in either table shows up as an FK mismatch between the HIP path and this
oracle (tests/test_gpu_parity.py) and as a table mismatch in tests/test_oracle_independence.py.

The reference takes its robots from jrl (git 2ba7c39, absent from /root/reference); what it pins itself:
  Panda joint limits                                   /root/reference/tests/model_test.py:27-44
  Panda FK(q=0) = [0.088,0,0.926, 0,0.92387953,0.38268343,0]   /root/reference/tests/evaluation_utils_test.py:20-24
Fetch / FetchArm are "parity unpinned" (public fetch_description URDF, restated from memory).
"""
from __future__ import annotations

import math
import xml.etree.ElementTree as ET
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

FIXED, REVOLUTE, PRISMATIC = 0, 1, 2
_KIND = {"fixed": FIXED, "revolute": REVOLUTE, "continuous": REVOLUTE, "prismatic": PRISMATIC}

# franka_description panda_arm + hand (panda_link0 -> panda_hand)
PANDA_URDF = """
<robot name="panda">
  <joint name="panda_joint1" type="revolute"><origin xyz="0 0 0.333" rpy="0 0 0"/><axis xyz="0 0 1"/>
    <limit lower="-2.8973" upper="2.8973"/></joint>
  <joint name="panda_joint2" type="revolute"><origin xyz="0 0 0" rpy="-1.5707963267948966 0 0"/><axis xyz="0 0 1"/>
    <limit lower="-1.7628" upper="1.7628"/></joint>
  <joint name="panda_joint3" type="revolute"><origin xyz="0 -0.316 0" rpy="1.5707963267948966 0 0"/><axis xyz="0 0 1"/>
    <limit lower="-2.8973" upper="2.8973"/></joint>
  <joint name="panda_joint4" type="revolute"><origin xyz="0.0825 0 0" rpy="1.5707963267948966 0 0"/><axis xyz="0 0 1"/>
    <limit lower="-3.0718" upper="-0.0698"/></joint>
  <joint name="panda_joint5" type="revolute"><origin xyz="-0.0825 0.384 0" rpy="-1.5707963267948966 0 0"/><axis xyz="0 0 1"/>
    <limit lower="-2.8973" upper="2.8973"/></joint>
  <joint name="panda_joint6" type="revolute"><origin xyz="0 0 0" rpy="1.5707963267948966 0 0"/><axis xyz="0 0 1"/>
    <limit lower="-0.0175" upper="3.7525"/></joint>
  <joint name="panda_joint7" type="revolute"><origin xyz="0.088 0 0" rpy="1.5707963267948966 0 0"/><axis xyz="0 0 1"/>
    <limit lower="-2.8973" upper="2.8973"/></joint>
  <joint name="panda_joint8" type="fixed"><origin xyz="0 0 0.107" rpy="0 0 0"/></joint>
  <joint name="panda_hand_joint" type="fixed"><origin xyz="0 0 0" rpy="0 0 -0.7853981633974483"/></joint>
</robot>
"""

# fetch_description fetch.urdf, base_link -> gripper_link (continuous joints get [-pi, pi])
FETCH_URDF = """
<robot name="fetch">
  <joint name="torso_lift_joint" type="prismatic"><origin xyz="-0.086875 0 0.37743" rpy="0 0 0"/><axis xyz="0 0 1"/>
    <limit lower="0.0" upper="0.38615"/></joint>
  <joint name="shoulder_pan_joint" type="revolute"><origin xyz="0.119525 0 0.34858" rpy="0 0 0"/><axis xyz="0 0 1"/>
    <limit lower="-1.6056" upper="1.6056"/></joint>
  <joint name="shoulder_lift_joint" type="revolute"><origin xyz="0.117 0 0.06" rpy="0 0 0"/><axis xyz="0 1 0"/>
    <limit lower="-1.221" upper="1.518"/></joint>
  <joint name="upperarm_roll_joint" type="continuous"><origin xyz="0.219 0 0" rpy="0 0 0"/><axis xyz="1 0 0"/></joint>
  <joint name="elbow_flex_joint" type="revolute"><origin xyz="0.133 0 0" rpy="0 0 0"/><axis xyz="0 1 0"/>
    <limit lower="-2.251" upper="2.251"/></joint>
  <joint name="forearm_roll_joint" type="continuous"><origin xyz="0.197 0 0" rpy="0 0 0"/><axis xyz="1 0 0"/></joint>
  <joint name="wrist_flex_joint" type="revolute"><origin xyz="0.1245 0 0" rpy="0 0 0"/><axis xyz="0 1 0"/>
    <limit lower="-2.16" upper="2.16"/></joint>
  <joint name="wrist_roll_joint" type="continuous"><origin xyz="0.1385 0 0" rpy="0 0 0"/><axis xyz="1 0 0"/></joint>
  <joint name="gripper_axis" type="fixed"><origin xyz="0.16645 0 0" rpy="0 0 0"/></joint>
</robot>
"""


@dataclass(frozen=True)
class OJoint:
    name: str
    kind: int
    origin_xyz: Tuple[float, float, float]
    origin_rpy: Tuple[float, float, float]
    axis: Tuple[float, float, float]
    limits: Optional[Tuple[float, float]]

    @property
    def actuated(self) -> bool:
        return self.kind != FIXED


def _floats(s: str) -> Tuple[float, ...]:
    return tuple(float(v) for v in s.split())


def parse_chain(urdf_text: str, skip: Tuple[str, ...] = ()) -> List[OJoint]:
    joints = []
    for j in ET.fromstring(urdf_text).findall("joint"):
        if j.get("name") in skip:
            continue
        origin = j.find("origin")
        axis = j.find("axis")
        limit = j.find("limit")
        typ = j.get("type")
        lim = None
        if typ == "continuous":
            lim = (-math.pi, math.pi)
        elif limit is not None:
            lim = (float(limit.get("lower")), float(limit.get("upper")))
        joints.append(
            OJoint(
                name=j.get("name"),
                kind=_KIND[typ],
                origin_xyz=_floats(origin.get("xyz")),
                origin_rpy=_floats(origin.get("rpy")),
                axis=_floats(axis.get("xyz")) if axis is not None else (0.0, 0.0, 1.0),
                limits=lim,
            )
        )
    return joints


def rpy_matrix(rpy) -> np.ndarray:
    """URDF fixed-axis roll-pitch-yaw as the product of three elementary rotations, Rz(yaw) @ Ry(pitch) @ Rx(roll)."""
    r, p, y = (float(v) for v in rpy)
    Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]], dtype=np.float64)
    Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]], dtype=np.float64)
    Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]], dtype=np.float64)
    return Rz @ Ry @ Rx


class OracleRobot:
    """What the oracle needs from a jrl.Robot: name, ndof, actuated_joints_limits and the joint chain."""

    def __init__(self, name: str, joints: List[OJoint]):
        self.name = name
        self.joints = tuple(joints)
        self.actuated_joints_limits = [tuple(j.limits) for j in joints if j.actuated]
        self.ndof = len(self.actuated_joints_limits)

    def sample_joint_angles(self, n: int, joint_limit_eps: float = 0.0, rng: Optional[np.random.Generator] = None) -> np.ndarray:
        """Uniform inside the limits shrunk by eps (/root/reference/scripts/build_dataset.py:186 uses eps = 0.25 deg)."""
        rng = np.random.default_rng(0) if rng is None else rng
        lo = np.array([l[0] for l in self.actuated_joints_limits]) + joint_limit_eps
        hi = np.array([l[1] for l in self.actuated_joints_limits]) - joint_limit_eps
        return (lo + (hi - lo) * rng.random((n, self.ndof))).astype(np.float32)


def robot(name: str) -> OracleRobot:
    if name == "panda":
        return OracleRobot("panda", parse_chain(PANDA_URDF))
    if name == "fetch":
        return OracleRobot("fetch", parse_chain(FETCH_URDF))
    if name == "fetch_arm":  # jrl FetchArm: the 7-joint arm, base = torso_lift_link (the torso joint is not on the chain)
        return OracleRobot("fetch_arm", parse_chain(FETCH_URDF, skip=("torso_lift_joint",)))
    raise ValueError(f"oracle has no chain for robot '{name}'")
