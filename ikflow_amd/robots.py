"""Robot descriptions for the IK hot path: serial kinematic chains + joint limits.

The reference takes these from the third-party package ``jrl`` (pinned at git 2ba7c39, not in
/root/reference, not installed): call sites ``ikflow/ikflow_solver.py:102,114,205,208``,
``ikflow/model.py:314``, ``ikflow/evaluation_utils.py:86``.  What is restated here is only what the hot
path reads from a ``jrl.Robot``: ``name``, ``ndof``, ``actuated_joints_limits`` and the URDF chain from
the base link to the end-effector link.  The chains come from the robots' public URDFs.

Pins held by the reference's own tests:
  * Panda joint limits                         tests/model_test.py:27-44
  * Panda FK(q=0) = [0.088,0,0.926, 0,0.92387953,0.38268343,0]   tests/evaluation_utils_test.py:20-24
    (reproduced only with ``panda_hand`` as the end-effector link - see tests/test_oracle_golden.py)
FetchArm / Fetch are restated from the public Fetch URDF and are NOT pinned by any reference test.

The numerical methods of a Robot (FK, Jacobian, LM step, clamping) are HIP kernels reached through the
C-ABI (``ikflow_amd/csrc``); this module only holds the description and forwards to the engine.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

JOINT_FIXED = 0
JOINT_REVOLUTE = 1
JOINT_PRISMATIC = 2


@dataclass(frozen=True)
class Joint:
    """One URDF joint on the base->end-effector chain: parent_T_child = T(origin_xyz, origin_rpy) * motion(q)."""

    name: str
    kind: int  # JOINT_FIXED / JOINT_REVOLUTE / JOINT_PRISMATIC
    origin_xyz: Tuple[float, float, float]
    origin_rpy: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    axis: Tuple[float, float, float] = (0.0, 0.0, 1.0)
    limits: Optional[Tuple[float, float]] = None  # (lower, upper), actuated joints only

    @property
    def actuated(self) -> bool:
        return self.kind != JOINT_FIXED


def rpy_to_matrix(rpy: Sequence[float]) -> np.ndarray:
    """URDF fixed-axis roll/pitch/yaw -> 3x3 rotation (R = Rz(yaw) Ry(pitch) Rx(roll)), float64."""
    r, p, y = (float(v) for v in rpy)
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array(
        [
            [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
            [-sp, cp * sr, cp * cr],
        ],
        dtype=np.float64,
    )


class Robot:
    """Description of a serial arm + the jrl.Robot methods the hot path calls.

    Mirrors the part of ``jrl.robot.Robot`` that ``ikflow`` touches (see module docstring).  Tensor
    methods run on the GPU engine (``ikflow_amd._lib``); they never fall back to CPU arithmetic.
    """

    def __init__(self, name: str, joints: Sequence[Joint]):
        self._name = name
        self._joints: Tuple[Joint, ...] = tuple(joints)
        lims = [j.limits for j in self._joints if j.actuated]
        assert all(l is not None for l in lims), "every actuated joint needs limits"
        self._limits: List[Tuple[float, float]] = [(float(l[0]), float(l[1])) for l in lims]

    # -- construction from data -----------------------------------------------------------------------
    @classmethod
    def from_urdf(cls, path_or_xml: str, base_link: Optional[str] = None, end_effector_link: Optional[str] = None,
                  name: Optional[str] = None, continuous_limits: Tuple[float, float] = (-math.pi, math.pi)) -> "Robot":
        """Build the base->end-effector chain of a robot from its URDF (a file path or the XML text): a new robot is data,
        not code.  This is what ``jrl.Robot`` does with the URDFs it ships (``jrl/robots.py``; the package is absent here).

        The kinematic tree is read from every ``<joint>``'s ``<parent link>`` / ``<child link>``; the chain is the unique
        path from ``end_effector_link`` up to ``base_link`` (joints on other branches - fingers, head, wheels - are not on
        it).  Per joint: ``<origin xyz rpy>`` (missing = identity), ``<axis xyz>`` (missing = 1 0 0, the URDF default),
        ``<limit lower upper>``; ``continuous`` joints get ``continuous_limits`` (jrl: [-pi, pi]); ``revolute`` /
        ``prismatic`` joints are actuated, ``fixed`` joints stay on the chain and are folded into their neighbours by the
        engine (``ikflow_amd.engine.fold_chain``).  ``mimic`` joints on the chain and ``floating`` / ``planar`` joints are
        refused: the engine has one scalar per actuated joint.

        When NO joint of the document names a parent / child (a bare list of ``<joint>`` elements, as test fixtures keep
        them) and both link arguments are None, the document order is the chain."""
        import os
        import xml.etree.ElementTree as ET

        text = path_or_xml
        if "<" not in path_or_xml:
            assert os.path.isfile(path_or_xml), f"URDF file '{path_or_xml}' was not found"
            with open(path_or_xml, "r") as f:
                text = f.read()
        root = ET.fromstring(text)
        assert root.tag == "robot", f"not a URDF: the root element is <{root.tag}>, expected <robot>"
        kinds = {"fixed": JOINT_FIXED, "revolute": JOINT_REVOLUTE, "continuous": JOINT_REVOLUTE, "prismatic": JOINT_PRISMATIC}

        def floats(s: Optional[str], default):
            if s is None:
                return tuple(default)
            v = tuple(float(x) for x in s.split())
            assert len(v) == 3, f"expected three numbers, got '{s}'"
            return v

        def make(j) -> Joint:
            jname, typ = j.get("name"), j.get("type")
            if typ not in kinds:
                raise ValueError(f"joint '{jname}' has type '{typ}': only fixed / revolute / continuous / prismatic joints can be on the chain")
            if j.find("mimic") is not None and typ != "fixed":
                raise ValueError(f"joint '{jname}' mimics another joint: not supported on the base->end-effector chain")
            origin, axis, limit = j.find("origin"), j.find("axis"), j.find("limit")
            lim = None
            if typ == "continuous":
                lim = (float(continuous_limits[0]), float(continuous_limits[1]))
            elif typ in ("revolute", "prismatic"):
                if limit is None or limit.get("lower") is None or limit.get("upper") is None:
                    raise ValueError(f"joint '{jname}' ({typ}) has no <limit lower=.. upper=..>")
                lim = (float(limit.get("lower")), float(limit.get("upper")))
                assert lim[0] <= lim[1], f"joint '{jname}': lower limit above upper limit"
            ax = floats(axis.get("xyz") if axis is not None else None, (1.0, 0.0, 0.0))
            if typ != "fixed":
                assert any(abs(a) > 0 for a in ax), f"joint '{jname}' has a zero axis"
            return Joint(jname, kinds[typ], floats(origin.get("xyz") if origin is not None else None, (0.0, 0.0, 0.0)),
                         floats(origin.get("rpy") if origin is not None else None, (0.0, 0.0, 0.0)),
                         ax if typ != "fixed" else (0.0, 0.0, 1.0), lim)

        jelems = root.findall("joint")
        assert jelems, "the URDF has no <joint> elements"
        has_tree = any(j.find("parent") is not None or j.find("child") is not None for j in jelems)
        if not has_tree:
            assert base_link is None and end_effector_link is None, (
                "this URDF names no parent / child links, so base_link / end_effector_link cannot be resolved")
            chain = [make(j) for j in jelems]
        else:
            assert base_link is not None and end_effector_link is not None, "base_link and end_effector_link are required"
            by_child = {}
            for j in jelems:
                par, chi = j.find("parent"), j.find("child")
                assert par is not None and chi is not None, f"joint '{j.get('name')}' lacks <parent> or <child>"
                c = chi.get("link")
                assert c not in by_child, f"link '{c}' is the child of two joints: not a tree"
                by_child[c] = (par.get("link"), j)
            links = {l.get("name") for l in root.findall("link")} | set(by_child) | {p for p, _ in by_child.values()}
            assert base_link in links, f"base_link '{base_link}' is not a link of this URDF"
            assert end_effector_link in links, f"end_effector_link '{end_effector_link}' is not a link of this URDF"
            rev, link, seen = [], end_effector_link, set()
            while link != base_link:
                if link not in by_child or link in seen:
                    raise ValueError(f"'{end_effector_link}' is not a descendant of '{base_link}' in this URDF")
                seen.add(link)
                parent, j = by_child[link]
                rev.append(make(j))
                link = parent
            chain = rev[::-1]
        robot = cls(name if name is not None else root.get("name", "robot"), chain)
        assert robot.ndof >= 1, "the chain has no actuated joint"
        return robot

    # -- description ---------------------------------------------------------------------------------
    @property
    def name(self) -> str:
        return self._name

    @property
    def ndof(self) -> int:
        return len(self._limits)

    @property
    def actuated_joints_limits(self) -> List[Tuple[float, float]]:
        return list(self._limits)

    @property
    def joints(self) -> Tuple[Joint, ...]:
        return self._joints

    def __str__(self) -> str:
        return f"<Robot[{self._name}] ndof={self.ndof}>"

    def chain_table(self) -> np.ndarray:
        """Pack the chain for the C-ABI: one row of 16 float32 per joint.

        row = [kind, ax, ay, az, R00..R02, tx, R10..R12, ty, R20..R22, tz]  (fixed origin transform 3x4)
        """
        rows = np.zeros((len(self._joints), 16), dtype=np.float32)
        for i, j in enumerate(self._joints):
            R = rpy_to_matrix(j.origin_rpy)
            ax = np.asarray(j.axis, dtype=np.float64)
            n = np.linalg.norm(ax)
            ax = ax / n if n > 0 else ax
            rows[i, 0] = float(j.kind)
            rows[i, 1:4] = ax
            for r in range(3):
                rows[i, 4 + 4 * r : 7 + 4 * r] = R[r]
                rows[i, 7 + 4 * r] = j.origin_xyz[r]
        return rows

    # -- sampling helpers (host side, numpy; used by tests/bench to make reachable target poses) ------
    def sample_joint_angles(self, n: int, joint_limit_eps: float = 0.0, rng: Optional[np.random.Generator] = None):
        """Uniform joint samples inside the limits shrunk by joint_limit_eps
        (dataset convention of scripts/build_dataset.py:186: eps = deg2rad(0.25))."""
        rng = np.random.default_rng(0) if rng is None else rng
        lo = np.array([l[0] for l in self._limits]) + joint_limit_eps
        hi = np.array([l[1] for l in self._limits]) - joint_limit_eps
        return (lo + (hi - lo) * rng.random((n, self.ndof))).astype(np.float32)

    def sample_joint_angles_and_poses(self, n: int, joint_limit_eps: float = 0.0, only_non_self_colliding: bool = False,
                                      tqdm_enabled: bool = False, return_torch: bool = False, device=None,
                                      rng: Optional[np.random.Generator] = None):
        """(q [n x ndof], poses [n x 7]) like jrl.Robot.sample_joint_angles_and_poses, which the reference's tests and
        harnesses draw their target poses from (tests/ikflow_solver_test.py:73-75, scripts/benchmark_runtime.py).  FK runs
        on the GPU engine; numpy arrays come back unless return_torch.  only_non_self_colliding needs a collision model
        (Robot.set_collision_capsules / use_approximate_collision_model): colliding samples are redrawn."""
        import torch

        from ikflow_amd.engine import kinematics_engine_for

        eng = kinematics_engine_for(self, device)
        rng = np.random.default_rng() if rng is None else rng
        q = torch.tensor(self.sample_joint_angles(n, joint_limit_eps, rng), device=eng.device)
        if only_non_self_colliding:
            assert self.has_collision_model, "only_non_self_colliding needs a collision model (Robot.set_collision_capsules)"
            for _ in range(64):
                bad = self.config_self_collides(q)
                k = int(bad.sum().item())
                if k == 0:
                    break
                q[bad] = torch.tensor(self.sample_joint_angles(k, joint_limit_eps, rng), device=eng.device)
        poses = eng.forward_kinematics(q)
        if return_torch:
            return q, poses
        return q.cpu().numpy(), poses.cpu().numpy()

    # -- engine-backed tensor methods (jrl.Robot API surface used by ikflow) --------------------------
    def _engine(self, like):
        from ikflow_amd.engine import kinematics_engine_for

        return kinematics_engine_for(self, getattr(like, "device", None))

    def forward_kinematics(self, q):
        """[n x ndof] joint angles -> [n x 7] poses (x,y,z,qw,qx,qy,qz). ikflow_solver.py:114."""
        return self._engine(q).forward_kinematics(q)

    def clamp_to_joint_limits(self, q):
        """Per-joint clamp to [lower, upper]; returns a new tensor. ikflow_solver.py:101-102."""
        return self._engine(q).clamp_to_joint_limits(q)

    def inverse_kinematics_step_levenburg_marquardt(self, target_poses, q):
        """One damped-least-squares step (lambda=1e-4, alpha=1, clamped). ikflow_solver.py:205,208."""
        return self._engine(q).lm_step(target_poses, q)

    def jacobian(self, q):
        """[n x ndof] -> [n x 6 x ndof] geometric Jacobian, rows = [angular(3); linear(3)]."""
        return self._engine(q).jacobian(q)

    # -- capsule self-collision model (jrl.Robot.config_self_collides / self_collision_distances mechanism) ---------
    def set_collision_capsules(self, capsules, ignored_pairs=()):
        """Attach a capsule collision model.  No geometry ships with the built-in robots: jrl's capsule tables are not
        part of this repository, so the caller supplies them.

        capsules: sequence of ``(after_joint, p0, p1, radius)`` - a segment p0-p1 (metres) with a radius, expressed in the
        URDF frame of the child link of joint ``after_joint`` (a joint name of this chain, actuated or fixed), or
        ``None`` for the base link.  ignored_pairs: capsule index pairs that are never tested (adjacent links);
        capsules riding on the same moving frame are never tested against each other.
        """
        names = [j.name for j in self._joints]
        folded = []
        for after, p0, p1, radius in capsules:
            T = np.eye(4)
            frame = 0
            if after is not None:
                assert after in names, f"unknown joint '{after}' (chain joints: {names})"
                for j in self._joints[: names.index(after) + 1]:
                    if j.actuated:
                        frame += 1
                        T = np.eye(4)  # the engine frame follows the joint's motion; later fixed origins accumulate below
                    else:
                        F = np.eye(4)
                        F[:3, :3] = rpy_to_matrix(j.origin_rpy)
                        F[:3, 3] = j.origin_xyz
                        T = T @ F
            a = (T @ np.array([*p0, 1.0], dtype=np.float64))[:3]
            b = (T @ np.array([*p1, 1.0], dtype=np.float64))[:3]
            folded.append((frame, tuple(a), tuple(b), float(radius)))
        ignored = {tuple(sorted((int(a), int(b)))) for a, b in ignored_pairs}
        pairs = [(a, b) for a in range(len(folded)) for b in range(a + 1, len(folded))
                 if folded[a][0] != folded[b][0] and (a, b) not in ignored]
        self._collision_model = (folded, pairs)
        return self

    def use_approximate_collision_model(self):
        """Attach this repository's own approximate capsule model (Panda only; see PANDA_APPROX_CAPSULES - not jrl's
        geometry, no parity claim).  Capsule pairs on adjacent moving frames are skipped as well as the listed ones."""
        if self._name != "panda":
            raise ValueError(f"no approximate capsule model ships for '{self._name}' (only for 'panda')")
        self.set_collision_capsules(PANDA_APPROX_CAPSULES, PANDA_APPROX_IGNORED)
        folded, pairs = self._collision_model
        self._collision_model = (folded, [(a, b) for a, b in pairs if abs(folded[a][0] - folded[b][0]) > 1])
        return self

    @property
    def has_collision_model(self) -> bool:
        return getattr(self, "_collision_model", None) is not None

    def _collision_engine(self, q):
        assert self.has_collision_model, "no capsule collision model: call Robot.set_collision_capsules(...) first"
        eng = self._engine(q)
        if getattr(eng, "_collision_source", None) is not self._collision_model:
            eng.set_collision_model(*self._collision_model)
            eng._collision_source = self._collision_model
        return eng

    def self_collision_distances(self, q):
        """[n x ndof] -> [n] signed clearance of the closest tested capsule pair (negative = overlapping)."""
        return self._collision_engine(q).self_collision(q)[0]

    def config_self_collides(self, q):
        """[n x ndof] -> [n] bool (one configuration [ndof] -> Python bool, like jrl.Robot.config_self_collides)."""
        import torch

        single = getattr(q, "ndim", 2) == 1
        qq = q.reshape(1, -1) if single else q
        col = self._collision_engine(qq).self_collision(qq)[1]
        return bool(col[0].item()) if single else col


_HALF_PI = math.pi / 2.0


def Panda() -> Robot:
    """Franka Emika Panda, base ``panda_link0`` -> end effector ``panda_hand`` (public franka_description URDF).

    Limits: reference tests/model_test.py:27-44.
    """
    z = (0.0, 0.0, 1.0)
    return Robot(
        "panda",
        [
            Joint("panda_joint1", JOINT_REVOLUTE, (0.0, 0.0, 0.333), (0.0, 0.0, 0.0), z, (-2.8973, 2.8973)),
            Joint("panda_joint2", JOINT_REVOLUTE, (0.0, 0.0, 0.0), (-_HALF_PI, 0.0, 0.0), z, (-1.7628, 1.7628)),
            Joint("panda_joint3", JOINT_REVOLUTE, (0.0, -0.316, 0.0), (_HALF_PI, 0.0, 0.0), z, (-2.8973, 2.8973)),
            Joint("panda_joint4", JOINT_REVOLUTE, (0.0825, 0.0, 0.0), (_HALF_PI, 0.0, 0.0), z, (-3.0718, -0.0698)),
            Joint("panda_joint5", JOINT_REVOLUTE, (-0.0825, 0.384, 0.0), (-_HALF_PI, 0.0, 0.0), z, (-2.8973, 2.8973)),
            Joint("panda_joint6", JOINT_REVOLUTE, (0.0, 0.0, 0.0), (_HALF_PI, 0.0, 0.0), z, (-0.0175, 3.7525)),
            Joint("panda_joint7", JOINT_REVOLUTE, (0.088, 0.0, 0.0), (_HALF_PI, 0.0, 0.0), z, (-2.8973, 2.8973)),
            Joint("panda_joint8", JOINT_FIXED, (0.0, 0.0, 0.107)),
            Joint("panda_hand_joint", JOINT_FIXED, (0.0, 0.0, 0.0), (0.0, 0.0, -math.pi / 4.0)),
        ],
    )


# An APPROXIMATE capsule model of the Panda, written for this repository from the public link dimensions (segments along
# the link bodies between successive joint origins, radii of the link housings).  It is NOT jrl's capsule table and NOT the
# Klampt mesh check the reference runs (ikflow/evaluation_utils.py:115-126): flags produced with it are an engineering
# estimate, good for filtering gross self-intersections, and carry no parity claim.  (after_joint, p0, p1, radius) in the
# child-link frame of `after_joint`; None = base link.
PANDA_APPROX_CAPSULES = [
    (None, (0.0, 0.0, 0.0), (0.0, 0.0, 0.14), 0.09),                    # 0  link0: base column
    ("panda_joint1", (0.0, 0.0, -0.19), (0.0, 0.0, -0.03), 0.07),        # 1  link1: shoulder column
    ("panda_joint2", (0.0, 0.0, -0.06), (0.0, 0.0, 0.06), 0.07),         # 2  link2: shoulder housing (along the joint axis)
    ("panda_joint2", (0.0, -0.07, 0.0), (0.0, -0.14, 0.0), 0.065),       # 3  link2: lower upper-arm
    ("panda_joint3", (0.0, 0.0, -0.16), (0.0, 0.0, -0.03), 0.065),       # 4  link3: upper upper-arm
    ("panda_joint4", (0.0, 0.0, -0.06), (0.0, 0.0, 0.06), 0.065),        # 5  link4: elbow housing (along the joint axis)
    ("panda_joint4", (-0.0825, 0.07, 0.0), (-0.0825, 0.13, 0.0), 0.06),  # 6  link4: lower forearm
    ("panda_joint5", (0.0, 0.0, -0.24), (0.0, 0.0, -0.07), 0.06),        # 7  link5: forearm
    ("panda_joint6", (0.0, 0.0, 0.0), (0.088, 0.0, 0.0), 0.05),          # 8  link6: wrist
    ("panda_joint7", (0.0, 0.0, 0.02), (0.0, 0.0, 0.09), 0.045),         # 9  link7: flange
    ("panda_hand_joint", (0.0, -0.07, 0.025), (0.0, 0.07, 0.025), 0.035),  # 10 hand body
    ("panda_hand_joint", (0.0, 0.0, 0.06), (0.0, 0.0, 0.10), 0.03),      # 11 fingers
]
# pairs never tested besides capsules on the same or on adjacent moving frames (skipped automatically below): housings
# that sit on coincident joint origins two frames apart
PANDA_APPROX_IGNORED = [(1, 3), (2, 4), (5, 7), (6, 8), (7, 9), (7, 10), (8, 10), (8, 11)]


def _fetch_arm_joints() -> List[Joint]:
    x, y, zz = (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)
    pi = math.pi
    return [
        Joint("shoulder_pan_joint", JOINT_REVOLUTE, (0.119525, 0.0, 0.34858), axis=zz, limits=(-1.6056, 1.6056)),
        Joint("shoulder_lift_joint", JOINT_REVOLUTE, (0.117, 0.0, 0.06), axis=y, limits=(-1.221, 1.518)),
        Joint("upperarm_roll_joint", JOINT_REVOLUTE, (0.219, 0.0, 0.0), axis=x, limits=(-pi, pi)),
        Joint("elbow_flex_joint", JOINT_REVOLUTE, (0.133, 0.0, 0.0), axis=y, limits=(-2.251, 2.251)),
        Joint("forearm_roll_joint", JOINT_REVOLUTE, (0.197, 0.0, 0.0), axis=x, limits=(-pi, pi)),
        Joint("wrist_flex_joint", JOINT_REVOLUTE, (0.1245, 0.0, 0.0), axis=y, limits=(-2.16, 2.16)),
        Joint("wrist_roll_joint", JOINT_REVOLUTE, (0.1385, 0.0, 0.0), axis=x, limits=(-pi, pi)),
        Joint("gripper_axis", JOINT_FIXED, (0.16645, 0.0, 0.0)),
    ]


def FetchArm() -> Robot:
    """Fetch's 7-joint arm, base ``torso_lift_link`` -> ``gripper_link`` (public fetch_description URDF;
    continuous joints limited to [-pi, pi]).  Unpinned against jrl."""
    return Robot("fetch_arm", _fetch_arm_joints())


def Fetch() -> Robot:
    """Fetch with the prismatic torso lift (8 dof), base ``base_link`` -> ``gripper_link``. Unpinned."""
    torso = Joint(
        "torso_lift_joint", JOINT_PRISMATIC, (-0.086875, 0.0, 0.37743), axis=(0.0, 0.0, 1.0), limits=(0.0, 0.38615)
    )
    return Robot("fetch", [torso] + _fetch_arm_joints())


_ROBOTS = {"panda": Panda, "fetch_arm": FetchArm, "fetch": Fetch}

# Robots whose released model is registered (ikflow/model_descriptions.yaml) but whose chain this repository does not
# carry: their URDF lives in jrl, which is not on this machine, and nothing the reference holds pins a chain typed from
# memory.  They are DATA: Robot.from_urdf(<file>, base_link, end_effector_link, name=<robot_name>), or
# register_robot_urdf(...) once so that get_robot / get_ik_solver find them.  Link names as jrl uses them.
URDF_ONLY_ROBOTS = {
    "rizon4": dict(base_link="base_link", end_effector_link="flange", ndof=7),  # Flexiv Rizon 4: 7 revolute joints
}


def register_robot_urdf(robot_name: str, path_or_xml: str, base_link: str, end_effector_link: str) -> None:
    """Make get_robot(robot_name) build the robot from this URDF (e.g. the Flexiv Rizon 4 description for 'rizon4')."""
    _ROBOTS[robot_name] = lambda: Robot.from_urdf(path_or_xml, base_link, end_effector_link, name=robot_name)


def get_robot(robot_name: str) -> Robot:
    """Mirror of ``jrl.robots.get_robot`` for the robots whose released models are on the hot path
    (ikflow/model_loading.py:81-83)."""
    if robot_name not in _ROBOTS:
        if robot_name in URDF_ONLY_ROBOTS:
            h = URDF_ONLY_ROBOTS[robot_name]
            raise ValueError(
                f"Robot '{robot_name}' has no built-in chain (its URDF ships with jrl, which is not available here): build it with "
                f"Robot.from_urdf(<urdf>, '{h['base_link']}', '{h['end_effector_link']}', name='{robot_name}') and pass robot=..., or call "
                f"register_robot_urdf('{robot_name}', <urdf>, '{h['base_link']}', '{h['end_effector_link']}') once")
        raise ValueError(f"Unable to find robot '{robot_name}' (available: {sorted(_ROBOTS)})")
    return _ROBOTS[robot_name]()
