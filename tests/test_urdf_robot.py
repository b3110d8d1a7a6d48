"""Robot.from_urdf: a robot is data.  The oracle keeps the public URDFs' <joint> elements as XML (oracle/robot_tables.py);
here that XML goes through the PRODUCT's URDF reader - once as it is (a bare joint list) and once dressed up as a real URDF
(links, parent / child, branches that are not on the chain) - and must give the built-in tables joint by joint.
Reference: the robots come from jrl (get_robot, ikflow/model_loading.py:81-83); Rizon4 is ikflow/model_descriptions.yaml:90-97."""
import math
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest
import torch

from ikflow_amd.engine import fold_chain
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import MODEL_DESCRIPTIONS, hparams_for, layout_from, random_state_dict
from ikflow_amd.model_loading import get_ik_solver
from ikflow_amd.robots import JOINT_FIXED, Fetch, FetchArm, Panda, Robot, get_robot, register_robot_urdf
from oracle import kinematics_oracle as ko
from oracle import robot_tables as rt

RIZON_MODEL = "rizon4__snowy-brook-208__global_step=2.75M"


def as_full_urdf(joint_xml: str, base: str, skip=(), extra: str = "") -> str:
    """Dress the oracle's bare joint list up as a URDF tree: link i -> joint i -> link i+1 from `base`, plus `extra` branches.
    Returns (text, name of the last link)."""
    root = ET.fromstring(joint_xml)
    out = [f'<robot name="{root.get("name")}">', f'<link name="{base}"/>']
    parent = base
    for j in root.findall("joint"):
        if j.get("name") in skip:
            continue
        child = j.get("name") + "_child"
        j.insert(0, ET.Element("child", {"link": child}))
        j.insert(0, ET.Element("parent", {"link": parent}))
        out.append(f'<link name="{child}"><inertial><mass value="1"/></inertial></link>')
        out.append(ET.tostring(j, encoding="unicode"))
        parent = child
    out.append(extra.replace("$EE", parent).replace("$BASE", base))
    out.append("</robot>")
    return "\n".join(out), parent


BRANCHES = """
<link name="finger_l"/><link name="finger_r"/><link name="head"/><link name="camera"/>
<joint name="finger_joint1" type="prismatic"><parent link="$EE"/><child link="finger_l"/><origin xyz="0 0 0.0584"/><axis xyz="0 1 0"/>
  <limit lower="0" upper="0.04"/></joint>
<joint name="finger_joint2" type="prismatic"><parent link="$EE"/><child link="finger_r"/><origin xyz="0 0 0.0584"/><axis xyz="0 -1 0"/>
  <limit lower="0" upper="0.04"/><mimic joint="finger_joint1"/></joint>
<joint name="head_pan" type="continuous"><parent link="$BASE"/><child link="head"/><origin xyz="0 0 1" rpy="0 0 0"/><axis xyz="0 0 1"/></joint>
<joint name="camera_mount" type="floating"><parent link="head"/><child link="camera"/></joint>
"""


def same_chain(a: Robot, b: Robot):
    assert a.ndof == b.ndof and len(a.joints) == len(b.joints)
    for ja, jb in zip(a.joints, b.joints):
        assert (ja.name, ja.kind) == (jb.name, jb.kind)
        np.testing.assert_allclose(ja.origin_xyz, jb.origin_xyz, atol=0, rtol=0)
        np.testing.assert_allclose(ja.origin_rpy, jb.origin_rpy, atol=1e-15)
        if ja.kind != JOINT_FIXED:
            np.testing.assert_allclose(ja.axis, jb.axis, atol=0, rtol=0)
            np.testing.assert_allclose(ja.limits, jb.limits, atol=1e-15)
    np.testing.assert_allclose(a.actuated_joints_limits, b.actuated_joints_limits, atol=1e-15)
    # what the engine is given: folded pre-transforms and the tool transform
    (ca, ta), (cb, tb) = fold_chain(a), fold_chain(b)
    np.testing.assert_allclose(ta, tb, atol=1e-15)
    for (ka, axa, pa), (kb, axb, pb) in zip(ca, cb):
        assert ka == kb
        np.testing.assert_allclose(axa, axb, atol=0)
        np.testing.assert_allclose(pa, pb, atol=1e-15)


@pytest.mark.parametrize("builtin,xml,skip,base", [
    (Panda, rt.PANDA_URDF, (), "panda_link0"),
    (Fetch, rt.FETCH_URDF, (), "base_link"),
    (FetchArm, rt.FETCH_URDF, ("torso_lift_joint",), "torso_lift_link"),
])
def test_oracle_urdf_text_through_the_product_reader_gives_the_builtin_tables(builtin, xml, skip, base, tmp_path):
    ref = builtin()
    if not skip:  # the bare joint list, exactly as the oracle keeps it: document order is the chain
        same_chain(Robot.from_urdf(xml, name=ref.name), ref)
    text, ee = as_full_urdf(xml, base, skip, BRANCHES)
    same_chain(Robot.from_urdf(text, base, ee, name=ref.name), ref)
    f = tmp_path / "robot.urdf"  # and from a file
    f.write_text(text)
    same_chain(Robot.from_urdf(str(f), base, ee, name=ref.name), ref)
    # a chain that starts further down the tree: the torso joint is above the base and drops out by itself
    if builtin is Fetch:
        sub = Robot.from_urdf(text, "torso_lift_joint_child", ee, name="fetch_arm")
        same_chain(sub, FetchArm())


def test_urdf_defaults_and_continuous_limits():
    text = """<robot name="r"><link name="a"/><link name="b"/><link name="c"/><link name="d"/>
      <joint name="j0" type="continuous"><parent link="a"/><child link="b"/></joint>
      <joint name="f" type="fixed"><parent link="b"/><child link="c"/><origin xyz="1 2 3"/></joint>
      <joint name="j1" type="revolute"><parent link="c"/><child link="d"/><origin rpy="0.1 0.2 0.3"/><axis xyz="0 0 2"/>
        <limit lower="-1" upper="2" effort="10" velocity="1"/></joint></robot>"""
    r = Robot.from_urdf(text, "a", "d")
    assert r.name == "r" and r.ndof == 2 and [j.name for j in r.joints] == ["j0", "f", "j1"]
    j0, f, j1 = r.joints
    assert j0.origin_xyz == (0, 0, 0) and j0.origin_rpy == (0, 0, 0) and j0.axis == (1.0, 0.0, 0.0)  # URDF defaults
    assert j0.limits == (-math.pi, math.pi)  # continuous -> [-pi, pi] (what jrl does; robots.py FetchArm)
    assert f.origin_xyz == (1.0, 2.0, 3.0) and f.limits is None
    assert j1.origin_rpy == (0.1, 0.2, 0.3) and j1.limits == (-1.0, 2.0)
    joints, _ = fold_chain(r)
    np.testing.assert_allclose(joints[1][1], (0, 0, 1))  # the axis is normalised for the engine
    assert Robot.from_urdf(text, "a", "d", continuous_limits=(-6.0, 6.0)).joints[0].limits == (-6.0, 6.0)
    assert Robot.from_urdf(text, "c", "d").ndof == 1


def test_urdf_errors_are_loud():
    text, ee = as_full_urdf(rt.PANDA_URDF, "panda_link0", (), BRANCHES)
    with pytest.raises(AssertionError, match="not a link"):
        Robot.from_urdf(text, "panda_link0", "nope")
    with pytest.raises(AssertionError, match="not a link"):
        Robot.from_urdf(text, "nope", ee)
    with pytest.raises(ValueError, match="not a descendant"):
        Robot.from_urdf(text, ee, "panda_link0")  # upside down
    with pytest.raises(ValueError, match="not a descendant"):
        Robot.from_urdf(text, "head", ee)  # another branch
    with pytest.raises(ValueError, match="mimics"):
        Robot.from_urdf(text, "panda_link0", "finger_r")
    with pytest.raises(ValueError, match="floating"):
        Robot.from_urdf(text, "panda_link0", "camera")
    with pytest.raises(AssertionError, match="required"):
        Robot.from_urdf(text)
    with pytest.raises(ValueError, match="no <limit"):
        Robot.from_urdf('<robot name="r"><joint name="j" type="revolute"><parent link="a"/><child link="b"/></joint></robot>', "a", "b")
    with pytest.raises(AssertionError, match="not a URDF"):
        Robot.from_urdf("<sdf/>")
    with pytest.raises(AssertionError, match="was not found"):
        Robot.from_urdf(os.path.join(os.sep, "no", "such", "file.urdf"))
    assert Robot.from_urdf(text, "panda_link0", "finger_l").ndof == 8  # a finger IS a legal end effector


# A 7-revolute-joint arm with MADE-UP dimensions (NOT the Flexiv Rizon 4: its URDF ships with jrl, which is absent here, and a
# chain typed from memory could not be checked against anything).  It stands in for "the file the user supplies".
SYNTHETIC_7DOF = """
<robot name="synthetic7">
  <joint name="joint1" type="revolute"><origin xyz="0 0 0.155" rpy="0 0 -3.141592653589793"/><axis xyz="0 0 1"/><limit lower="-2.79" upper="2.79"/></joint>
  <joint name="joint2" type="revolute"><origin xyz="0 0.03 0.21" rpy="0 0 0"/><axis xyz="0 1 0"/><limit lower="-2.26" upper="2.26"/></joint>
  <joint name="joint3" type="revolute"><origin xyz="0 0.035 0.205" rpy="0 0 0"/><axis xyz="0 0 1"/><limit lower="-2.96" upper="2.96"/></joint>
  <joint name="joint4" type="revolute"><origin xyz="-0.02 -0.03 0.19" rpy="0 0 -3.141592653589793"/><axis xyz="0 1 0"/><limit lower="-1.86" upper="2.79"/></joint>
  <joint name="joint5" type="revolute"><origin xyz="-0.02 0.025 0.195" rpy="0 0 -3.141592653589793"/><axis xyz="0 0 1"/><limit lower="-2.96" upper="2.96"/></joint>
  <joint name="joint6" type="revolute"><origin xyz="0 0.03 0.19" rpy="0 0 0"/><axis xyz="0 1 0"/><limit lower="-1.48" upper="4.62"/></joint>
  <joint name="joint7" type="revolute"><origin xyz="-0.055 0.07 0.11" rpy="0 -1.5707963267948966 0"/><axis xyz="0 0 1"/><limit lower="-2.96" upper="2.96"/></joint>
  <joint name="link7_to_flange" type="fixed"><origin xyz="0 0 0.081" rpy="0 0 -3.141592653589793"/></joint>
</robot>
"""


def synthetic_rizon_robot():
    text, ee = as_full_urdf(SYNTHETIC_7DOF, "base_link", (), BRANCHES)
    return Robot.from_urdf(text, "base_link", ee, name="rizon4")


def test_rizon4_is_registered_and_takes_a_urdf_robot():
    """The released hyper-parameters (model_descriptions.yaml:90-97) are registered; the robot is data."""
    assert MODEL_DESCRIPTIONS[RIZON_MODEL] == dict(nb_nodes=12, dim_latent_space=7, coeff_fn_config=3, coeff_fn_internal_size=1024,
                                                   rnvp_clamp=2.5, robot_name="rizon4")
    with pytest.raises(ValueError, match="from_urdf"):
        get_robot("rizon4")
    with pytest.raises(ValueError, match="from_urdf"):
        get_ik_solver(RIZON_MODEL, synthetic_weights_seed=0)
    robot = synthetic_rizon_robot()
    solver, hp = get_ik_solver(RIZON_MODEL, robot=robot, synthetic_weights_seed=0)
    assert isinstance(solver, IKFlowSolver) and solver.ndof == 7 and solver.network_width == 7 and hp.nb_nodes == 12
    lay = layout_from(hparams_for(RIZON_MODEL), robot)
    assert lay.flops_per_solution() == 101572608  # the Panda architecture
    with pytest.raises(AssertionError):  # the name must be the model's robot, as in the reference (model_loading.py:83)
        get_ik_solver(RIZON_MODEL, robot=Panda(), synthetic_weights_seed=0)
    # registering the file once makes get_robot / get_ik_solver find it
    from ikflow_amd import robots as pr
    text, ee = as_full_urdf(SYNTHETIC_7DOF, "base_link", (), BRANCHES)
    try:
        register_robot_urdf("rizon4", text, "base_link", ee)
        assert get_robot("rizon4").ndof == 7
        s2, _ = get_ik_solver(RIZON_MODEL, synthetic_weights_seed=0)
        assert s2.robot.name == "rizon4"
    finally:
        pr._ROBOTS.pop("rizon4", None)


@pytest.mark.gpu
def test_urdf_robot_on_the_hip_path_matches_the_oracle():
    """FK, flow and exact IK of a robot that exists only as a URDF: HIP path (product reader, tree form) against the oracle
    (its own reader, bare joint list).  Tolerances as for the built-in robots (FK 2e-6, flow 1e-5)."""
    from oracle import flow_oracle as fo

    robot = synthetic_rizon_robot()
    orob = rt.OracleRobot("rizon4", rt.parse_chain(SYNTHETIC_7DOF))
    dev = torch.device("cuda:0")
    q = torch.tensor(orob.sample_joint_angles(2000, 0.004, np.random.default_rng(3)))
    fk_ref = ko.forward_kinematics(orob, q)
    fk = robot.forward_kinematics(q.to(dev)).cpu()
    assert (fk[:, :3] - fk_ref[:, :3]).abs().max().item() <= 2e-6
    dots = (fk[:, 3:] * fk_ref[:, 3:]).sum(1).abs()
    assert (1 - dots).max().item() <= 2e-6
    # the Rizon4 architecture (= Panda's) on this robot
    hp = hparams_for(RIZON_MODEL)
    lay = fo.OracleLayout(12, 7, 8, 1024, 3, 2.5, 7, False)
    sd = fo.make_state_dict(lay, orob, seed=2)
    s = IKFlowSolver(hp, robot)
    s.load_state_dict_tensors(sd)
    n = 600
    lat = torch.randn(n, 7, generator=torch.Generator().manual_seed(5))
    got = s.generate_ik_solutions(fk_ref[:n].to(dev), latent=lat.to(dev)).cpu()
    ref = fo.generate_ik_solutions_torch(sd, lay, orob, fk_ref[:n], lat)
    assert (got - ref).abs().max().item() <= 1e-5
    # exact IK from perturbed-truth seeds converges on the URDF robot's own kinematics
    eng = s.engine(dev)
    seeds = robot.clamp_to_joint_limits((q[:n] + 0.03 * torch.randn(n, 7, generator=torch.Generator().manual_seed(6))).to(dev))
    sol, valid = eng.refine_exact(fk_ref[:n].to(dev), seeds, 1, 1e-3, 0.01)
    assert valid.float().mean().item() > 0.97
    pe, re = ko.calculate_pose_error(orob, sol.cpu()[valid.cpu()], fk_ref[:n][valid.cpu()])
    assert pe.max().item() < 1e-3 and re.max().item() < 0.01
