"""Shared builders for the tests: seeded models, poses, latents.

Weights, layouts, joint limits, sampled configurations and target poses all come from the ORACLE's own tables and
generators (oracle/flow_oracle.py, oracle/robot_tables.py); the product only contributes the objects its API needs
(ikflow_amd Robot + IkflowModelParameters).  tests/test_oracle_independence.py checks the two sets of tables agree."""
import numpy as np
import torch

from ikflow_amd.model import TINY_MODEL_PARAMS, IkflowModelParameters, hparams_for
from ikflow_amd.robots import get_robot
from oracle import flow_oracle as fo
from oracle import kinematics_oracle as ko
from oracle.robot_tables import OracleRobot
from oracle.robot_tables import robot as oracle_robot_by_name


def O(robot) -> OracleRobot:
    """The oracle's own description of `robot` (looked up by name only)."""
    return robot if isinstance(robot, OracleRobot) else oracle_robot_by_name(robot if isinstance(robot, str) else robot.name)


def released_model(model_name, seed=0, gain=1.0):
    """(product Robot, product hparams, oracle layout, oracle-generated state_dict) of a released architecture."""
    lay = fo.layout_for(model_name)
    robot_name = fo.RELEASED[model_name][0]
    return get_robot(robot_name), hparams_for(model_name), lay, fo.make_state_dict(lay, robot_name, seed=seed, output_gain=gain)


def panda_model(seed=0, gain=1.0):
    return released_model("panda__full__lp191_5.25m", seed, gain)


def fetch_arm_model(seed=0, gain=1.0):
    return released_model("fetch_arm__large__mh186_9.25m", seed, gain)


def tiny_model(seed=0, gain=1.0):
    lay = fo.layout_for("tiny")
    return get_robot("panda"), TINY_MODEL_PARAMS, lay, fo.make_state_dict(lay, "panda", seed=seed, output_gain=gain)


def custom_model(nb_nodes=2, dim=9, n_hidden=2, width=256, robot_name="panda", softflow=True, sigmoid=False, seed=0, gain=1.0):
    """Any (coeff_fn_config, coeff_fn_internal_size, ...) the reference's IkflowModelParameters can express."""
    hp = IkflowModelParameters()
    hp.nb_nodes, hp.dim_latent_space, hp.coeff_fn_config, hp.coeff_fn_internal_size = nb_nodes, dim, n_hidden, width
    hp.softflow_enabled, hp.sigmoid_on_output = softflow, sigmoid
    lay = fo.OracleLayout(nb_nodes, dim, 8 if softflow else 7, width, n_hidden, float(hp.rnvp_clamp), O(robot_name).ndof, sigmoid)
    return get_robot(robot_name), hp, lay, fo.make_state_dict(lay, robot_name, seed=seed, output_gain=gain)


def reachable_poses(robot, n, seed=0, eps=0.004363323129985824):
    """poses = FK(q), q ~ U(lo+eps, hi-eps) (scripts/build_dataset.py:186 convention)."""
    q = torch.tensor(O(robot).sample_joint_angles(n, eps, np.random.default_rng(seed)))
    return q, ko.forward_kinematics(robot, q)


def latents(n, dim, seed=1):
    return torch.randn(n, dim, generator=torch.Generator().manual_seed(seed))
