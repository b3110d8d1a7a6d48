"""Model registry / loader - mirror of ikflow/model_loading.py:60-90 for the released-model hyper-parameters.

The reference downloads the weight pickle from a GCS URL into ~/.cache/ikflow/models/ (model_loading.py:31-50); there is
no network path here, so `get_ik_solver` looks for the already-downloaded file in that same cache directory (or an
explicit `weights_path`), and can otherwise build the solver with seeded synthetic weights of the right shape."""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

from ikflow_amd.config import MODELS_DIR
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import MODEL_DESCRIPTIONS, IkflowModelParameters, layout_from, random_state_dict
from ikflow_amd.robots import Robot, get_robot

# file names of the released weights (last URL component, model_loading.py:53-58)
MODEL_WEIGHT_FILENAMES = {
    "panda__full__lp191_5.25m": "panda__lyric-puddle-191__global_step%3D5.25M.pkl",
    "panda_lite_tpm": "panda_arm-young-night-84.pkl",
    "fetch_full_temp_nsc_tpm": "fetch__sleek-microwave-65__global_step%3D9.25M.pkl",
    "fetch__large__ns183_9.75m": "fetch__northern-sea-183__global_step%3D9.75M.pkl",
    "fetch_arm__large__mh186_9.25m": "fetch_arm__major-hill-186__global_step%3D9.25M.pkl",
    "rizon4__snowy-brook-208__global_step=2.75M": "rizon4__snowy-brook-208__global_step%3D2.75M.pkl",
}


def get_all_model_names() -> Tuple[str]:
    return tuple(MODEL_DESCRIPTIONS.keys())


def model_filename(url: str) -> str:
    """https://storage.googleapis.com/ikflow_models/atlas_desert-sweep-6.pkl -> atlas_desert-sweep-6.pkl"""
    return url.split("/")[-1]


def _assert_model_downloaded_correctly(filepath: str):
    filesize_mb = os.path.getsize(filepath) * 0.000001
    assert filesize_mb > 10, f"Model weights saved at '{filepath}' has only {filesize_mb} MB - was it saved correctly?"


def get_ik_solver(
    model_name: str,
    robot: Optional[Robot] = None,
    compile_model: Optional[Dict] = None,
    weights_path: Optional[str] = None,
    synthetic_weights_seed: Optional[int] = None,
) -> Tuple[IKFlowSolver, IkflowModelParameters]:
    """Build the `IKFlowSolver` for a model of MODEL_DESCRIPTIONS and set its weights.

    weights: `weights_path` if given, else ~/.cache/ikflow/models/<released file name> if present, else - only when
    `synthetic_weights_seed` is given - seeded random weights with the reference's initialisation."""
    assert model_name in MODEL_DESCRIPTIONS, f"Model name '{model_name}' not found in model descriptions"
    hparams = MODEL_DESCRIPTIONS[model_name]
    robot_name = hparams["robot_name"]
    if robot is None:
        robot = get_robot(robot_name)
    assert robot.name == robot_name
    hyper_parameters = IkflowModelParameters()
    hyper_parameters.__dict__.update(hparams)
    ik_solver = IKFlowSolver(hyper_parameters, robot, compile_model=compile_model)

    path = weights_path
    if path is None:
        cand = os.path.join(MODELS_DIR, MODEL_WEIGHT_FILENAMES.get(model_name, ""))
        if os.path.isfile(cand):
            path = cand
    if path is not None:
        assert os.path.isfile(path), f"File '{path}' was not found. Unable to load model weights"
        _assert_model_downloaded_correctly(path)
        ik_solver.load_state_dict(path)
    elif synthetic_weights_seed is not None:
        ik_solver.load_state_dict_tensors(random_state_dict(layout_from(hyper_parameters, robot), robot, synthetic_weights_seed))
    else:
        raise FileNotFoundError(
            f"No weight file for '{model_name}': expected {os.path.join(MODELS_DIR, MODEL_WEIGHT_FILENAMES.get(model_name, '?'))} "
            "(the reference downloads it there; this build has no network path). Pass weights_path=..., or "
            "synthetic_weights_seed=<int> for seeded random weights."
        )
    return ik_solver, hyper_parameters
