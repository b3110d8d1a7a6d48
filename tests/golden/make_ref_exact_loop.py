"""Generates tests/golden/ref_exact_loop.npz: the REFERENCE's own exact-IK control flow, executed in the build container.

``IKFlowSolver._generate_exact_ik_solutions`` and ``generate_exact_ik_solutions`` (/root/reference/ikflow/ikflow_solver.py:119-247,
345-411) are taken out of the reference file with ``ast`` and run as plain functions on a namespace object in place of ``self``.
What they call outside themselves is bound as follows - stated exactly, because it bounds what the fixture pins:

  self._run_inference(latent, cond, t0, True, False)            -> oracle/flow_oracle.py (TINY model, seeded weights)
  self.robot.inverse_kinematics_step_levenburg_marquardt(p, q)  -> oracle/kinematics_oracle.py lm_step        (jrl is absent)
  self._calculate_pose_error(q, p)                              -> oracle/kinematics_oracle.py
  draw_latent                                                   -> the reference's own function (ikflow_solver.py:16-29), wrapped
                                                                   only to record what it returns
  DEFAULT_TORCH_DTYPE = torch.float32 (ikflow/config.py:8); mm_to_m / make_text_green_or_red (jrl.utils: a unit conversion used in a
  default argument that is overridden here, and a print colouring helper) -> one-line lambdas

The same is done for ``generate_ik_solutions`` + ``_run_inference`` (:254-343, :85-110; ``self.nn_model(latent, c=cond, rev=True)``
bound to the oracle's ``flow_inverse_torch``, ``self.robot.clamp_to_joint_limits`` to the oracle's clamp, DEVICE = "cpu"): the fixture
then pins rows A1 / A2 - the conditional assembly ``cat([y, 0])`` / ``y.expand((n, 7))``, the ``[:, :ndof]`` slice, the clamp call, the
latent drawn with the reference's ``draw_latent`` when none is passed - and the argument asserts (quirks Q1, Q2).

So the fixture pins the CONTROL FLOW of rows B5 / B6 - validity mask, ``idx % n_invalid`` "highest valid repeat wins", slot order,
boolean-mask compaction, retry rounds, the ``new_solutions.all()`` early return - against the reference's statements themselves; the
kinematics and the flow underneath are the oracle's on both sides and are NOT pinned by it.  tests/test_oracle_golden.py replays the
recorded latents through the oracle's restatement of the loop and requires identical outputs.

Run from the repo root (only where /root/reference exists):  python tests/golden/make_ref_exact_loop.py
"""
import ast
import functools
import os
import sys
import types
from time import time
from typing import Callable, Dict, Optional, Tuple, Union

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import flow_oracle as fo  # noqa: E402
from oracle import kinematics_oracle as ko  # noqa: E402

REF = "/root/reference/ikflow/ikflow_solver.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_loop(recorded):
    tree = ast.parse(open(REF).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "draw_latent"]
    for cls in tree.body:
        if isinstance(cls, ast.ClassDef) and cls.name == "IKFlowSolver":
            keep += [f for f in cls.body if isinstance(f, ast.FunctionDef) and f.name in (
                "_generate_exact_ik_solutions", "generate_exact_ik_solutions", "generate_ik_solutions", "_run_inference")]
    ns = {"torch": torch, "time": time, "Tuple": Tuple, "Optional": Optional, "Union": Union, "Callable": Callable, "Dict": Dict,
          "DEFAULT_TORCH_DTYPE": torch.float32, "DEVICE": "cpu", "mm_to_m": lambda x: x / 1000.0,
          "make_text_green_or_red": lambda s, green: s,
          # annotation alias of ikflow/evaluation_utils.py:19 (only named in a return annotation)
          "SOLUTION_EVALUATION_RESULT_TYPE": Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, float]}
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF, "exec"), ns)
    ref_draw = ns["draw_latent"]

    def recording_draw_latent(*a, **k):
        z = ref_draw(*a, **k)
        recorded.append(z.clone())
        return z

    ns["draw_latent"] = recording_draw_latent
    return ns


def run_case(n, seed, pos_thr, rot_thr, rc=(1, 3, 10)):
    lay = fo.layout_for("tiny")
    sd = fo.make_state_dict(lay, "panda", seed=2)
    robot = ko._R("panda")
    q = torch.tensor(robot.sample_joint_angles(n, 0.004363323129985824, np.random.default_rng(seed)))
    poses = ko.forward_kinematics("panda", q)
    recorded = []
    ns = load_reference_loop(recorded)
    me = types.SimpleNamespace(
        ndof=robot.ndof, _network_width=lay.dim, _model_weights_loaded=True,
        robot=types.SimpleNamespace(inverse_kinematics_step_levenburg_marquardt=lambda p, qq: ko.lm_step("panda", p, qq)),
        _run_inference=lambda latent, cond, t0, clamp, detailed: fo.run_inference_torch(sd, lay, "panda", latent, cond.contiguous(), clamp),
        _calculate_pose_error=lambda qq, p: ko.calculate_pose_error("panda", qq, p),
    )
    me._generate_exact_ik_solutions = functools.partial(ns["_generate_exact_ik_solutions"], me)
    torch.manual_seed(1000 + seed)
    sol, valid = ns["generate_exact_ik_solutions"](me, poses, repeat_counts=rc, pos_error_threshold=pos_thr,
                                                   rot_error_threshold=rot_thr, run_lma_on_cpu=False)
    return poses, recorded, sol, valid


def run_approx_cases(out):
    """generate_ik_solutions / _run_inference of the reference over the oracle's flow: batch form, single-pose form, drawn latent,
    unclamped; and which argument combinations its asserts reject."""
    lay = fo.layout_for("tiny")
    sd = fo.make_state_dict(lay, "panda", seed=3, output_gain=1.5)
    robot = ko._R("panda")
    recorded = []
    ns = load_reference_loop(recorded)
    me = types.SimpleNamespace(
        ndof=robot.ndof, _network_width=lay.dim, _model_weights_loaded=True,
        nn_model=lambda latent, c, rev: (fo.flow_inverse_torch(sd, lay, latent, c.contiguous()), None),
        robot=types.SimpleNamespace(clamp_to_joint_limits=lambda q: ko.clamp_to_joint_limits("panda", q)),
    )
    me._run_inference = functools.partial(ns["_run_inference"], me)
    gen = functools.partial(ns["generate_ik_solutions"], me)
    n = 24
    q = torch.tensor(robot.sample_joint_angles(n, 0.004363323129985824, np.random.default_rng(77)))
    poses = ko.forward_kinematics("panda", q)
    lat = 2.0 * torch.randn(n, lay.dim, generator=torch.Generator().manual_seed(78))  # wide: some joints hit the clamp
    out["ik_poses"], out["ik_latent"] = poses.numpy().copy(), lat.numpy().copy()
    out["ik_batch_clamped"] = gen(poses, latent=lat).numpy().copy()
    out["ik_batch_unclamped"] = gen(poses, latent=lat, clamp_to_joint_limits=False).numpy().copy()
    out["ik_single_pose"] = gen(poses[3], n=n, latent=lat).numpy().copy()
    out["ik_single_pose_1x7"] = gen(poses[3:4], n=n, latent=lat).numpy().copy()  # quirk Q2: [1 x 7] is the single-pose form
    torch.manual_seed(4321)
    out["ik_drawn_latent"] = gen(poses[5], n=6, latent_scale=0.5).numpy().copy()  # latent drawn by the reference's draw_latent
    out["ik_drawn_latent_value"] = recorded[-1].numpy().copy()
    rejected = []
    for name, kw in {"scale_int": dict(y=poses, latent_scale=1), "single_needs_n": dict(y=poses[0]), "n_zero": dict(y=poses[0], n=0),
                     "y_list": dict(y=[0.0] * 7, n=2), "y_6_columns": dict(y=poses[:, :6]), "latent_numpy": dict(y=poses, latent=lat.numpy()),
                     "refine": dict(y=poses, refine_solutions=True), "ok_batch": dict(y=poses)}.items():
        try:
            gen(**kw)
        except AssertionError:
            rejected.append(name)
    out["ik_asserted"] = np.array(",".join(rejected))
    print("approx cases: asserted on", rejected)


def main():
    torch.set_num_threads(1)
    out = {}
    run_approx_cases(out)
    for tag, (n, seed, pos_thr, rot_thr) in {"a": (40, 21, 0.2, 1.0), "b": (7, 5, 0.3, 1.5), "c": (64, 9, 0.05, 0.3)}.items():
        poses, lats, sol, valid = run_case(n, seed, pos_thr, rot_thr)
        out[f"{tag}_poses"] = poses.numpy().copy()
        out[f"{tag}_thresholds"] = np.array([pos_thr, rot_thr])
        out[f"{tag}_n_rounds"] = np.int64(len(lats))
        for i, z in enumerate(lats):
            out[f"{tag}_latent_{i}"] = z.numpy().copy()
        out[f"{tag}_solutions"] = sol.numpy().copy()
        out[f"{tag}_valid"] = valid.numpy().copy()
        print(tag, "n", n, "valid", int(valid.sum()), "rounds run", len(lats), "latent rows", [z.shape[0] for z in lats])
    path = os.path.join(HERE, "ref_exact_loop.npz")
    np.savez(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
