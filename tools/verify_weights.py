#!/usr/bin/env python
"""verify_weights.py <file.pkl|file.npz> [--model NAME] [--poses N] [--solutions K]

First contact with a RELEASED weight file (none has ever been available offline; the URLs of
/root/reference/ikflow/model_descriptions.yaml are remote).  The file format is the reference's: ``pickle.dump`` of
``{FrEIA GraphINN key: torch.Tensor}`` (ikflow/ikflow_solver.py:413-429), optionally prefixed ``nn_model.``
(scripts/download_model_from_wandb_checkpoint.py:13-28).  What is checked:

  1. keys / shapes    every tensor the architecture needs is there with the right shape (the architecture is taken from
                      --model, or inferred from the shapes); unknown extra keys are listed
  2. permutations     perm / perm_inv are inverse of each other and equal numpy-MT19937 ``permutation(D)`` under seed i -
                      the day a real file passes this, row A5 of SURVEY 8 (FrEIA PermuteRandom(seed=i)) is pinned by the
                      reference's own artefact instead of by recall
  3. linear transform M . M_inv = I, and M_inv's diagonal equals max(|lo|, |hi|) of the robot's joint limits (plain graph)
                      or the scaling node's slope (sigmoid graph)
  4. finite, sane     no NaN / inf; hidden weights inside the f16 range (else the f16x3 mode would be refused)
  5. pose error       (needs the MI355X) K solutions for each of N reachable target poses through the engine: mean position
                      and rotation error.  The README advertises millimetres / sub-degree for the released models;
                      random or mis-mapped weights give tens of centimetres.

Exit status 0 = everything checked passed, 1 = a structural check failed, 2 = structure fine but the pose error is not
that of a trained model.
"""
from __future__ import annotations

import argparse
import json
import os
import pickle
import sys
from typing import Dict, Optional, Tuple

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np


def load_any(path: str) -> Dict[str, np.ndarray]:
    """The reference's pickle (torch tensors) or an .npz with the same keys -> {key: numpy array}, prefixes stripped the
    way IKFlowSolver.load_state_dict_tensors strips them."""
    from ikflow_amd.model import state_dict_to_numpy

    if path.endswith(".npz"):
        with np.load(path) as z:
            sd = {k: z[k] for k in z.files}
    else:
        with open(path, "rb") as f:
            sd = pickle.load(f)
    sd = state_dict_to_numpy(sd)
    out = {}
    for k, v in sd.items():
        for pre in ("nn_model.", "_orig_mod."):
            if k.startswith(pre):
                k = k[len(pre):]
        out[k.replace(".s1.", ".subnet1.").replace(".s2.", ".subnet2.")] = v
    return out


def infer_architecture(sd: Dict[str, np.ndarray]) -> Dict:
    """nb_nodes, D, width, n_hidden, dim_cond, sigmoid graph or not - from key names and shapes alone."""
    import re

    D = int(sd["module_list.0.M_inv"].shape[0])
    glow = sorted({int(m.group(1)) for k in sd for m in [re.match(r"module_list\.(\d+)\.subnet1\.0\.weight", k)] if m})
    assert glow, "no module_list.<i>.subnet1.0.weight key: not a GLOW coupling flow"
    first = glow[0]
    sigmoid = first == 3  # plain graph: block 0's coupling is module 2; sigmoid graph: module 3
    w0 = sd[f"module_list.{first}.subnet1.0.weight"]
    width, cin = int(w0.shape[0]), int(w0.shape[1])
    dim_cond = cin - D // 2
    lin = sorted({int(m.group(1)) for k in sd for m in [re.match(rf"module_list\.{first}\.subnet1\.(\d+)\.weight", k)] if m})
    return dict(nb_nodes=len(glow), dim=D, width=width, n_hidden=len(lin) - 1, dim_cond=dim_cond, sigmoid_on_output=sigmoid)


def check_structure(sd: Dict[str, np.ndarray], model_name: Optional[str] = None, robot_name: Optional[str] = None) -> Tuple[bool, Dict]:
    """Checks 1-4.  Returns (ok, report)."""
    from ikflow_amd.model import (MODEL_DESCRIPTIONS, FlowLayout, freia_permutation, hparams_for, key_perm, key_perm_inv,
                                  layout_from, validate_state_dict)
    from ikflow_amd.robots import get_robot

    rep: Dict = {"n_tensors": len(sd), "n_parameters": int(sum(v.size for v in sd.values())),
                 "bytes_fp32": int(sum(v.size * 4 for v in sd.values() if v.dtype.kind == "f"))}
    ok = True
    arch = infer_architecture(sd)
    rep["inferred"] = arch
    if model_name is not None:
        robot = get_robot(MODEL_DESCRIPTIONS[model_name]["robot_name"])
        layout = layout_from(hparams_for(model_name), robot)
        want = dict(nb_nodes=layout.nb_nodes, dim=layout.dim, width=layout.width, n_hidden=layout.n_hidden, dim_cond=layout.dim_cond,
                    sigmoid_on_output=layout.sigmoid_on_output)
        rep["architecture_matches_model"] = want == arch
        ok = ok and want == arch
    else:
        robot = get_robot(robot_name) if robot_name else None
        layout = FlowLayout(nb_nodes=arch["nb_nodes"], dim=arch["dim"], dim_cond=arch["dim_cond"], width=arch["width"],
                            n_hidden=arch["n_hidden"], clamp=2.5, ndof=(robot.ndof if robot else min(arch["dim"], 7)),
                            sigmoid_on_output=arch["sigmoid_on_output"])
    # 1. keys / shapes
    try:
        validate_state_dict(layout, sd)
        rep["keys_and_shapes"] = "ok"
    except RuntimeError as e:
        rep["keys_and_shapes"] = str(e)
        return False, rep
    off = layout.module_offset
    known = {"module_list.0.M", "module_list.0.M_inv", "module_list.0.b", "module_list.0.logDetM"}
    for i in range(layout.nb_nodes):
        known |= {key_perm(i, off), key_perm_inv(i, off)}
        known |= {k for k in sd if k.startswith(f"module_list.{2 * i + 2 + off}.subnet")}
    rep["unknown_keys"] = sorted(set(sd) - known)[:20]
    # 2. permutations
    perm_report = []
    for i in range(layout.nb_nodes):
        pinv = np.asarray(sd[key_perm_inv(i, off)]).astype(np.int64)
        want = freia_permutation(layout.dim, i)
        want_inv = np.zeros_like(want)
        want_inv[want] = np.arange(layout.dim)
        entry = {"block": i, "perm_inv_is_permutation": sorted(pinv.tolist()) == list(range(layout.dim)),
                 "equals_numpy_seed_i": bool(np.array_equal(pinv, want_inv))}
        if key_perm(i, off) in sd:
            perm = np.asarray(sd[key_perm(i, off)]).astype(np.int64)
            entry["perm_perm_inv_consistent"] = bool(np.array_equal(pinv[perm], np.arange(layout.dim)))
            ok = ok and entry["perm_perm_inv_consistent"]
        ok = ok and entry["perm_inv_is_permutation"]
        perm_report.append(entry)
    rep["permutations_equal_numpy_seed_i"] = all(e["equals_numpy_seed_i"] for e in perm_report)
    rep["permutation_blocks_differing"] = [e["block"] for e in perm_report if not e["equals_numpy_seed_i"]]
    # a file that carries other permutations is still loadable (the file's tables take precedence) - reported, not failed
    # 3. linear transform
    M_inv = np.asarray(sd["module_list.0.M_inv"], dtype=np.float64)
    if "module_list.0.M" in sd:
        err = float(np.abs(np.asarray(sd["module_list.0.M"], dtype=np.float64) @ M_inv - np.eye(layout.dim)).max())
        rep["M_times_M_inv_minus_I"] = err
        ok = ok and err < 1e-4
    rep["M_inv_is_diagonal"] = bool(np.count_nonzero(M_inv - np.diag(np.diag(M_inv))) == 0)
    if robot is not None:
        lim = robot.actuated_joints_limits
        if layout.sigmoid_on_output:
            want_diag = [hi - lo for lo, hi in lim]
        else:
            want_diag = [max(abs(lo), abs(hi)) for lo, hi in lim]
        got = np.diag(M_inv)[: robot.ndof]
        rep["M_inv_diag_matches_joint_limits"] = bool(np.allclose(got, want_diag, rtol=1e-4))
        rep["M_inv_diag"] = [round(float(v), 5) for v in np.diag(M_inv)]
        ok = ok and rep["M_inv_diag_matches_joint_limits"]
    # 4. finite / range
    bad = [k for k, v in sd.items() if v.dtype.kind == "f" and not np.isfinite(v).all()]
    rep["non_finite_tensors"] = bad[:10]
    ok = ok and not bad
    hidden_max = max((float(np.abs(v).max()) for k, v in sd.items() if k.endswith(".weight") and v.ndim == 2 and v.shape[0] == v.shape[1] == layout.width), default=0.0)
    rep["max_abs_hidden_weight"] = hidden_max
    rep["f16x3_mode_usable"] = hidden_max <= 65504.0
    rep["layout"] = dict(nb_nodes=layout.nb_nodes, dim=layout.dim, dim_cond=layout.dim_cond, width=layout.width, n_hidden=layout.n_hidden,
                         sigmoid_on_output=layout.sigmoid_on_output, weight_bytes=layout.weight_bytes(), flops_per_solution=layout.flops_per_solution())
    return ok, rep


def pose_error_report(sd, model_name: str, n_poses: int, k_solutions: int) -> Dict:
    """Check 5 on the GPU: mean L2 / geodesic error of K approximate solutions for each of N reachable poses, and the
    exact-IK success rate at the README thresholds (1 mm / 0.01 rad)."""
    import torch

    from ikflow_amd.ikflow_solver import IKFlowSolver
    from ikflow_amd.model import MODEL_DESCRIPTIONS, hparams_for
    from ikflow_amd.robots import get_robot

    robot = get_robot(MODEL_DESCRIPTIONS[model_name]["robot_name"])
    s = IKFlowSolver(hparams_for(model_name), robot)
    s.load_state_dict_tensors(sd)
    dev = "cuda:0"
    q = torch.tensor(robot.sample_joint_angles(n_poses, 0.004363323129985824, np.random.default_rng(0)), device=dev)
    poses = robot.forward_kinematics(q)
    tiled = poses.repeat((k_solutions, 1))
    sol, pe, re, lim, _, _ = s.generate_ik_solutions(tiled, return_detailed=True)
    ex_sol, ex_valid = s.generate_exact_ik_solutions(poses, pos_error_threshold=1e-3, rot_error_threshold=0.01)
    return {"poses": n_poses, "solutions_per_pose": k_solutions, "mean_pos_error_mm": float(pe.mean() * 1e3),
            "mean_rot_error_deg": float(torch.rad2deg(re).mean()), "median_pos_error_mm": float(pe.median() * 1e3),
            "joint_limits_exceeded_fraction": float(lim.float().mean()),
            "exact_ik_valid_fraction_1mm_0.01rad": float(ex_valid.float().mean())}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("path")
    ap.add_argument("--model", default=None, help="a name of ikflow_amd.model.MODEL_DESCRIPTIONS (else inferred from the shapes)")
    ap.add_argument("--robot", default=None, help="robot name for the limit check when --model is not given")
    ap.add_argument("--poses", type=int, default=200)
    ap.add_argument("--solutions", type=int, default=50)
    a = ap.parse_args()
    sd = load_any(a.path)
    ok, rep = check_structure(sd, a.model, a.robot)
    status = 0 if ok else 1
    if ok and a.model is not None:
        import torch

        if torch.cuda.is_available():
            rep["pose_error"] = pe = pose_error_report(sd, a.model, a.poses, a.solutions)
            trained = pe["mean_pos_error_mm"] < 20.0 and pe["mean_rot_error_deg"] < 5.0
            rep["looks_trained"] = trained
            status = 0 if trained else 2
        else:
            rep["pose_error"] = "skipped: no GPU visible (the engine has no CPU path)"
    print(json.dumps(rep, indent=1))
    sys.exit(status)


if __name__ == "__main__":
    main()
