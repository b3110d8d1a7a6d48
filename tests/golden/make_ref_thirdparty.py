"""ARMED BUT DORMANT: the pin for the third-party arithmetic of the hot path (VERDICT r04 "missing #3").

What the oracle restates from memory of absent libraries - FrEIA 0.2's GraphINN / GLOWCouplingBlock / PermuteRandom / FixedLinearTransform and
jrl@2ba7c39's forward kinematics, Levenberg-Marquardt step, geodesic distance and joint-limit clamp - cannot be pinned in the build container:
neither library is installed and there is no network.  This script produces the vectors the day they ARE importable:

    pip install FrEIA==0.2                                                     # uv.lock:533-541
    pip install "jrl @ git+https://github.com/jstmn/jrl.git@2ba7c3995b36b32886a8aa021a00c73b2cd55b2c"   # uv.lock:875-877
    python tests/golden/make_ref_thirdparty.py [--ikflow /path/to/ikflow/checkout]      ->  tests/golden/ref_thirdparty.npz

and `tests/test_thirdparty_pin.py` compares the oracle with them - live (the libraries importable: vectors are regenerated in memory)
or from the committed fixture (anywhere).  Until then both skip cleanly.  Nothing of the reference or of the libraries is copied: the fixture
holds inputs, outputs, seeds and key names.

What it records, each by calling the library / the reference itself:
  flow      the reference's own graph builder `glow_cNF_model` (ikflow/model.py:291-356; imported from an ikflow checkout when one is given
            or installed, else rebuilt from the same FrEIA calls: InputNode, ConditionNode, FixedLinearTransform, PermuteRandom(seed=i),
            GLOWCouplingBlock(clamp, split_len=D // 2), GraphINN) on SMALL widths (the arithmetic does not depend on the width) with
            torch.manual_seed weights, the last Linear of every subnet scaled so that the coupling coefficients are O(1);
            state_dict, `nn_model(latent, c=cond, rev=True)` (ikflow_solver.py:98), and the forward pass of its output (a round trip);
  perm      PermuteRandom(seed=i).perm / perm_inv for i = 0..15, D = 7, 8, 9, 10 (model.py:339);
  fk        robot.forward_kinematics(q) for Panda, Fetch, FetchArm (ikflow_solver.py:114), incl. q = 0 and the limits' corners;
  lm        robot.inverse_kinematics_step_levenburg_marquardt(target_poses, q) with its defaults (ikflow_solver.py:205,208);
  geodesic  jrl.math_utils.geodesic_distance_between_quaternions(q1, q2) (ikflow_solver.py:116) on random, equal, opposite and orthogonal pairs;
  limits    robot.actuated_joints_limits and robot.clamp_to_joint_limits (ikflow_solver.py:101-102).
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "ref_thirdparty.npz")
ROBOTS = ("panda", "fetch", "fetch_arm")
FLOW_CASES = (  # (name, robot, nb_nodes, D, coeff_fn_config, width)
    ("panda_d7", "panda", 3, 7, 3, 48),
    ("fetch_d8", "fetch", 2, 8, 3, 32),
    ("fetch_arm_d10", "fetch_arm", 2, 10, 2, 32),
    ("panda_d9_cfg4", "panda", 2, 9, 4, 24),
)
N_ROWS = 24


def libraries_available() -> bool:
    try:
        import FrEIA.framework  # noqa: F401
        import FrEIA.modules  # noqa: F401
        import jrl.robots  # noqa: F401
        return True
    except Exception:
        return False


def _jrl_robot(name: str):
    import jrl.robots as R

    for cand in ({"panda": ("Panda",), "fetch": ("Fetch",), "fetch_arm": ("FetchArm",)}[name]):
        if hasattr(R, cand):
            return getattr(R, cand)()
    return R.get_robot(name)


def _reference_model_builder(ikflow_path):
    """The reference's own `glow_cNF_model` + `IkflowModelParameters` when an ikflow checkout is importable, else None."""
    for p in filter(None, (ikflow_path, "/root/reference")):
        if os.path.isdir(os.path.join(p, "ikflow")) and p not in sys.path:
            sys.path.insert(0, p)
    try:
        from ikflow.model import IkflowModelParameters, glow_cNF_model

        return glow_cNF_model, IkflowModelParameters
    except Exception:
        return None


def _build_graph(robot, nb_nodes, dim, cfg, width, builder):
    """GraphINN of the reference's graph (sigmoid_on_output = False, softflow conditional of 8 entries)."""
    import FrEIA.framework as Ff
    import FrEIA.modules as Fm
    import torch.nn as nn

    if builder is not None:
        glow_cNF_model, Params = builder
        hp = Params()
        hp.nb_nodes, hp.dim_latent_space, hp.coeff_fn_config, hp.coeff_fn_internal_size = nb_nodes, dim, cfg, width
        hp.sigmoid_on_output = False
        return glow_cNF_model(hp, robot, 8, dim).cpu(), "ikflow.model.glow_cNF_model"

    def subnet(ch_in, ch_out):  # the Sequential of ikflow/model.py:51-96
        layers = [nn.Linear(ch_in, width), nn.LeakyReLU()]
        for _ in range(cfg - 1):
            layers += [nn.Linear(width, width), nn.LeakyReLU()]
        return nn.Sequential(*layers, nn.Linear(width, ch_out))

    nodes = [Ff.InputNode(dim, name="input")]
    cond = Ff.ConditionNode(8)
    M, b = torch.eye(dim), torch.zeros(dim)
    for i in range(robot.ndof):
        lo, hi = robot.actuated_joints_limits[i]
        M[i, i] = 1.0 / max(abs(lo), abs(hi))
    nodes.append(Ff.Node([nodes[-1].out0], Fm.FixedLinearTransform, {"M": M, "b": b}))
    for i in range(nb_nodes):
        nodes.append(Ff.Node([nodes[-1].out0], Fm.PermuteRandom, {"seed": i}))
        nodes.append(Ff.Node(nodes[-1].out0, Fm.GLOWCouplingBlock, {"subnet_constructor": subnet, "clamp": 2.5, "split_len": dim // 2}, conditions=cond))
    return Ff.GraphINN(nodes + [cond, Ff.OutputNode([nodes[-1].out0], name="output")], verbose=False), "FrEIA calls of ikflow/model.py:291-356 restated"


def collect(ikflow_path=None) -> dict:
    """Every vector as {name: numpy array}.  Needs FrEIA and jrl."""
    import FrEIA
    import FrEIA.modules as Fm
    from jrl.math_utils import geodesic_distance_between_quaternions

    torch.set_num_threads(1)
    out = {"meta": np.array(json.dumps({"FrEIA": getattr(FrEIA, "__version__", "?"), "torch": torch.__version__,
                                        "flow_cases": [list(c) for c in FLOW_CASES], "robots": list(ROBOTS)}))}
    builder = _reference_model_builder(ikflow_path)
    robots = {name: _jrl_robot(name) for name in ROBOTS}
    # ---- flow
    for ci, (case, rname, nb, dim, cfg, width) in enumerate(FLOW_CASES):
        torch.manual_seed(700 + ci)
        model, how = _build_graph(robots[rname], nb, dim, cfg, width, builder)
        model.eval()
        sd = model.state_dict()
        with torch.no_grad():
            for k, v in sd.items():  # coupling coefficients of O(1): a fresh nn.Linear stack outputs ~1e-1
                if ".subnet" in k and k.endswith(f"{2 * cfg}.weight"):
                    v.mul_(3.0)
        g = torch.Generator().manual_seed(dim * 100 + nb)
        latent = torch.randn(N_ROWS, dim, generator=g)
        pose = torch.randn(N_ROWS, 7, generator=g)
        pose[:, 3:] /= pose[:, 3:].norm(dim=1, keepdim=True)
        cond = torch.cat([pose, torch.zeros(N_ROWS, 1)], dim=1)
        cond[N_ROWS // 2:, 7] = 0.25  # (a non-zero softflow entry on half the rows)
        with torch.inference_mode():
            rev, _ = model(latent, c=cond, rev=True)
            fwd, _ = model(rev, c=cond, rev=False)
        out[f"flow_{case}_how"] = np.array(how)
        out[f"flow_{case}_keys"] = np.array(json.dumps(list(sd.keys())))
        for k, v in sd.items():
            out[f"flow_{case}_sd_{k}"] = v.detach().cpu().numpy().copy()
        out[f"flow_{case}_latent"], out[f"flow_{case}_cond"] = latent.numpy().copy(), cond.numpy().copy()
        out[f"flow_{case}_rev"], out[f"flow_{case}_fwd_of_rev"] = rev.numpy().copy(), fwd.numpy().copy()
    # ---- permutations
    for dim in (7, 8, 9, 10):
        perm, perm_inv = [], []
        for seed in range(16):
            mod = Fm.PermuteRandom([(dim,)], seed=seed)
            perm.append(mod.perm.detach().cpu().numpy().astype(np.int64))
            perm_inv.append(mod.perm_inv.detach().cpu().numpy().astype(np.int64))
        out[f"perm_d{dim}"], out[f"perm_inv_d{dim}"] = np.stack(perm), np.stack(perm_inv)
    # ---- kinematics
    for rname, robot in robots.items():
        lim = np.array(robot.actuated_joints_limits, dtype=np.float64)
        rng = np.random.default_rng(len(rname))
        q = rng.uniform(lim[:, 0], lim[:, 1], size=(48, robot.ndof)).astype(np.float32)
        q[0] = 0.0
        q[1], q[2] = lim[:, 0].astype(np.float32), lim[:, 1].astype(np.float32)
        qt = torch.tensor(q)
        fk = robot.forward_kinematics(qt)
        seeds = torch.tensor((q + rng.normal(0, 0.05, size=q.shape)).astype(np.float32))
        out[f"kin_{rname}_limits"] = lim
        out[f"kin_{rname}_q"], out[f"kin_{rname}_fk"] = q, fk.detach().cpu().numpy().copy()
        out[f"kin_{rname}_lm_seed"] = seeds.numpy().copy()
        out[f"kin_{rname}_lm_step"] = robot.inverse_kinematics_step_levenburg_marquardt(fk, seeds.clone()).detach().cpu().numpy().copy()
        wild = torch.tensor((q * 1.7).astype(np.float32))
        out[f"kin_{rname}_clamp_in"] = wild.numpy().copy()
        out[f"kin_{rname}_clamp_out"] = robot.clamp_to_joint_limits(wild.clone()).detach().cpu().numpy().copy()
    # ---- geodesic distance
    g = torch.Generator().manual_seed(5)
    q1 = torch.randn(64, 4, generator=g)
    q2 = torch.randn(64, 4, generator=g)
    q1 /= q1.norm(dim=1, keepdim=True)
    q2 /= q2.norm(dim=1, keepdim=True)
    q2[0], q2[1] = q1[0], -q1[1]
    q1[2], q2[2] = torch.tensor([1.0, 0, 0, 0]), torch.tensor([0.0, 1, 0, 0])
    q2[3] = q1[3] + 1e-4 * torch.randn(4, generator=g)
    q2[3] /= q2[3].norm()
    out["geo_q1"], out["geo_q2"] = q1.numpy().copy(), q2.numpy().copy()
    out["geo_dist"] = geodesic_distance_between_quaternions(q1, q2).detach().cpu().numpy().copy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ikflow", default=None, help="path of an ikflow checkout (its own glow_cNF_model is used when importable)")
    args = ap.parse_args()
    if not libraries_available():
        print("FrEIA / jrl are not importable here: nothing generated (see this file's docstring for the two pip lines).")
        return 2
    vec = collect(args.ikflow)
    np.savez_compressed(OUT, **vec)
    print(f"wrote {OUT}: {len(vec)} arrays, {os.path.getsize(OUT) / 1024:.0f} KB")
    return 0


if __name__ == "__main__":
    sys.exit(main())
