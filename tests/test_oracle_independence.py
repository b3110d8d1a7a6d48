"""CPU tests: the oracle stands on its own tables, and those tables agree with the product's.

The oracle (oracle/*.py) imports nothing from ikflow_amd: constants, the split rule, state_dict key names, permutation
tables, the fixed linear transform, the released hyper-parameters, the robots' URDF chains and the weight generator are all
written down twice.  A mistake in either copy shows up HERE as a table mismatch (and on the GPU box as a parity failure);
a mistake shared by both copies can only be caught by reference-derived vectors: tests/golden/ref_vectors.npz (outputs of the
reference's own ikflow/model.py code, see tests/golden/make_ref_vectors.py) and the literals of tests/test_oracle_golden.py.
"""
import ast
import json
import os

import numpy as np
import pytest
import torch

from helpers import O, custom_model
from ikflow_amd import model as pm
from ikflow_amd import robots as pr
from oracle import flow_oracle as fo
from oracle import kinematics_oracle as ko
from oracle import robot_tables as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_oracle_imports_nothing_from_the_product_and_product_nothing_from_the_oracle():
    def imported(path):
        mods = set()
        for node in ast.walk(ast.parse(open(path).read())):
            if isinstance(node, ast.Import):
                mods |= {a.name.split(".")[0] for a in node.names}
            elif isinstance(node, ast.ImportFrom) and node.module:
                mods.add(node.module.split(".")[0])
        return mods

    for f in sorted(os.listdir(os.path.join(ROOT, "oracle"))):
        if f.endswith(".py"):
            assert "ikflow_amd" not in imported(os.path.join(ROOT, "oracle", f)), f"oracle/{f} imports the product"
    for f in sorted(os.listdir(os.path.join(ROOT, "ikflow_amd"))):
        if f.endswith(".py"):
            assert "oracle" not in imported(os.path.join(ROOT, "ikflow_amd", f)), f"ikflow_amd/{f} imports the oracle"


def test_constants_agree():
    assert pm.ATAN_CLAMP_GAIN == fo.GLOW_ATAN_GAIN == 0.636
    assert pm.LEAKY_RELU_SLOPE == fo.LEAKY_SLOPE == 0.01 == torch.nn.LeakyReLU().negative_slope
    assert pm.SIGMOID_SCALING_ABS_MAX == fo.SIGMOID_PAD_ABS_MAX == 1.0
    assert ko.LM_LAMBDA == 1e-4 and ko.LM_ALPHA == 1.0 and ko.ACOS_EPS == 1e-7


@pytest.mark.parametrize("name", ["panda", "fetch_arm", "fetch"])
def test_robot_tables_agree(name):
    prod, orc = pr.get_robot(name), rt.robot(name)
    assert prod.name == orc.name and prod.ndof == orc.ndof and len(prod.joints) == len(orc.joints)
    np.testing.assert_allclose(np.array(prod.actuated_joints_limits), np.array(orc.actuated_joints_limits), rtol=0, atol=1e-12)
    for a, b in zip(prod.joints, orc.joints):
        assert a.name == b.name and a.kind == b.kind, (a, b)
        np.testing.assert_allclose(a.origin_xyz, b.origin_xyz, atol=1e-12, err_msg=a.name)
        np.testing.assert_allclose(a.origin_rpy, b.origin_rpy, atol=1e-12, err_msg=a.name)
        if a.actuated:
            np.testing.assert_allclose(a.axis, b.axis, atol=1e-12, err_msg=a.name)
    assert (pr.JOINT_FIXED, pr.JOINT_REVOLUTE, pr.JOINT_PRISMATIC) == (rt.FIXED, rt.REVOLUTE, rt.PRISMATIC)
    # the sampling helper too (bench.py uses the product's, the tests the oracle's)
    a = prod.sample_joint_angles(32, 0.01, np.random.default_rng(3))
    b = orc.sample_joint_angles(32, 0.01, np.random.default_rng(3))
    np.testing.assert_array_equal(a, b)


def test_rpy_conventions_agree():
    rng = np.random.default_rng(0)
    for rpy in rng.uniform(-np.pi, np.pi, (50, 3)):
        np.testing.assert_allclose(pr.rpy_to_matrix(rpy), rt.rpy_matrix(rpy), atol=1e-14)
    # fixed-axis convention on a known case: roll 90 deg then yaw 90 deg (extrinsic x, then z)
    R = rt.rpy_matrix((np.pi / 2, 0.0, np.pi / 2))
    np.testing.assert_allclose(R @ np.array([0.0, 1.0, 0.0]), [0.0, 0.0, 1.0], atol=1e-15)
    np.testing.assert_allclose(R @ np.array([1.0, 0.0, 0.0]), [0.0, 1.0, 0.0], atol=1e-15)


def test_folded_chain_reproduces_the_oracle_fk():
    """engine.fold_chain (what the C-ABI receives) evaluated with a few lines of numpy against the oracle's joint-by-joint
    walk over ITS table: checks the product table, the fold and the oracle chain against each other."""
    from ikflow_amd.engine import fold_chain

    for name in ("panda", "fetch_arm", "fetch"):
        prod = pr.get_robot(name)
        joints, tool = fold_chain(prod)
        q = rt.robot(name).sample_joint_angles(16, 0.0, np.random.default_rng(1)).astype(np.float64)
        ref = ko.forward_kinematics(name, torch.from_numpy(q)).numpy()
        for r in range(q.shape[0]):
            T = np.eye(4)
            for (kind, ax, pre), qi in zip(joints, q[r]):
                P = np.eye(4)
                P[:3, :4] = pre
                T = T @ P
                Mq = np.eye(4)
                if kind == pr.JOINT_REVOLUTE:
                    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
                    Mq[:3, :3] = np.eye(3) + np.sin(qi) * K + (1 - np.cos(qi)) * (K @ K)
                else:
                    Mq[:3, 3] = ax * qi
                T = T @ Mq
            P = np.eye(4)
            P[:3, :4] = tool
            T = T @ P
            np.testing.assert_allclose(T[:3, 3], ref[r, :3], atol=1e-12)


def _cases():
    cases = [(n, False) for n in fo.RELEASED]
    return cases


@pytest.mark.parametrize("model_name", [n for n in fo.RELEASED if n != "tiny"])
def test_released_hyper_parameters_and_layouts_agree(model_name):
    rob, nb, dim, cfg, width, clamp = fo.RELEASED[model_name]
    d = pm.MODEL_DESCRIPTIONS[model_name]
    assert (d["robot_name"], d["nb_nodes"], d["dim_latent_space"], d["coeff_fn_config"], d["coeff_fn_internal_size"], d["rnvp_clamp"]) == (
        rob, nb, dim, cfg, width, clamp)
    lp = pm.layout_from(pm.hparams_for(model_name), pr.get_robot(rob))
    lo = fo.layout_for(model_name)
    assert fo.OracleLayout.of(lp) == lo
    assert (lp.split1, lp.split2, lp.module_offset + 1) == (lo.len1, lo.len2, lo.first_block_module)
    assert lp.flops_per_solution() == lo.flops_per_solution()
    assert sorted(pm.MODEL_DESCRIPTIONS) == sorted([n for n in fo.RELEASED if n != "tiny"] + list(fo.RELEASED_NO_CHAIN))
    for name, row in fo.RELEASED_NO_CHAIN.items():  # hyper-parameters only: the robot is data (Robot.from_urdf)
        d = pm.MODEL_DESCRIPTIONS[name]
        assert (d["robot_name"], d["nb_nodes"], d["dim_latent_space"], d["coeff_fn_config"], d["coeff_fn_internal_size"], d["rnvp_clamp"]) == row


def test_work_figures_of_the_product_counters():
    """SURVEY 8(d) / BASELINE.md section 3 literals (what bench.py's roofline is computed from)."""
    lay = pm.layout_from(pm.hparams_for("panda__full__lp191_5.25m"), pr.Panda())
    assert (lay.n_weights(), lay.flops_per_solution(), lay.weight_bytes(), lay.row_io_bytes()) == (50786304, 101572608, 203440800, 84)
    lay = pm.layout_from(pm.hparams_for("fetch_arm__large__mh186_9.25m"), pr.FetchArm())
    assert (lay.n_weights(), lay.flops_per_solution(), lay.weight_bytes(), lay.row_io_bytes()) == (67862528, 135725056, 271844608, 96)
    lay = pm.layout_from(pm.TINY_MODEL_PARAMS, pr.Panda())
    assert (lay.n_weights(), lay.flops_per_solution(), lay.weight_bytes()) == (426240, 852480, 1717464)
    assert (lay.split1, lay.split2, lay.dim_cond) == (4, 5, 8)


@pytest.mark.parametrize("kw", [
    dict(nb_nodes=3, dim=9, n_hidden=2, width=256),                                  # TINY
    dict(nb_nodes=2, dim=7, n_hidden=1, width=256),
    dict(nb_nodes=2, dim=8, n_hidden=4, width=512, robot_name="fetch"),
    dict(nb_nodes=2, dim=10, n_hidden=3, width=768, robot_name="fetch_arm"),
    dict(nb_nodes=3, dim=9, n_hidden=2, width=256, softflow=False, sigmoid=True),    # sigmoid_on_output graph
])
def test_state_dict_tables_agree(kw):
    """Key names, shapes, permutation tables, the fixed linear transform and even the drawn weights of the two independent
    generators coincide (both restate nn.Linear's default initialisation in the reference's construction order)."""
    robot, hp, lay_o, sd_o = custom_model(seed=5, **kw)
    lay_p = pm.layout_from(hp, robot)
    assert fo.OracleLayout.of(lay_p) == lay_o
    sd_p = pm.random_state_dict(lay_p, robot, seed=5)
    assert set(sd_p) == set(sd_o), set(sd_p) ^ set(sd_o)
    for k in sd_o:
        assert sd_p[k].shape == sd_o[k].shape and sd_p[k].dtype == sd_o[k].dtype, k
        np.testing.assert_array_equal(sd_p[k], sd_o[k], err_msg=k)
    pm.validate_state_dict(lay_p, sd_o)  # the product's loader-side check accepts the oracle's dict
    for i in range(lay_o.nb_nodes):
        assert pm.key_perm_inv(i, lay_p.module_offset) == f"module_list.{lay_o.perm_module(i)}.perm_inv"
        for which in (1, 2):
            for layer in range(lay_o.n_hidden + 1):
                assert pm.key_linear(i, which, layer, "weight", lay_p.module_offset) == f"module_list.{lay_o.glow_module(i)}.subnet{which}.{2 * layer}.weight"
        np.testing.assert_array_equal(pm.freia_permutation(lay_o.dim, i), fo.permute_random_tables(lay_o.dim, i)[0])


# ---- vectors produced by the reference's own code (tests/golden/make_ref_vectors.py) ----------------------------------
def test_reference_subnet_constructor_vectors():
    """ikflow/model.py:51-96 executed from the reference file: the oracle's subnet restatement reproduces its parameters
    (same nn.Linear construction order under the same seed), its Sequential key names and its outputs, for configs 1..4;
    the product's key naming matches those names."""
    z = np.load(os.path.join(GOLD, "ref_vectors.npz"))
    W, CIN, COUT = 32, 11, 8
    for n_layers in (1, 2, 3, 4):
        keys = json.loads(str(z[f"c{n_layers}_keys"]))
        assert json.loads(str(z[f"c{n_layers}_modules"])) == ["Linear", "LeakyReLU"] * n_layers + ["Linear"]
        assert (z[f"c{n_layers}_slopes"] == fo.LEAKY_SLOPE).all() and len(z[f"c{n_layers}_slopes"]) == n_layers
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(100 + n_layers)
            net = fo.make_subnet(W, n_layers, CIN, COUT)
        sd = net.state_dict()
        assert list(sd.keys()) == keys
        for k in keys:
            np.testing.assert_array_equal(sd[k].numpy(), z[f"c{n_layers}_{k}"], err_msg=f"config {n_layers} {k}")
        # the oracle's evaluator on a one-block dict carrying the REFERENCE's parameters
        lay = fo.OracleLayout(nb_nodes=1, dim=6, dim_cond=8, width=W, n_hidden=n_layers, clamp=2.5, ndof=6)
        one = {f"module_list.{lay.glow_module(0)}.subnet1.{k}": z[f"c{n_layers}_{k}"] for k in keys}
        y = fo.subnet_torch(one, lay, 0, 1, torch.from_numpy(z[f"c{n_layers}_x"]))
        np.testing.assert_allclose(y.numpy(), z[f"c{n_layers}_y"], rtol=0, atol=1e-6)
        # the product's key naming (what ikf_load_weights looks up)
        prod_keys = [pm.key_linear(0, 1, l, w).split("subnet1.")[1] for l in range(n_layers + 1) for w in ("weight", "bias")]
        assert prod_keys == keys


def test_reference_hyper_parameter_defaults():
    z = np.load(os.path.join(GOLD, "ref_vectors.npz"))
    assert pm.IkflowModelParameters().__dict__ == json.loads(str(z["hparam_defaults"]))
    assert pm.TINY_MODEL_PARAMS.__dict__ == json.loads(str(z["hparam_tiny"]))
    t = json.loads(str(z["hparam_tiny"]))
    assert fo.RELEASED["tiny"][1:] == (t["nb_nodes"], t["dim_latent_space"], t["coeff_fn_config"], t["coeff_fn_internal_size"], t["rnvp_clamp"])


def test_reference_fixed_linear_transform_and_flipped_sigmoid_vectors():
    """ikflow/model.py:191-233 and :120-146 executed from the reference file: rev = (x - b).mm(M_inv), fwd = x.mm(M) + b;
    flipped sigmoid rev = 1 / (1 + exp(-x)).  The oracle's scaling-node tables and its inverse-pass tail reproduce them."""
    z = np.load(os.path.join(GOLD, "ref_vectors.npz"))
    lay = fo.OracleLayout(nb_nodes=1, dim=9, dim_cond=7, width=32, n_hidden=1, clamp=2.5, ndof=7, sigmoid_on_output=True)
    M, M_inv, b = fo.fixed_linear_transform(lay, "panda")
    np.testing.assert_allclose(M, z["flt_M"], rtol=1e-7, atol=0)
    np.testing.assert_allclose(M_inv, z["flt_M_inv"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(b, z["flt_b"], rtol=1e-6, atol=1e-8)
    x01 = torch.from_numpy(z["flt_rev_in"])
    np.testing.assert_allclose((x01 - torch.from_numpy(b)).mm(torch.from_numpy(M_inv)).numpy(), z["flt_rev_out"], rtol=0, atol=2e-6)
    xq = torch.from_numpy(z["flt_fwd_in"])
    np.testing.assert_allclose((xq.mm(torch.from_numpy(M)) + torch.from_numpy(b)).numpy(), z["flt_fwd_out"], rtol=0, atol=2e-6)
    assert z["flt_logdet"][0] == -z["flt_logdet"][1]
    # the tail of flow_inverse_torch on a zero-block "flow": sigmoid, then the scaling node's inverse
    sd = {"module_list.0.M_inv": M_inv, "module_list.0.b": b}
    tail = fo.flow_inverse_torch(sd, fo.OracleLayout(0, 9, 7, 32, 1, 2.5, 7, True), torch.from_numpy(z["sig_rev_in"]), torch.zeros(6, 7))
    want = (torch.from_numpy(z["sig_rev_out"]) - torch.from_numpy(b)).mm(torch.from_numpy(M_inv))
    np.testing.assert_allclose(tail.numpy(), want.numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(1.0 / (1.0 + np.exp(-z["sig_rev_in"].astype(np.float64))), z["sig_rev_out"], rtol=2e-6)
    # the product's tables for the same node
    robot, hp, lay_o, _ = custom_model(nb_nodes=1, dim=9, n_hidden=1, width=256, softflow=False, sigmoid=True)
    Mp, Mp_inv, bp = pm.fixed_linear_transform(pm.layout_from(hp, robot), robot)
    np.testing.assert_array_equal(Mp_inv, M_inv)
    np.testing.assert_array_equal(bp, b)


def test_reference_helper_vectors_latent_limits_tiling():
    """ikflow_solver.draw_latent (:16-29), evaluation_utils.calculate_joint_limits_exceeded (:100-112) and
    _get_target_pose_batch (:22-34) executed from the reference files: the product's host-side mirrors and the oracle
    reproduce them (the HIP limit kernels are checked against the same vectors in tests/test_gpu_parity.py)."""
    from ikflow_amd import evaluation_utils as eu
    from ikflow_amd.ikflow_solver import draw_latent

    z = np.load(os.path.join(GOLD, "ref_vectors.npz"))
    torch.manual_seed(1234)
    np.testing.assert_array_equal(draw_latent("gaussian", 0.75, (5, 7), "cpu").numpy(), z["latent_gaussian"])
    np.testing.assert_array_equal(draw_latent("uniform", 2.0, (5, 7), "cpu").numpy(), z["latent_uniform"])
    cfg = torch.from_numpy(z["limits_cfg"])
    got = ko.calculate_joint_limits_exceeded(cfg, O("panda").actuated_joints_limits)
    np.testing.assert_array_equal(got.numpy(), z["limits_exceeded"])
    assert not z["limits_exceeded"][:14].any() and z["limits_exceeded"][14] and z["limits_exceeded"][15]  # strict, float32
    one = torch.arange(7, dtype=torch.float32)
    np.testing.assert_array_equal(eu._get_target_pose_batch(one, 4).numpy(), z["tpb_single"])
    batch = cfg[:5].clone()
    assert bool(z["tpb_batch_is_identity"]) and eu._get_target_pose_batch(batch, 5) is batch
