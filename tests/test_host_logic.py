"""CPU tests of the host side: API mirror (assert behaviour of ikflow_solver.py), weight-table validation, chain
folding, C-ABI library export table.  No compute call reaches the GPU here."""
import ctypes
import os
import pickle
import re

import numpy as np
import pytest
import torch

from helpers import panda_model, tiny_model
from ikflow_amd import _lib, config
from ikflow_amd.engine import fold_chain
from ikflow_amd.ikflow_solver import IKFlowSolver, draw_latent
from ikflow_amd.model import (MODEL_DESCRIPTIONS, TINY_MODEL_PARAMS, IkflowModelParameters, hparams_for, key_linear,
                              layout_from, random_state_dict, validate_state_dict)
from ikflow_amd.model_loading import get_all_model_names, get_ik_solver, model_filename
from ikflow_amd.robots import Fetch, FetchArm, Panda, get_robot
from oracle import kinematics_oracle as ko

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_exports_every_declared_symbol():
    """Both flavours of the library export every symbol of both headers; the boundary header (include/ikflow_amd.h: what a binding of the
    reference needs) carries no measurement / tuning entry point and stays one screen-and-a-bit long; those live in ikflow_amd_debug.h."""
    boundary = open(os.path.join(ROOT, "include", "ikflow_amd.h")).read()
    debug = open(os.path.join(ROOT, "include", "ikflow_amd_debug.h")).read()
    names = lambda text: set(re.findall(r"\b(ikf_[a-z_0-9]+)\s*\(", text)) - {"ikf_latent_fn", "ikf_seed_fn"}
    declared = names(boundary) | names(debug)
    assert names(boundary) and names(debug) and not (names(boundary) & names(debug))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for tuning in ("ikf_set_gemm_variant", "ikf_time_gemm", "ikf_profile_begin", "ikf_profile_end", "ikf_split_kernel_name", "ikf_dominant_kernel_name"):
        assert tuning in names(debug) and tuning not in names(boundary)
    assert len(boundary.splitlines()) <= 240
    for flavour in ("", "probes"):
        lib = _lib.load(flavour)
        for name in declared:
            assert hasattr(lib, name), f"{name} is declared in include/ but not exported by the {flavour or 'product'} library"
        assert lib.ikf_abi_version() == _lib.IKF_ABI_VERSION
        assert lib.ikf_dominant_kernel_name().decode() == "k_flow_gemm"
        assert bool(lib.ikf_probes_build()) == (flavour == "probes")
    assert ctypes.sizeof(_lib.ikf_joint) == 4 + 12 + 48 and ctypes.sizeof(_lib.ikf_model_desc) == 9 * 4 + 2 * 32 + 8 * 64 + 48 + 4


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_engine_fails_loudly_without_gpu():
    from ikflow_amd.engine import EngineError

    robot, hp, lay, sd = tiny_model()
    s = IKFlowSolver(hp, robot)
    s.load_state_dict_tensors(sd)
    with pytest.raises(EngineError, match="no CPU path"):
        s.generate_ik_solutions(torch.zeros(7), n=4)
    with pytest.raises(EngineError):
        robot.forward_kinematics(torch.zeros(1, 7))
    desc = _lib.ikf_model_desc()
    desc.abi_version = _lib.IKF_ABI_VERSION
    h = ctypes.c_void_p()
    assert _lib.load().ikf_create(ctypes.byref(desc), 0, ctypes.byref(h)) == _lib.IKF_ERR_NO_DEVICE
    assert "no HIP device" in _lib.last_error()


def test_solver_argument_asserts_mirror_reference():
    """ikflow_solver.py:309-326 - every assert fires before any device work."""
    robot, hp, lay, sd = tiny_model()
    s = IKFlowSolver(hp, robot)
    y = torch.zeros(7)
    with pytest.raises(AssertionError, match="Model weights have not been loaded"):
        s.generate_ik_solutions(y, n=3)
    s.load_state_dict_tensors(sd)
    with pytest.raises(AssertionError):
        s.generate_ik_solutions([0.0] * 7, n=3)  # y must be a tensor
    with pytest.raises(AssertionError):
        s.generate_ik_solutions(y)  # single pose needs n
    with pytest.raises(AssertionError):
        s.generate_ik_solutions(torch.zeros(1, 7))  # quirk Q2: [1 x 7] is the single-pose form
    with pytest.raises(AssertionError):
        s.generate_ik_solutions(y, n=0)
    with pytest.raises(AssertionError, match="y must be of shape"):
        s.generate_ik_solutions(torch.zeros(4, 6))
    with pytest.raises(AssertionError):
        s.generate_ik_solutions(y, n=3, latent_scale=1)  # quirk Q1: must be a float
    with pytest.raises(AssertionError, match="latent must either be"):
        s.generate_ik_solutions(y, n=3, latent=np.zeros((3, 9)))
    with pytest.raises(AssertionError, match="refine_solutions is deprecated"):
        s.generate_ik_solutions(y, n=3, refine_solutions=True)
    with pytest.raises(AssertionError, match="must be of shape"):
        s.generate_exact_ik_solutions(torch.zeros(4, 6))
    with pytest.raises(AssertionError, match="must be a tuple"):
        s.generate_exact_ik_solutions(torch.zeros(4, 7), repeat_counts=[1, 3])
    with pytest.raises(AssertionError, match="return_detailed is not currently supported"):
        s.generate_exact_ik_solutions(torch.zeros(4, 7), return_detailed=True)
    assert (s.robot.name, s.network_width, s.conditional_size, s.ndof) == ("panda", 9, 8, 7)


def test_constructor_asserts():
    with pytest.raises(AssertionError, match="IkflowModelParameters"):
        IKFlowSolver({"nb_nodes": 3}, Panda())
    with pytest.raises(AssertionError, match="Robot type"):
        IKFlowSolver(TINY_MODEL_PARAMS, "panda")
    hp = IkflowModelParameters()
    hp.sigmoid_on_output = True
    with pytest.raises(AssertionError, match="incompatible"):
        IKFlowSolver(hp, Panda())
    hp = IkflowModelParameters()
    del hp.sigmoid_on_output  # old checkpoints lack it (ikflow_solver.py:43-44)
    IKFlowSolver(hp, Panda())
    assert hp.sigmoid_on_output is False


def test_draw_latent():
    torch.manual_seed(0)
    a = draw_latent("gaussian", 0.75, (5, 7), "cpu")
    torch.manual_seed(0)
    assert torch.equal(a, 0.75 * torch.randn((5, 7)))
    u = draw_latent("uniform", 2.0, (1000, 3), "cpu")
    assert u.min() >= -2.0 and u.max() <= 2.0
    with pytest.raises(AssertionError):
        draw_latent("laplace", 1.0, (2, 2), "cpu")
    with pytest.raises(AssertionError):
        draw_latent("gaussian", 0.0, (2, 2), "cpu")


def test_state_dict_validation_and_pickle_roundtrip(tmp_path):
    robot, hp, _, sd = tiny_model()
    lay = layout_from(hp, robot)
    validate_state_dict(lay, sd)
    assert len([k for k in sd if k.endswith("weight")]) == lay.nb_nodes * 2 * (lay.n_hidden + 1)
    assert sd[key_linear(0, 1, 0, "weight")].shape == (256, 4 + 8) and sd[key_linear(0, 2, 2, "weight")].shape == (8, 256)
    bad = dict(sd)
    del bad[key_linear(1, 2, 1, "bias")]
    with pytest.raises(RuntimeError, match="Missing key"):
        validate_state_dict(lay, bad)
    bad = dict(sd)
    bad[key_linear(0, 1, 0, "weight")] = np.zeros((256, 11), np.float32)
    with pytest.raises(RuntimeError, match="size mismatch"):
        validate_state_dict(lay, bad)
    # the reference's file format: pickle of {name: torch.Tensor} (ikflow_solver.py:416-418), optionally "nn_model."-prefixed
    path = tmp_path / "w.pkl"
    with open(path, "wb") as f:
        pickle.dump({"nn_model." + k: torch.from_numpy(v) for k, v in sd.items()}, f)
    s = IKFlowSolver(hp, robot)
    s.load_state_dict(str(path))
    assert s._model_weights_loaded
    for k, v in sd.items():
        assert np.array_equal(s._state_dict_np[k], v)
    np.savez(tmp_path / "w.npz", **sd)
    s2 = IKFlowSolver(hp, robot)
    s2.load_state_dict(str(tmp_path / "w.npz"))
    assert s2._model_weights_loaded
    with open(tmp_path / "junk.pkl", "wb") as f:
        f.write(b"not a pickle")
    with pytest.raises(pickle.UnpicklingError):
        IKFlowSolver(hp, robot).load_state_dict(str(tmp_path / "junk.pkl"))


def test_model_registry():
    assert set(get_all_model_names()) == set(MODEL_DESCRIPTIONS)
    assert model_filename("https://storage.googleapis.com/ikflow_models/atlas_desert-sweep-6.pkl") == "atlas_desert-sweep-6.pkl"
    hp = hparams_for("fetch_arm__large__mh186_9.25m")
    assert (hp.nb_nodes, hp.dim_latent_space, hp.coeff_fn_internal_size, hp.softflow_enabled) == (16, 10, 1024, True)
    with pytest.raises(AssertionError, match="not found in model descriptions"):
        get_ik_solver("panda_tpm")
    with pytest.raises(FileNotFoundError):
        get_ik_solver("panda_lite_tpm")
    s, hp = get_ik_solver("panda_lite_tpm", synthetic_weights_seed=1)
    assert s._model_weights_loaded and hp.nb_nodes == 6 and s.robot.name == "panda"
    with pytest.raises(ValueError):
        get_robot("atlas")


def test_fold_chain_matches_joint_by_joint_fk():
    """The engine walks actuated joints with the fixed URDF transforms pre-multiplied; same FK as the oracle's walk."""
    for robot in (Panda(), FetchArm(), Fetch()):
        joints, tool = fold_chain(robot)
        assert len(joints) == robot.ndof
        q = robot.sample_joint_angles(16, 0.0, np.random.default_rng(3)).astype(np.float64)
        ref = ko.forward_kinematics(robot, torch.from_numpy(q)).numpy()
        for r in range(q.shape[0]):
            T = np.eye(4)
            for (kind, ax, pre), qi in zip(joints, q[r]):
                F = np.eye(4)
                F[:3, :4] = pre
                T = T @ F
                Mo = np.eye(4)
                if kind == 1:
                    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
                    Mo[:3, :3] = np.eye(3) + np.sin(qi) * K + (1 - np.cos(qi)) * (K @ K)
                else:
                    Mo[:3, 3] = ax * qi
                T = T @ Mo
            F = np.eye(4)
            F[:3, :4] = tool
            T = T @ F
            np.testing.assert_allclose(T[:3, 3], ref[r, :3], atol=1e-12)


def test_config_and_device_rule():
    assert config.DEFAULT_TORCH_DTYPE == torch.float32
    assert config.DEVICE == ("cuda:0" if torch.cuda.is_available() else "cpu") or config.DEVICE.startswith("cuda:")


def test_evaluation_utils_host_side():
    """target-pose tiling rule of evaluation_utils.py:22-34 and the no-CPU-path contract of the mirror module."""
    from ikflow_amd import evaluation_utils as eu
    from ikflow_amd.engine import EngineError

    one = torch.arange(7, dtype=torch.float32)
    assert eu._get_target_pose_batch(one, 4).shape == (4, 7)
    assert eu._get_target_pose_batch(one.numpy(), 3).shape == (3, 7)
    batch = torch.zeros(5, 7)
    assert eu._get_target_pose_batch(batch, 5) is batch
    if not torch.cuda.is_available():
        with pytest.raises(EngineError):
            eu.pose_errors(batch, batch)
        with pytest.raises(EngineError):
            eu.calculate_joint_limits_exceeded(torch.zeros(2, 3), [(-1, 1)] * 3)


def test_collision_capsule_folding_and_pairs():
    """Robot.set_collision_capsules: fixed URDF offsets folded into the frame of the preceding actuated joint; pairs = all
    capsule pairs on different moving frames minus the ignored ones."""
    robot = Panda()
    assert not robot.has_collision_model
    caps = [(None, (0, 0, 0), (0, 0, 0.3), 0.06), ("panda_joint7", (0, 0, 0), (0, 0, 0.05), 0.04),
            ("panda_joint8", (0, 0, 0), (0, 0, 0.05), 0.04), ("panda_hand_joint", (0.01, 0, 0), (0.01, 0, 0.05), 0.03)]
    robot.set_collision_capsules(caps, ignored_pairs=[(1, 0)])
    folded, pairs = robot._collision_model
    assert [f[0] for f in folded] == [0, 7, 7, 7]
    np.testing.assert_allclose(folded[2][1], (0, 0, 0.107), atol=1e-12)  # panda_joint8 origin (fixed, z = 0.107)
    c, s_ = np.cos(-np.pi / 4), np.sin(-np.pi / 4)  # panda_hand_joint: rpy (0, 0, -pi/4) behind joint8
    np.testing.assert_allclose(folded[3][1], (c * 0.01, s_ * 0.01, 0.107), atol=1e-12)
    assert pairs == [(0, 2), (0, 3)]  # (0,1) ignored; 1, 2, 3 ride on the same frame
    with pytest.raises(AssertionError):
        robot.set_collision_capsules([("no_such_joint", (0, 0, 0), (0, 0, 1), 0.1)])


def test_bench_and_entry_contract_host_side(monkeypatch):
    """bench.py defaults (N = 1, a K / W that finish within minutes, the BASELINE batch), the committed PMC summary it
    reports `roofline.traffic` from, and the two driver entry points."""
    import sys

    import bench

    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.batch, a.precision, a.million, a.no_cells, a.dist_dry_run, a.no_live_pmc) == (1, 4096, "f32", False, False, False, False)
    assert 0 < a.warmup < a.steps <= 100
    # the newest committed profile round is of the row-owner launch (one launch per 4096-row step: 2.8 ms, 8 x the 203 MB weight
    # stream through the eight XCD L2s); a summary of another dominant kernel is not reported as this one's
    us, src = bench.rocprof_kernel_avg_us("k_flow_rowowner")
    assert 2500.0 < us < 3200.0 and src.endswith("_bench_kernel_stats.csv")
    traffic, src = bench.pmc_traffic_per_launch("k_flow_rowowner")
    assert isinstance(traffic, int) and 8 * 200_000_000 < traffic < 9 * 210_000_000 and src.endswith("_pmc_summary.json")
    assert bench.pmc_traffic_per_launch("k_flow_gemm<") == (None, None)
    b, src = bench.committed_call_traffic("b128")
    assert b is not None and 2e8 < b < 6e8 and src.endswith("_pmc_summary_b128.json")
    assert bench.FP32_MFMA_PEAK_TFLOPS == 157.3
    import __graft_entry__ as ge

    assert callable(ge.build) and callable(ge.smoke)


def test_verify_weights_tool_on_a_synthetic_pickle(tmp_path):
    """tools/verify_weights.py on a pickle laid out like a released file: torch tensors, "nn_model." prefix
    (scripts/download_model_from_wandb_checkpoint.py:13-28), M / M_inv / b / logDetM of module 0, perm + perm_inv."""
    import subprocess
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import verify_weights as vw

    robot, hp, lay, sd = panda_model(seed=2)
    full = {"nn_model." + k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    full["nn_model.module_list.0.logDetM"] = torch.tensor(-6.1)
    path = tmp_path / "panda__lyric-puddle-191__global_step%3D5.25M.pkl"
    with open(path, "wb") as f:
        pickle.dump(full, f)
    loaded = vw.load_any(str(path))
    assert "module_list.0.M_inv" in loaded and all(not k.startswith("nn_model.") for k in loaded)
    arch = vw.infer_architecture(loaded)
    assert arch == dict(nb_nodes=12, dim=7, width=1024, n_hidden=3, dim_cond=8, sigmoid_on_output=False)
    ok, rep = vw.check_structure(loaded, "panda__full__lp191_5.25m")
    assert ok and rep["keys_and_shapes"] == "ok" and rep["permutations_equal_numpy_seed_i"] and rep["M_inv_diag_matches_joint_limits"]
    assert rep["unknown_keys"] == [] and rep["f16x3_mode_usable"] and rep["layout"]["weight_bytes"] == 203440800
    # a file whose permutation tables are not the seed-i ones is reported (and still loadable); a broken one fails
    other = dict(loaded)
    other["module_list.5.perm_inv"] = np.roll(other["module_list.5.perm_inv"], 1)
    other["module_list.5.perm"] = np.argsort(other["module_list.5.perm_inv"])
    ok2, rep2 = vw.check_structure(other, "panda__full__lp191_5.25m")
    assert ok2 and rep2["permutation_blocks_differing"] == [2]
    other["module_list.5.perm_inv"] = np.zeros(7, dtype=np.int64)
    assert not vw.check_structure(other, "panda__full__lp191_5.25m")[0]
    wrong = dict(loaded)
    wrong["module_list.0.M_inv"] = wrong["module_list.0.M_inv"] * 2.0
    assert not vw.check_structure(wrong, "panda__full__lp191_5.25m")[0]
    assert not vw.check_structure(loaded, "fetch_arm__large__mh186_9.25m")[0]  # wrong architecture for the file
    # the command line: structure fine, no GPU here -> exit 0 with the pose-error leg skipped
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verify_weights.py"), str(path), "--model",
                            "panda__full__lp191_5.25m"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        assert "skipped: no GPU" in r.stdout


def test_sigmoid_graph_requires_the_scaling_offset():
    """module_list.0.b of the sigmoid_on_output graph is -slope * lo, never zero: a file without it is refused."""
    from helpers import custom_model

    robot, hp, _, sd = custom_model(nb_nodes=2, dim=9, n_hidden=2, width=256, softflow=False, sigmoid=True)
    lay = layout_from(hp, robot)
    validate_state_dict(lay, sd)
    bad = {k: v for k, v in sd.items() if k != "module_list.0.b"}
    with pytest.raises(RuntimeError, match="module_list.0.b"):
        validate_state_dict(lay, bad)
    plain_robot, plain_hp, _, plain_sd = tiny_model()
    validate_state_dict(layout_from(plain_hp, plain_robot), {k: v for k, v in plain_sd.items() if k != "module_list.0.b"})  # optional there


def test_kinematics_engine_cache_key_distinguishes_same_named_robots():
    from ikflow_amd.engine import _robot_key
    from ikflow_amd.robots import Joint, Robot

    a, b = Panda(), Panda()
    assert _robot_key(a) == _robot_key(b)
    joints = list(a.joints)
    j3 = joints[3]
    joints[3] = Joint(j3.name, j3.kind, j3.origin_xyz, j3.origin_rpy, j3.axis, (-3.0, -0.1))  # edited limits, same name
    assert _robot_key(Robot("panda", joints)) != _robot_key(a)


def test_approximate_panda_capsule_model_host_side():
    from ikflow_amd.robots import PANDA_APPROX_CAPSULES

    robot = Panda().use_approximate_collision_model()
    folded, pairs = robot._collision_model
    assert len(folded) == len(PANDA_APPROX_CAPSULES) == 12 and [f[0] for f in folded] == [0, 1, 2, 2, 3, 4, 4, 5, 6, 7, 7, 7]
    assert all(abs(folded[a][0] - folded[b][0]) > 1 for a, b in pairs) and len(pairs) == 45
    with pytest.raises(ValueError, match="only for 'panda'"):
        FetchArm().use_approximate_collision_model()


def _plan_for(lib, n_cu, rows, ro=1, cl=1):
    buf = ctypes.create_string_buffer(1024)
    assert lib.ikf_plan_describe_for(n_cu, rows, ro, cl, buf, 1024) == 0
    return buf.value.decode()


def test_planner_decisions_and_invariants_host_side():
    """plan_flow is host logic: through the handle-free entry point the CPU suite checks what the GPU test
    `test_plan_of_a_call_by_batch_size` checks on a live handle - the decisions of section 4.0's table on a 256-CU chip - and, over every row
    count up to three rounds on several chip sizes, what must hold whatever the cost tables say: the chunks add up to the call, every
    cluster chunk's grid (row tiles x members) fits the chip with one workgroup per CU, no chunk is empty, and - on the 256-CU chip the costs
    were measured on - nothing goes to the per-layer kernels where the resident-row forms are allowed (one weight image for every size)."""
    import time

    lib = _lib.load()
    want = {1: "cluster32:1", 8: "cluster32:8", 128: "cluster32:128", 129: "cluster16:129", 256: "cluster16:256", 257: "cluster8:257",
            512: "cluster8:512", 513: "cluster8:512 cluster32:1", 600: "cluster8:512 cluster32:88", 1024: "cluster4:1024",
            1025: "cluster4:1024 cluster32:1", 1536: "cluster8:512 cluster4:1024", 2048: "cluster2:2048", 2049: "cluster2:2048 cluster32:1",
            2560: "cluster8:512 cluster2:2048", 3072: "cluster4:1024 cluster2:2048", 3400: "rowowner:3400", 4096: "rowowner:4096",
            4097: "rowowner:4096 cluster32:1", 4296: "rowowner:4096 cluster16:200", 8192: "rowowner:8192", 15788: "rowowner:15788",
            125000: "rowowner:122880 cluster2:2048 cluster32:72"}
    got = {n: _plan_for(lib, 256, n) for n in want}
    assert got == want, {n: (got[n], want[n]) for n in want if got[n] != want[n]}
    assert _plan_for(lib, 256, 0) == "" and _plan_for(lib, 256, 300, 0, 0) == "perlayer:300"
    assert _plan_for(lib, 256, 4096, 1, 0) == "rowowner:4096" and _plan_for(lib, 256, 4096, 0, 1) == "cluster2:2048 cluster2:2048"
    assert _plan_for(lib, 256, 200, 1, 0) == "perlayer:200"   # (a short call without the cluster form: not worth a row-owner round)
    buf = ctypes.create_string_buffer(8)
    assert lib.ikf_plan_describe_for(256, 125000, 1, 1, buf, 8) != 0 and lib.ikf_plan_describe_for(0, 10, 1, 1, buf, 8) != 0
    t0 = time.perf_counter()
    checked = 0
    for n_cu in (256, 240, 304, 64, 8):
        round_rows = 16 * n_cu
        for rows in list(range(1, 3 * round_rows + 40, 7 if n_cu > 64 else 1)) + [round_rows - 1, round_rows, round_rows + 1, 10 * round_rows + 5]:
            chunks = [c.split(":") for c in _plan_for(lib, n_cu, rows).split()]
            assert sum(int(r) for _, r in chunks) == rows and all(int(r) > 0 for _, r in chunks), (n_cu, rows, chunks)
            for form, r in chunks:
                assert form != "perlayer" or n_cu != 256, (n_cu, rows, chunks)   # (the chip the costs were measured on: one weight image for every size)
                if form.startswith("cluster"):
                    g = int(form[len("cluster"):])
                    assert g in (2, 4, 8, 16, 32) and (int(r) + 15) // 16 * g <= n_cu, (n_cu, rows, chunks)
            assert sum(1 for f, _ in chunks if f == "rowowner") <= 1 and (chunks[0][0] == "rowowner" or rows < round_rows), (n_cu, rows, chunks)
            checked += 1
    assert (time.perf_counter() - t0) / checked < 1e-3   # planned on the host in front of every call
    # without the row-owner launch (a forced tile variant, ikf_set_gemm_variant 180 + 187) the cluster form takes the whole batch: full
    # 2-member launches are peeled off iteratively - a million rows once overflowed the stack of the recursive tail planner
    big = 16 * 1024 * 1024 * 2
    buf = ctypes.create_string_buffer(big)
    for n_cu in (256, 64):
        for rows in (1_000_000, 10_000_000):
            t0 = time.perf_counter()
            assert lib.ikf_plan_describe_for(n_cu, rows, 0, 1, buf, big) == 0
            dt = time.perf_counter() - t0
            chunks = [c.split(":") for c in buf.value.decode().split()]
            assert sum(int(r) for _, r in chunks) == rows and all(f.startswith("cluster") for f, _ in chunks)
            assert all((int(r) + 15) // 16 * int(f[len("cluster"):]) <= n_cu for f, r in chunks)
            assert dt < 0.5, dt


def test_kernel_trace_steady_state_summary(tmp_path):
    """tools/kernel_trace_steady.py (what bench.py's roofline.frac is computed from): raw mean, median and the steady-state mean - a launch counts
    once 10 launches of the kernel have run since the queue's last idle gap (> 200 us): the clock-ramp launches behind a gap stay out."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_trace_steady import kernel_trace_stats

    rows, t = [], 1_000_000
    def launch(name, dur_ns, gap_ns=2_000):
        nonlocal t
        t += gap_ns
        rows.append((name, t, t + dur_ns))
        t += dur_ns
    launch("void ikf::k_rowowner_pack(ikf::RoPackArgs)", 10_000)
    for i in range(30):                       # behind the start of the trace: 10 ramp launches (slow), then steady ones
        launch("void ikf::k_flow_rowowner<4>(ikf::RoArgs)", 3_500_000 if i < 10 else 2_800_000)
    launch("void ikf::k_flow_rowowner<4>(ikf::RoArgs)", 3_400_000, gap_ns=50_000_000)   # an idle gap: the ramp starts over
    for i in range(19):
        launch("void ikf::k_flow_rowowner<4>(ikf::RoArgs)", 3_000_000 if i < 9 else 2_800_000)
    launch("void ikf::k_flow_gemm_skinny<true, 2>(ikf::FusedGemmArgs)", 1_000)            # ("skinny" kernels are never the dominant one)
    p = tmp_path / "kt_kernel_trace.csv"
    with open(p, "w") as f:
        f.write('"Kind","Kernel_Name","Start_Timestamp","End_Timestamp"\n')
        for name, a, b in rows:
            f.write(f'"KERNEL_DISPATCH","{name}",{a},{b}\n')
    s = kernel_trace_stats(str(p), "k_flow_rowowner")
    assert s["launches"] == 50 and s["steady_launches"] == 30
    assert abs(s["steady_mean_us"] - 2800.0) < 1e-6 and abs(s["median_us"] - 2800.0) < 1e-6
    assert s["mean_us"] > 2950.0 and s["max_us"] == 3500.0
    assert kernel_trace_stats(str(p), "k_split_gemm") is None
