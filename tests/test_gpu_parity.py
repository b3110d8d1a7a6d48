"""GPU parity tests: the HIP path (through the C-ABI) against the oracle on the same seeded inputs.

Tolerances: flow joint angles 1e-5 absolute vs the PyTorch-CPU fp32 oracle (BASELINE.json north_star), also bounded
by the fp64 twin; FK 2e-6; LM step vs the fp64 twin 5e-6 (the kernel solves in fp64 internally).
"""
import os

import numpy as np
import time

import pytest
import torch

from helpers import O, custom_model, fetch_arm_model, latents, panda_model, reachable_poses, released_model, tiny_model
from ikflow_amd.ikflow_solver import IKFlowSolver
from oracle import flow_oracle as fo
from oracle import kinematics_oracle as ko

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FLOW_TOL = 1e-5


def _solver(robot, hp, sd, flavour=""):
    """flavour "probes": the handle lives in lib/libikflow_amd_probes.so (-DIKF_PROBES) - the product's kernels plus the priced-and-rejected
    forms of rounds 2 - 3 (ikf_set_gemm_variant 106 / 108 / 121 / 163 / 164 / 171), kept reproducible there and out of the shipped library."""
    s = IKFlowSolver(hp, robot)
    s.library_flavour = flavour
    s.load_state_dict_tensors(sd)
    return s


def _flow_case(model, n, clamp=True, seed=0):
    robot, hp, lay, sd = model
    _, poses = reachable_poses(robot, n, seed)
    lat = latents(n, lay.dim, seed + 1)
    ref32 = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat, clamp=clamp)
    cond = torch.cat([poses, torch.zeros(n, 1)], 1).numpy()
    ref64 = fo.run_inference_f64(sd, lay, robot, lat.numpy(), cond, clamp)
    s = _solver(robot, hp, sd)
    # quirk Q2 (ikflow_solver.py:313-315,333): a [1 x 7] y has numel()==7 and is the single-pose form -> needs n
    got = s.generate_ik_solutions(poses.to(DEV), n=(1 if n == 1 else None), latent=lat.to(DEV), clamp_to_joint_limits=clamp).cpu()
    return got, ref32, ref64


@pytest.mark.parametrize("n", [1, 3, 16, 127, 128, 129, 300])
def test_flow_tiny_matches_oracle(n):
    got, ref32, ref64 = _flow_case(tiny_model(), n)
    assert got.shape == ref32.shape and got.dtype == torch.float32
    assert (got - ref32).abs().max().item() <= FLOW_TOL
    assert np.abs(got.numpy() - ref64).max() <= FLOW_TOL


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_flow_tiny_every_tile_boundary_and_forced_config(precision):
    """Row counts on both sides of every tile-picker boundary (256 / 512 / 1024 / 1536 / 2048 rows) in one solver, then
    the same rows with each tile configuration forced: every contraction kernel, its row clamping and its padded stores
    against the oracle."""
    robot, hp, lay, sd = tiny_model()
    s = _solver(robot, hp, sd)
    s.set_precision(precision)
    n_max = 2100
    _, poses = reachable_poses(robot, n_max, 21)
    lat = latents(n_max, lay.dim, 22)
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat, clamp=True)
    for n in (2, 31, 33, 255, 256, 257, 511, 512, 513, 768, 769, 1023, 1025, 1536, 1537, 2048, 2049, 2100):
        got = s.generate_ik_solutions(poses[:n].to(DEV), latent=lat[:n].to(DEV)).cpu()
        assert (got - ref[:n]).abs().max().item() <= FLOW_TOL, f"{precision} n={n}"
    eng = s.engine(DEV)
    for variant in (101, 102, 103, 104, 105, 107):  # tile configs 0..4 and 6 (ikf_set_gemm_variant)
        eng.set_gemm_variant(variant)
        for n in (1, 100, 300, 700):
            got = s.generate_ik_solutions(poses[:n].to(DEV), n=(1 if n == 1 else None), latent=lat[:n].to(DEV)).cpu()
            assert (got - ref[:n]).abs().max().item() <= FLOW_TOL, f"{precision} variant={variant} n={n}"
    eng.set_gemm_variant(-1)


@pytest.mark.parametrize("n,clamp", [(16, True), (500, True), (512, False)])
def test_flow_panda_matches_oracle(n, clamp):
    got, ref32, ref64 = _flow_case(panda_model(), n, clamp)
    # clamped outputs are joint angles inside the limits: absolute 1e-5.  Unclamped flow outputs reach |x| of 10 and more, where 1e-5
    # absolute is a handful of fp32 ulps of the value itself (and the torch-CPU oracle is 5.5e-6 from its fp64 twin there): relative to
    # max(1, |x|), as in every other unclamped comparison of this file
    scale = np.ones_like(ref64) if clamp else np.maximum(1.0, np.abs(ref64))
    d32 = np.abs(got.numpy() - ref32.numpy())
    e32 = (d32 / scale).max()
    e64 = (np.abs(got.numpy() - ref64) / scale).max()
    o64 = (np.abs(ref32.numpy() - ref64) / scale).max()
    w = np.unravel_index(d32.argmax(), d32.shape)
    print(f"panda n={n}: |hip-cpu32|={e32:.3e} |hip-f64|={e64:.3e} |cpu32-f64|={o64:.3e} (largest absolute difference {d32.max():.3e} at row {w[0]} "
          f"joint {w[1]}, value {ref64[w]:.3f})")
    assert e32 <= FLOW_TOL
    assert e64 <= FLOW_TOL


@pytest.mark.parametrize("gain,n", [(2.0, 256), (2.0, 4096), (2.5, 4096), (2.5, 100), (2.5, 256), (2.5, 512), (2.5, 1024), (2.5, 2048), (2.5, 3000)])
def test_flow_panda_trained_like_gain(gain, n):
    """Coupling coefficients of O(1) - last-Linear outputs scaled by `gain`, so atan / exp work away from 0 and the clamp saturates:
    every row of the batch through every form of the plan (100 / 256 / 512 / 1024 / 2048 rows: the cluster form with 32 / 16 / 8 / 4 / 2 members;
    3000: two cluster chunks; 4096: the row-owner launch), unclamped outputs, 1e-5 relative to max(1, |x|) against the fp64 twin, and no further
    from it than twice the torch-CPU oracle is (+ 1e-6)."""
    robot, hp, lay, sd = panda_model(seed=3, gain=gain)
    _, poses = reachable_poses(robot, n, 5)
    lat = latents(n, lay.dim, 6)
    cond = torch.cat([poses, torch.zeros(n, 1)], 1)
    ref32 = fo.flow_inverse_torch(sd, lay, lat, cond)[:, : lay.ndof]
    ref64 = fo.flow_inverse_f64(sd, lay, lat.numpy(), cond.numpy())[:, : lay.ndof]
    s = _solver(robot, hp, sd)
    got = s.generate_ik_solutions(poses.to(DEV), latent=lat.to(DEV), clamp_to_joint_limits=False).cpu()
    assert bool(torch.isfinite(got).all())
    scale = np.maximum(1.0, np.abs(ref64))
    e64 = (np.abs(got.numpy() - ref64) / scale).max()
    o64 = (np.abs(ref32.numpy() - ref64) / scale).max()
    print(f"gain {gain} n={n}: rel |hip-f64|={e64:.3e} |cpu32-f64|={o64:.3e}, max |x| {np.abs(ref64).max():.1f}")
    # (measured r04 at gain 2.5: 1.6e-6 / 2.1e-6 / 2.5e-6 / 3.7e-6 / 3.7e-6 / 3.7e-6 / 3.4e-6 for 100 ... 4096 rows against the oracle's own
    # 1.9e-6 ... 2.7e-6; with ONE f32 chain of 1024 products per output the row-owner launch was at 8.1e-6, the 8- and 2-member cluster forms at
    # 4.4e-6 / 5.5e-6 - every form now runs two half-length chains, or shorter ones where waves split the k range)
    assert e64 <= FLOW_TOL and e64 <= 2 * o64 + 1e-6


@pytest.mark.parametrize("which,n", [("panda", 500), ("panda", 4096), ("tiny", 300), ("tiny", 5000), ("fetch_arm", 200), ("fetch_arm", 4200)])
def test_flow_f16_split_precision_matches_oracle(which, n):
    """precision="f16x3": hidden contractions as three f16 MFMA products of error-compensated hi/lo operands.
    Same 1e-5 tolerance against the PyTorch-CPU oracle, and no further from the fp64 twin than the exact-f32 path is."""
    model = {"panda": panda_model, "tiny": tiny_model, "fetch_arm": fetch_arm_model}[which]()
    robot, hp, lay, sd = model
    _, poses = reachable_poses(robot, n, 11)
    lat = latents(n, lay.dim, 12)
    m = min(n, 512)  # oracle on a slice for the big batch
    ref32 = fo.generate_ik_solutions_torch(sd, lay, robot, poses[:m], lat[:m], clamp=False)
    cond = torch.cat([poses[:m], torch.zeros(m, 1)], 1).numpy()
    ref64 = fo.run_inference_f64(sd, lay, robot, lat[:m].numpy(), cond, False)
    s = _solver(robot, hp, sd)
    got32 = s.generate_ik_solutions(poses.to(DEV), latent=lat.to(DEV), clamp_to_joint_limits=False).cpu()[:m]
    s.set_precision("f16x3")
    assert s.engine(DEV).precision == "f16x3"
    got16 = s.generate_ik_solutions(poses.to(DEV), latent=lat.to(DEV), clamp_to_joint_limits=False).cpu()[:m]
    scale = np.maximum(1.0, np.abs(ref64))
    e16_64 = (np.abs(got16.numpy() - ref64) / scale).max()
    e32_64 = (np.abs(got32.numpy() - ref64) / scale).max()
    e16_cpu = ((got16 - ref32).abs().numpy() / scale).max()
    print(f"{which} n={n}: |f16x3 - f64| {e16_64:.3e}   |f32mfma - f64| {e32_64:.3e}   |f16x3 - cpu32| {e16_cpu:.3e}")
    assert e16_cpu <= FLOW_TOL and e16_64 <= FLOW_TOL
    assert e16_64 <= 1.5 * e32_64 + 5e-7
    clamped = s.generate_ik_solutions(poses.to(DEV), latent=lat.to(DEV)).cpu()[:m]
    ref_cl = fo.generate_ik_solutions_torch(sd, lay, robot, poses[:m], lat[:m], clamp=True)
    assert (clamped - ref_cl).abs().max().item() <= FLOW_TOL


def test_sigmoid_on_output_variant_matches_oracle_and_stays_in_limits():
    """ikflow/model.py:304-307 graph (scaling node + flipped sigmoid, no softflow -> 7-entry conditional);
    reference tests/model_test.py:108-123 property on the HIP path."""
    from test_oracle_golden import _sigmoid_model

    robot, hp, lay, sd = _sigmoid_model(seed=1)
    s = _solver(robot, hp, sd)
    assert s.conditional_size == 7
    n = 300
    _, poses = reachable_poses(robot, n, 3)
    lat = latents(n, lay.dim, 4)
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat, clamp=False)
    got = s.generate_ik_solutions(poses.to(DEV), latent=lat.to(DEV), clamp_to_joint_limits=False).cpu()
    assert (got - ref).abs().max().item() <= FLOW_TOL
    wild = 1e8 * latents(n, lay.dim, 5)
    out = s.generate_ik_solutions(poses.to(DEV), latent=wild.to(DEV), clamp_to_joint_limits=False).cpu()
    assert bool(torch.isfinite(out).all())
    for i, (lo, hi) in enumerate(O(robot).actuated_joints_limits):
        assert out[:, i].min().item() >= lo - 1e-5 and out[:, i].max().item() <= hi + 1e-5


def test_flow_fetch_arm_matches_oracle():
    got, ref32, ref64 = _flow_case(fetch_arm_model(), 200)
    assert (got - ref32).abs().max().item() <= FLOW_TOL
    assert np.abs(got.numpy() - ref64).max() <= FLOW_TOL


def test_single_pose_form_and_latent_draw():
    robot, hp, lay, sd = tiny_model()
    s = _solver(robot, hp, sd)
    y = torch.tensor([0.25, 0.0, 0.5, 1.0, 0.0, 0.0, 0.0])
    lat = latents(16, lay.dim, 0)
    got = s.generate_ik_solutions(y.to(DEV), n=16, latent=lat.to(DEV)).cpu()
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, y, lat, n=16)
    assert (got - ref).abs().max().item() <= FLOW_TOL
    # batch form of the same pose gives the same rows
    got_b = s.generate_ik_solutions(y.expand(16, 7).contiguous().to(DEV), latent=lat.to(DEV)).cpu()
    assert torch.equal(got, got_b)
    # drawn latents: same torch generator call as the reference's draw_latent
    torch.manual_seed(7)
    a = s.generate_ik_solutions(y.to(DEV), n=8)
    torch.manual_seed(7)
    lat2 = 1.0 * torch.randn((8, lay.dim), device=DEV)
    b = s.generate_ik_solutions(y.to(DEV), n=8, latent=lat2)
    assert torch.equal(a, b)
    assert a.device.type == "cuda" and a.shape == (8, 7)


def test_reference_property_test_solve_multiple_poses():
    """tests/ikflow_solver_test.py:89-117 on the engine: equal rows -> equal outputs; different pose -> all differ."""
    from ikflow_amd.model import TINY_MODEL_PARAMS
    from ikflow_amd.robots import Panda

    s = IKFlowSolver(TINY_MODEL_PARAMS, Panda())
    ys = torch.zeros(2, 7, device=DEV)
    latent = torch.zeros(2, 9, device=DEV)
    sols = s.generate_ik_solutions(ys, None, latent=latent, refine_solutions=False, allow_uninitialized=True)
    torch.testing.assert_close(sols[0], sols[1])
    ys = torch.tensor([[0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0]], device=DEV, dtype=torch.float32)
    # unclamped: with a zero latent an untrained flow puts joint 4 near 0, outside Panda's [-3.07, -0.0698], and the
    # clamp would map both rows to the same limit value
    sols = s.generate_ik_solutions(ys, None, latent=latent, refine_solutions=False, allow_uninitialized=True, clamp_to_joint_limits=False)
    t1, t2 = sols[0][None, :], sols[1][None, :]
    for j in range(t1.shape[1]):
        assert ((t2 - t1[0, j]).abs() < 1e-8).sum().item() == 0


def test_full_batch_properties_4096():
    """BASELINE config 2 size: row independence / determinism at B=4096 (size-independent properties)."""
    robot, hp, lay, sd = panda_model()
    s = _solver(robot, hp, sd)
    n = 4096
    _, poses = reachable_poses(robot, n, 0)
    lat = latents(n, lay.dim, 1)
    P, L = poses.to(DEV), lat.to(DEV)
    full = s.generate_ik_solutions(P, latent=L)
    again = s.generate_ik_solutions(P, latent=L)
    assert torch.equal(full, again)  # deterministic
    # rows are independent: any sub-batch reproduces its rows - bit-for-bit when it runs in the same tile configuration
    # (the k-order of every contraction and the 64-column slot order of the last Linear are fixed), and in any case
    # far inside the parity tolerance
    for lo, hi in [(0, 1), (5, 133), (4000, 4096), (1024, 1536), (0, 2048)]:
        part = s.generate_ik_solutions(P[lo:hi].contiguous(), n=(1 if hi - lo == 1 else None), latent=L[lo:hi].contiguous())
        assert (part - full[lo:hi]).abs().max().item() <= FLOW_TOL
    # pin one form for every size: now bitwise - the 128x128 per-layer tiles (101), then the row-owner launch (182)
    for variant in (101, 182):
        s.engine(DEV).set_gemm_variant(variant)
        pinned = s.generate_ik_solutions(P, latent=L)
        assert (pinned - full).abs().max().item() <= FLOW_TOL
        for lo, hi in [(0, 1), (5, 133), (4000, 4096)]:
            part = s.generate_ik_solutions(P[lo:hi].contiguous(), n=(1 if hi - lo == 1 else None), latent=L[lo:hi].contiguous())
            assert torch.equal(part, pinned[lo:hi]), (variant, lo, hi)
    s.engine(DEV).set_gemm_variant(-1)
    s.engine(DEV).set_gemm_variant(181)
    # oracle on a slice
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses[:256], lat[:256])
    assert (full[:256].cpu() - ref).abs().max().item() <= FLOW_TOL
    lo_t = torch.tensor([l[0] for l in O(robot).actuated_joints_limits], device=DEV)
    hi_t = torch.tensor([l[1] for l in O(robot).actuated_joints_limits], device=DEV)
    assert bool(((full >= lo_t) & (full <= hi_t)).all())


@pytest.mark.gpu
@pytest.mark.parametrize("model_name,n", [("panda__full__lp191_5.25m", 4096), ("panda__full__lp191_5.25m", 512), ("fetch__large__ns183_9.75m", 8192)])
def test_round_trip_through_the_forward_pass_at_the_baseline_batch_sizes(model_name, n):
    """Size-independent property at full size: the flow is a bijection, so the graph run FORWARD in fp64 on the host
    (oracle/flow_oracle.py::flow_forward_f64, written independently of every inverse pass) must bring EVERY row of the HIP path's unclamped
    output back to its latent - 4096 rows of the Panda model through the one row-owner launch, 512 through the cluster form, 8192 rows of the
    16-block Fetch model (D = ndof = 8; FetchArm's 7 of 10 output columns cannot be inverted) through two rounds.  No second inverse pass is
    involved, so an error shared by the kernels and the oracle's inverse would show here; coupling coefficients of O(1) (last Linear x 2)."""
    robot, hp, lay, sd = released_model(model_name, seed=11, gain=2.0)
    assert lay.dim == lay.ndof
    s = _solver(robot, hp, sd)
    _, poses = reachable_poses(robot, n, 71)
    z = latents(n, lay.dim, 72)
    x = s.generate_ik_solutions(poses.to(DEV), latent=z.to(DEV), clamp_to_joint_limits=False).cpu().numpy()
    cond = torch.cat([poses, torch.zeros(n, 1)], 1).numpy()
    back = fo.flow_forward_f64(sd, lay, x, cond)
    err = np.abs(back - z.numpy()) / np.maximum(1.0, np.abs(z.numpy()))
    # the oracle's own fp32 inverse on a slice sets the scale: what an fp32 inverse pass leaves after an exact forward pass
    k = 256
    ref = fo.flow_inverse_torch(sd, lay, z[:k], torch.tensor(cond[:k])).numpy()
    ref_err = np.abs(fo.flow_forward_f64(sd, lay, ref, cond[:k]) - z[:k].numpy()) / np.maximum(1.0, np.abs(z[:k].numpy()))
    assert err.max() <= 2e-5 and err.max() <= 4.0 * max(ref_err.max(), 1e-6), (float(err.max()), float(ref_err.max()))
    assert np.median(err) <= 2.0 * max(np.median(ref_err), 1e-7), (float(np.median(err)), float(np.median(ref_err)))


@pytest.mark.parametrize("which,n", [("panda", 4096), ("fetch_arm", 8192)])
def test_every_row_of_the_baseline_batches_against_the_oracle(which, n):
    """BASELINE configs 2 and 4 (and the round-0 flow seeds of config 3: the same 4096 Panda rows through the same kernels) with
    EVERY row compared against the torch-CPU oracle - all 32 (64) row tiles of the headline launch, i.e. every workgroup on
    every XCD, not a slice - in both precisions, clamped (the API default, what bench.py runs) and unclamped (f32).
    Tolerance 1e-5 (north star), reported per 128-row tile."""
    robot, hp, lay, sd = (panda_model if which == "panda" else fetch_arm_model)()
    s = _solver(robot, hp, sd)
    _, poses = reachable_poses(robot, n, 0 if which == "panda" else 40)
    lat = latents(n, lay.dim, 1 if which == "panda" else 41)
    P, L = poses.to(DEV), lat.to(DEV)
    ref = {c: fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat, clamp=c) for c in (True, False)}
    assert ref[True].shape == (n, robot.ndof)
    worst = {}
    for prec, clamps in (("f32", (True, False)), ("f16x3", (True,))):
        s.set_precision(prec)
        assert s.engine(DEV).precision == prec
        for c in clamps:
            got = s.generate_ik_solutions(P, latent=L, clamp_to_joint_limits=c).cpu()
            # clamped outputs are joint angles inside the limits: absolute 1e-5.  Unclamped flow outputs reach |x| of 10 and more,
            # where 1e-5 absolute is a handful of fp32 ulps of the result itself: relative to max(1, |x|) there, as in the other
            # unclamped tests of this file
            scale = torch.ones_like(ref[c]) if c else torch.clamp(ref[c].abs(), min=1.0)
            err = ((got - ref[c]).abs() / scale).max(1).values  # per row
            tiles = err.reshape(n // 128, 128).max(1).values  # per 128-row tile of the contraction launch
            worst[(prec, c)] = (err.max().item(), int(tiles.argmax()))
            assert bool(torch.isfinite(got).all())
            assert (tiles <= FLOW_TOL).all(), f"{which} {prec} clamp={c}: tile {int(tiles.argmax())} is {tiles.max().item():.2e} from the oracle"
    print(f"{which} B={n}: max |hip - oracle| over ALL rows: " + ", ".join(f"{k[0]}{'' if k[1] else ' unclamped'} {v[0]:.2e} (tile {v[1]})" for k, v in worst.items()))
    assert s.engine(DEV).split_fallback_count == 0


def test_chunked_large_batch_equals_small_batches():
    robot, hp, lay, sd = tiny_model()
    s = _solver(robot, hp, sd)
    n = 40000  # > 2 chunks of 16384
    poses = reachable_poses(robot, n, 2)[1].to(DEV)
    lat = latents(n, lay.dim, 3).to(DEV)
    big = s.generate_ik_solutions(poses, latent=lat)
    spans = [(0, 100), (16300, 16500), (32768, 33000), (39990, 40000)]
    for lo, hi in spans:  # small batches run the k-split tile form: equal to rounding
        part = s.generate_ik_solutions(poses[lo:hi].contiguous(), latent=lat[lo:hi].contiguous())
        assert (part - big[lo:hi]).abs().max().item() <= FLOW_TOL
    s.engine(DEV).set_gemm_variant(101)  # same tile form for every batch size: bit-identical rows
    big = s.generate_ik_solutions(poses, latent=lat)
    for lo, hi in spans:
        part = s.generate_ik_solutions(poses[lo:hi].contiguous(), latent=lat[lo:hi].contiguous())
        assert torch.equal(part, big[lo:hi])


# ---- kinematics ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("which", ["panda", "fetch_arm", "fetch"])
def test_fk_matches_oracle(which):
    from ikflow_amd.robots import get_robot

    robot = get_robot(which)
    q = torch.tensor(O(robot).sample_joint_angles(2000, 0.0, np.random.default_rng(4)))
    ref64 = ko.forward_kinematics(robot, q.double())
    got = robot.forward_kinematics(q.to(DEV)).cpu()
    # quaternion sign: compare as rotations where the largest component is near a tie
    assert (got[:, :3] - ref64[:, :3].float()).abs().max().item() <= 2e-6
    dq = torch.minimum((got[:, 3:] - ref64[:, 3:].float()).abs().max(1).values, (got[:, 3:] + ref64[:, 3:].float()).abs().max(1).values)
    assert dq.max().item() <= 2e-6
    same_sign = ((got[:, 3:] - ref64[:, 3:].float()).abs().max(1).values <= 2e-6).float().mean().item()
    assert same_sign > 0.99


def test_fk_golden_vector_on_gpu():
    """tests/evaluation_utils_test.py:18-32 known answers, through the HIP path."""
    from ikflow_amd.robots import Panda

    robot = Panda()
    pose = robot.forward_kinematics(torch.zeros(1, 7, device=DEV)).cpu()[0]
    gt = torch.tensor([0.088, 0.0, 0.926, 0.0, 0.92387953, 0.38268343, 0.0])
    np.testing.assert_allclose(pose.numpy(), gt.numpy(), atol=1e-5)
    from ikflow_amd.engine import kinematics_engine_for

    eng = kinematics_engine_for(robot, DEV)
    pe, re = eng.pose_error(torch.zeros(1, 7, device=DEV), torch.tensor([[1.0, 1, 1, 1, 0, 0, 0]], device=DEV))
    assert abs(pe.item() - 1.355440887681938) < 1e-6
    assert abs(re.item() - 3.1415927) < 5e-4


def test_pose_error_jacobian_clamp_limits():
    from ikflow_amd.engine import kinematics_engine_for
    from ikflow_amd.robots import Panda

    robot = Panda()
    eng = kinematics_engine_for(robot, DEV)
    n = 1000
    q, poses = reachable_poses(robot, n, 8)
    q2 = torch.tensor(O(robot).sample_joint_angles(n, 0.0, np.random.default_rng(9)))
    pe, re = eng.pose_error(q2.to(DEV), poses.to(DEV))
    pe_ref, re_ref = ko.calculate_pose_error(robot, q2, poses)
    assert (pe.cpu() - pe_ref).abs().max().item() <= 2e-6
    assert (re.cpu() - re_ref).abs().max().item() <= 2e-5  # acos amplifies rounding near dot = +-1
    J = eng.jacobian(q2.to(DEV)).cpu()
    J_ref = ko.jacobian(robot, q2.double()).float()
    assert (J - J_ref).abs().max().item() <= 3e-6
    wild = 5.0 * torch.randn(n, 7, generator=torch.Generator().manual_seed(1))
    cl = robot.clamp_to_joint_limits(wild.to(DEV)).cpu()
    assert torch.equal(cl, ko.clamp_to_joint_limits(robot, wild))
    ex = eng.joint_limits_exceeded(wild.to(DEV)).cpu()
    assert torch.equal(ex, ko.calculate_joint_limits_exceeded(wild, O(robot).actuated_joints_limits))
    # vectors from the reference's own calculate_joint_limits_exceeded (tests/golden/make_ref_vectors.py), values exactly on a
    # limit and one float32 ulp beyond included: both HIP entry points
    import os

    from ikflow_amd import evaluation_utils as eu

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz"))
    cfg = torch.from_numpy(z["limits_cfg"]).to(DEV)
    assert np.array_equal(eng.joint_limits_exceeded(cfg).cpu().numpy(), z["limits_exceeded"])
    assert np.array_equal(eu.calculate_joint_limits_exceeded(cfg, robot.actuated_joints_limits).cpu().numpy(), z["limits_exceeded"])


def test_evaluation_utils_reference_known_answers_and_oracle():
    """ikflow/evaluation_utils.py mirror: the reference's own known answers (tests/evaluation_utils_test.py:14-57) and the
    oracle on random poses, torch and numpy inputs."""
    from ikflow_amd import evaluation_utils as eu
    from ikflow_amd.robots import Panda

    robot = Panda()
    # tests/evaluation_utils_test.py:18-32 - one target pose, the zero configuration
    target_pose = torch.tensor([1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0], dtype=torch.float32)
    solutions = torch.zeros((1, 7), dtype=torch.float32)
    l2, ang = eu.solution_pose_errors(robot, solutions, target_pose)
    assert l2.device.type == "cpu" and l2.shape == (1,)
    assert abs(l2[0].item() - 1.355440887681938) < 1e-6
    assert abs(ang[0].item() - 3.1415927) < 5e-4
    # tests/evaluation_utils_test.py:36-57 - 3-column limits table, strict inequalities
    configs = torch.tensor([[0, 0, 0], [0, 0, 0], [-2, 0, 0], [0, -1.999, 0], [0, 2.0001, 0]])
    returned = eu.calculate_joint_limits_exceeded(configs, [(-1, 1), (-2, 2), (-3, 3)])
    assert returned.dtype == torch.bool and returned.shape == (5,)
    assert torch.equal(returned, torch.tensor([False, False, True, False, True]))

    # pose_errors / pose_errors_cm_deg against the oracle, torch (device) and numpy inputs
    n = 777
    g = torch.Generator().manual_seed(3)
    p1 = torch.randn(n, 7, generator=g)
    p2 = torch.randn(n, 7, generator=g)
    p1[:, 3:] /= p1[:, 3:].norm(dim=1, keepdim=True)
    p2[:, 3:] /= p2[:, 3:].norm(dim=1, keepdim=True)
    p2[:5] = p1[:5]  # identical poses: the acos clamp decides the result
    l2_ref = torch.norm(p1[:, :3] - p2[:, :3], dim=1)
    ang_ref = ko.geodesic_distance_between_quaternions(p1[:, 3:], p2[:, 3:])
    l2, ang = eu.pose_errors(p1.to(DEV), p2.to(DEV))
    assert l2.is_cuda and (l2.cpu() - l2_ref).abs().max().item() <= 2e-6
    assert (ang.cpu() - ang_ref).abs().max().item() <= 2e-5
    l2n, angn = eu.pose_errors(p1.numpy(), p2.numpy())
    assert isinstance(l2n, np.ndarray) and np.array_equal(l2n, l2.cpu().numpy()) and np.array_equal(angn, ang.cpu().numpy())
    cm, deg = eu.pose_errors_cm_deg(p1, p2)
    assert torch.allclose(cm, 100 * l2.cpu()) and torch.allclose(deg, torch.rad2deg(ang.cpu()))
    ang_eps = eu.pose_errors(p1[:5].to(DEV), p2[:5].to(DEV), acos_epsilon=1e-3)[1].cpu()
    assert (ang_eps - ko.geodesic_distance_between_quaternions(p1[:5, 3:], p2[:5, 3:], acos_epsilon=1e-3)).abs().max().item() <= 2e-5
    with pytest.raises(AssertionError):
        eu.pose_errors(p1, p2[:10])

    # evaluate_solutions: batch and single-pose forms
    q, poses = reachable_poses(robot, 300, 4)
    wild = (q + 0.5 * torch.randn(300, 7, generator=g)).float()
    l2e, ange, lim, coll = eu.evaluate_solutions(robot, poses, wild)
    pe_ref, re_ref = ko.calculate_pose_error(robot, wild, poses)
    assert (l2e - pe_ref).abs().max().item() <= 2e-6 and (ange - re_ref).abs().max().item() <= 2e-5
    assert torch.equal(lim, ko.calculate_joint_limits_exceeded(wild, O(robot).actuated_joints_limits)) and coll is None
    l2s, _, _, _ = eu.evaluate_solutions(robot, poses[0], wild)
    assert (l2s - ko.calculate_pose_error(robot, wild, poses[0].repeat(300, 1))[0]).abs().max().item() <= 2e-6
    with pytest.raises(NotImplementedError):
        eu.calculate_self_collisions(robot, wild)


@pytest.mark.parametrize("which", ["panda", "fetch_arm"])
def test_lm_step_matches_fp64_twin(which):
    from ikflow_amd.robots import get_robot

    robot = get_robot(which)
    n = 3000
    g = torch.Generator().manual_seed(11)
    qt = torch.tensor(O(robot).sample_joint_angles(n, 0.01, np.random.default_rng(12)))
    poses = ko.forward_kinematics(robot, qt)
    q0 = ko.clamp_to_joint_limits(robot, qt + 0.15 * torch.randn(n, robot.ndof, generator=g))
    got = robot.inverse_kinematics_step_levenburg_marquardt(poses.to(DEV), q0.to(DEV)).cpu()
    ref64 = ko.lm_step(robot, poses.double(), q0.double())
    ref32 = ko.lm_step(robot, poses, q0)
    e64 = (got.double() - ref64).abs().max(1).values
    o64 = (ref32.double() - ref64).abs().max(1).values
    print(f"lm {which}: |hip-f64| max {e64.max():.3e}  |cpu32-f64| max {o64.max():.3e} median {o64.median():.3e}")
    assert e64.max().item() <= 5e-6
    # and therefore as close to the fp32 CPU path as that path is to exact arithmetic
    assert ((got - ref32).abs().max(1).values.double() <= o64 + 5e-6).all()


# ---- exact IK ----------------------------------------------------------------------------------------------------
def _exact_inputs(robot, lay, n, repeat_counts, seed):
    q_true, poses = reachable_poses(robot, n, seed)
    lats = [latents(n * r, lay.dim, 100 + i) for i, r in enumerate(repeat_counts)]
    return poses, lats


@pytest.mark.parametrize("n", [7, 200])
def test_exact_ik_matches_oracle_control_flow(n):
    """Whole retry schedule against the oracle's restatement of ikflow_solver.py:119-247,345-411, with injected latents.
    Random weights make the flow seeds poor, so loose thresholds (0.2 m / 1 rad) are used to get a mix of solved/unsolved poses and
    all three retry rounds; rows whose error sits within 1e-4 relative of a threshold may flip and are excluded."""
    robot, hp, lay, sd = tiny_model(seed=2)
    s = _solver(robot, hp, sd)
    rc = (1, 3, 10)
    pos_thr, rot_thr = 0.2, 1.0
    poses, lats = _exact_inputs(robot, lay, n, rc, 21)

    def flow_fn(latent, poses_tiled):
        return fo.generate_ik_solutions_torch(sd, lay, robot, poses_tiled, latent[: poses_tiled.shape[0]], clamp=True)

    ref_sol, ref_valid = ko.generate_exact_ik_solutions(robot, flow_fn, poses, lats, rc, pos_thr, rot_thr)
    # same schedule with each LM step evaluated in fp64 (what the kernel does): the comparator for solution VALUES
    ref_sol64, ref_valid64 = ko.generate_exact_ik_solutions(robot, flow_fn, poses, lats, rc, pos_thr, rot_thr, lm_dtype=torch.float64)
    # the engine does not clear its output buffers (round 0's selection writes every pose's slot): hand the allocator blocks of NaN / 0xff
    # of the outputs' sizes to recycle, so that a slot the selection skipped would show
    junk_q = torch.full((n, 7), float("nan"), device=DEV)
    junk_v = torch.full((n,), 255, dtype=torch.uint8, device=DEV)
    torch.cuda.synchronize()
    del junk_q, junk_v
    sol, valid = s.generate_exact_ik_solutions(
        poses.to(DEV), repeat_counts=rc, pos_error_threshold=pos_thr, rot_error_threshold=rot_thr,
        latents=[l.to(DEV) for l in lats],
    )
    sol, valid = sol.cpu(), valid.cpu()
    assert sol.shape == (n, 7) and valid.dtype == torch.bool and valid.shape == (n,)
    assert bool(torch.isfinite(sol).all())
    frac32 = (valid == ref_valid).float().mean().item()
    frac64 = (valid == ref_valid64).float().mean().item()
    print(f"exact n={n}: valid {int(valid.sum())}/{n} (oracle fp32-LM {int(ref_valid.sum())}, fp64-LM {int(ref_valid64.sum())}), "
          f"agreement {frac32:.4f} / {frac64:.4f}")
    assert 0 < int(ref_valid.sum()) < n
    assert frac32 >= 0.95 and frac64 >= 0.97
    both = (valid == ref_valid64) & valid
    d = (sol[both] - ref_sol64[both]).abs().max(1).values
    print(f"   solution |hip - oracle(fp64 LM)|: max {d.max().item():.2e}, median {d.median().item():.2e}, "
          f"frac <= 1e-3: {(d <= 1e-3).float().mean().item():.3f}")
    # three LM steps from a far seed amplify the ~1e-6 rounding difference of the flow seeds; a row can also differ
    # outright when a near-threshold repeat flips which one "wins"
    assert int((d > 1e-3).sum()) <= max(1, int(0.05 * d.numel()))
    assert torch.equal(sol[~valid], torch.zeros_like(sol[~valid]))  # unsolved rows stay 0 (ikflow_solver.py:197)
    # every reported-valid solution really meets the thresholds and the joint limits
    pe, re = ko.calculate_pose_error(robot, sol[valid], poses[valid])
    assert (pe < pos_thr * 1.001).all() and (re < rot_thr * 1.001).all()
    assert torch.equal(sol[valid], ko.clamp_to_joint_limits(robot, sol[valid]))


def test_exact_ik_converges_from_good_seeds_at_4096():
    """BASELINE config 3 size. With random weights the flow cannot seed LM well, so convergence is exercised through
    the LM kernels directly: seeds = q_true + N(0, 0.05^2) must reach 1 mm / 0.01 rad in 3 steps for most rows."""
    from ikflow_amd.engine import kinematics_engine_for
    from ikflow_amd.robots import Panda

    robot = Panda()
    eng = kinematics_engine_for(robot, DEV)
    n = 4096
    q_true, poses = reachable_poses(robot, n, 31)
    q = ko.clamp_to_joint_limits(robot, q_true + 0.05 * torch.randn(n, 7, generator=torch.Generator().manual_seed(32))).to(DEV)
    P = poses.to(DEV)
    for _ in range(3):
        q = eng.lm_step(P, q)
    pe, re = eng.pose_error(q, P)
    ok = ((pe < 1e-3) & (re < 0.01)).float().mean().item()
    print(f"LM from perturbed truth: {ok:.4f} converged")
    assert ok > 0.95


def test_exact_ik_api_quirks():
    robot, hp, lay, sd = tiny_model()
    s = IKFlowSolver(hp, robot)
    poses = reachable_poses(robot, 4, 0)[1].to(DEV)
    with pytest.raises(AssertionError):
        s.generate_exact_ik_solutions(poses)  # weights not loaded (ikflow_solver.py:362)
    s.load_state_dict_tensors(sd)
    with pytest.raises(AssertionError):
        s.generate_exact_ik_solutions(poses, repeat_counts=[1, 3])  # must be a tuple (:359)
    with pytest.raises(AssertionError):
        s.generate_exact_ik_solutions(poses, return_detailed=True)  # (:361)
    sol, valid = s.generate_exact_ik_solutions(poses)
    assert sol.device.type == "cuda" and sol.shape == (4, 7) and valid.shape == (4,)


def test_return_detailed_tuple():
    robot, hp, lay, sd = tiny_model()
    s = _solver(robot, hp, sd)
    n = 64
    _, poses = reachable_poses(robot, n, 3)
    lat = latents(n, lay.dim, 4)
    sol, pe, re, lim, coll, rt = s.generate_ik_solutions(poses.to(DEV), latent=lat.to(DEV), return_detailed=True)
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat)
    pe_ref, re_ref = ko.calculate_pose_error(robot, ref, poses)
    assert (pe.cpu() - pe_ref).abs().max().item() < 1e-4 and (re.cpu() - re_ref).abs().max().item() < 1e-3
    assert lim.dtype == torch.bool and not bool(lim.any()) and coll is None and isinstance(rt, float)


def test_exact_ik_retry_rounds_cross_the_flow_chunk_boundary():
    """Tight thresholds leave (almost) every pose unsolved, so round 3 runs 10 x ~1800 = ~18000 flow rows: more than one
    16384-row chunk, with the conditional gathered through pose_idx[row % n_active] across the chunk boundary."""
    robot, hp, lay, sd = tiny_model(seed=2)
    s = _solver(robot, hp, sd)
    n, rc = 1800, (1, 3, 10)
    poses, lats = _exact_inputs(robot, lay, n, rc, 33)

    def flow_fn(latent, poses_tiled):
        return fo.generate_ik_solutions_torch(sd, lay, robot, poses_tiled, latent[: poses_tiled.shape[0]], clamp=True)

    ref_sol, ref_valid = ko.generate_exact_ik_solutions(robot, flow_fn, poses, lats, rc, 0.02, 0.1, lm_dtype=torch.float64)
    eng = s.engine(DEV)
    sol, valid, stats = eng.generate_exact(poses.to(DEV), rc, 0.02, 0.1, latents=[l.to(DEV) for l in lats], return_stats=True)
    sol, valid = sol.cpu(), valid.cpu()
    assert stats[2, 1] > 16384, stats  # third round really spans two chunks
    assert stats[0, 0] == n and stats[1, 0] == n - stats[0, 3] and stats[2, 0] == stats[1, 0] - stats[1, 3]
    assert int(valid.sum()) == int(stats[:, 3].sum())
    agree = (valid == ref_valid).float().mean().item()
    print(f"exact n={n}: valid {int(valid.sum())} (oracle {int(ref_valid.sum())}), agreement {agree:.4f}, stats {stats.tolist()}")
    assert agree >= 0.97
    both = (valid == ref_valid) & valid
    d = (sol[both] - ref_sol[both]).abs().max(1).values
    assert int((d > 1e-3).sum()) <= max(1, int(0.05 * d.numel()))
    assert torch.equal(sol[~valid], torch.zeros_like(sol[~valid]))


def test_cabi_status_codes_and_edge_sizes():
    """C-ABI error behaviour (include/ikflow_amd.h): distinct status codes + ikf_last_error text; n = 0 is a no-op."""
    import ctypes as C

    from ikflow_amd import _lib
    from ikflow_amd.engine import Engine, EngineError

    robot, hp, lay, sd = tiny_model()
    eng = Engine(lay, robot, DEV)
    lib = eng.lib
    q = torch.zeros(4, 7, device=DEV)
    out = torch.zeros(4, 7, device=DEV)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # weights not loaded -> IKF_ERR_NOT_LOADED (the reference's assert message)
    lat = torch.zeros(4, lay.dim, device=DEV)
    code = lib.ikf_generate_approx(eng._h, q.data_ptr(), 0, lat.data_ptr(), 4, 1, 0.0, out.data_ptr(), stream)
    assert code == _lib.IKF_ERR_NOT_LOADED and "Model weights have not been loaded" in _lib.last_error()
    # kinematics work without weights; null pointer -> IKF_ERR_NULL_POINTER; n = 0 -> OK
    assert lib.ikf_forward_kinematics(eng._h, q.data_ptr(), 4, out.data_ptr(), stream) == _lib.IKF_OK
    assert lib.ikf_forward_kinematics(eng._h, None, 4, out.data_ptr(), stream) == _lib.IKF_ERR_NULL_POINTER
    assert lib.ikf_forward_kinematics(eng._h, None, 0, None, stream) == _lib.IKF_OK
    assert lib.ikf_forward_kinematics(eng._h, q.data_ptr(), -1, out.data_ptr(), stream) == _lib.IKF_ERR_BAD_ARGUMENT
    # bad state_dict -> IKF_ERR_MISSING_TENSOR surfaced as RuntimeError by the shim
    bad = dict(sd)
    bad.pop("module_list.0.M_inv")
    with pytest.raises(RuntimeError, match="Missing key"):
        eng.load_state_dict(bad)
    eng.load_state_dict(sd)
    assert eng.weights_loaded
    assert lib.ikf_generate_approx(eng._h, q.data_ptr(), 0, lat.data_ptr(), 0, 1, 0.0, out.data_ptr(), stream) == _lib.IKF_OK
    rc = (C.c_int32 * 1)(0)
    cb = _lib.LATENT_FN(lambda *a: 0)
    valid = torch.zeros(4, dtype=torch.uint8, device=DEV)
    code = lib.ikf_generate_exact(eng._h, q.data_ptr(), 4, rc, 1, 3, 1e-3, 0.1, cb, None, out.data_ptr(), valid.data_ptr(), None, stream)
    assert code == _lib.IKF_ERR_BAD_ARGUMENT  # repeat count 0
    with pytest.raises(EngineError):
        eng.set_gemm_variant(77)
    # the model-free evaluation helpers
    pe = torch.zeros(4, device=DEV)
    lo3, hi3 = (C.c_float * 3)(-1, -2, -3), (C.c_float * 3)(1, 2, 3)
    ex = torch.zeros(4, dtype=torch.uint8, device=DEV)
    assert lib.ikf_pose_distance(q.data_ptr(), q.data_ptr(), 4, -1.0, pe.data_ptr(), pe.data_ptr(), stream) == _lib.IKF_OK
    assert lib.ikf_pose_distance(None, q.data_ptr(), 4, -1.0, pe.data_ptr(), pe.data_ptr(), stream) == _lib.IKF_ERR_NULL_POINTER
    assert lib.ikf_pose_distance(None, None, 0, -1.0, None, None, stream) == _lib.IKF_OK
    assert lib.ikf_limits_exceeded(q.data_ptr(), 4, 3, C.cast(lo3, C.c_void_p), C.cast(hi3, C.c_void_p), ex.data_ptr(), stream) == _lib.IKF_OK
    assert lib.ikf_limits_exceeded(q.data_ptr(), 4, 33, C.cast(lo3, C.c_void_p), C.cast(hi3, C.c_void_p), ex.data_ptr(), stream) == _lib.IKF_ERR_BAD_ARGUMENT
    assert lib.ikf_limits_exceeded(q.data_ptr(), 4, 3, None, C.cast(hi3, C.c_void_p), ex.data_ptr(), stream) == _lib.IKF_ERR_NULL_POINTER
    # empty exact call through the shim
    s = _solver(robot, hp, sd)
    sol, v = s.generate_exact_ik_solutions(torch.zeros(0, 7, device=DEV))
    assert sol.shape == (0, 7) and v.shape == (0,)
    # a descriptor the kernels are not built for is refused at create time
    from ikflow_amd.model import FlowLayout

    with pytest.raises(EngineError, match="coeff_fn_internal_size"):
        Engine(FlowLayout(nb_nodes=2, dim=9, dim_cond=8, width=5000, n_hidden=2, clamp=2.5, ndof=7), robot, DEV)
    with pytest.raises(EngineError, match="n_layers"):
        Engine(FlowLayout(nb_nodes=2, dim=9, dim_cond=8, width=256, n_hidden=5, clamp=2.5, ndof=7), robot, DEV)


def test_hip_path_reproduces_committed_golden_fixtures():
    """tests/golden/*.npz (made by tests/golden/make_golden.py): BASELINE config 1 inputs (the 3 README poses, batch 16)
    through the HIP path, both contraction precisions; FK / pose error / LM step fixtures through the kinematics kernels."""
    import os

    from ikflow_amd.engine import kinematics_engine_for
    from ikflow_amd.robots import Panda

    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name, model in (("tiny_flow.npz", tiny_model), ("panda_flow.npz", panda_model)):
        z = np.load(os.path.join(gold, name))
        robot, hp, lay, sd = model(seed=int(z["weights_seed"]))
        s = _solver(robot, hp, sd)
        P, L = torch.from_numpy(z["poses"]).to(DEV), torch.from_numpy(z["latent"]).to(DEV)
        for prec in ("f32", "f16x3"):
            s.set_precision(prec)
            got = s.generate_ik_solutions(P, latent=L).cpu().numpy()
            assert np.abs(got - z["q_clamped"]).max() <= FLOW_TOL, (name, prec)
            got_nc = s.generate_ik_solutions(P, latent=L, clamp_to_joint_limits=False).cpu().numpy()
            assert (np.abs(got_nc - z["q_unclamped"]) / np.maximum(1.0, np.abs(z["q_unclamped"]))).max() <= FLOW_TOL, (name, prec)
        s.set_precision("f32")
        if "q_single_pose0" in z.files:  # single-pose form: y = README pose 0, n = 16
            got1 = s.generate_ik_solutions(P[0].contiguous(), n=16, latent=L).cpu().numpy()
            assert np.abs(got1 - z["q_single_pose0"]).max() <= FLOW_TOL
    z = np.load(os.path.join(gold, "panda_kinematics.npz"))
    robot = Panda()
    eng = kinematics_engine_for(robot, DEV)
    fk = eng.forward_kinematics(torch.from_numpy(z["q"]).to(DEV)).cpu().numpy()
    assert np.abs(fk[:, :3] - z["fk"][:, :3]).max() <= 2e-6
    assert np.minimum(np.abs(fk[:, 3:] - z["fk"][:, 3:]).max(1), np.abs(fk[:, 3:] + z["fk"][:, 3:]).max(1)).max() <= 2e-6
    pe, re = eng.pose_error(torch.from_numpy(z["q0"]).to(DEV), torch.from_numpy(z["target"]).to(DEV))
    assert np.abs(pe.cpu().numpy() - z["pos_err"]).max() <= 2e-6 and np.abs(re.cpu().numpy() - z["rot_err"]).max() <= 3e-5
    lm = eng.lm_step(torch.from_numpy(z["target"]).to(DEV), torch.from_numpy(z["q0"]).to(DEV)).cpu().numpy()
    assert np.abs(lm - z["lm_step_f64"]).max() <= 5e-6
    J = eng.jacobian(torch.from_numpy(z["q"]).to(DEV)).cpu().numpy()
    assert np.abs(J - z["jac_f64"]).max() <= 3e-6


def test_baseline_sizes_size_independent_properties():
    """BASELINE configs 3 and 4 at full size through properties that need no oracle run."""
    # config 4: FetchArm, B = 8192 approximate: deterministic, finite, inside limits, oracle on a slice
    robot, hp, lay, sd = fetch_arm_model()
    s = _solver(robot, hp, sd)
    n = 8192
    _, poses = reachable_poses(robot, n, 40)
    lat = latents(n, lay.dim, 41)
    P, L = poses.to(DEV), lat.to(DEV)
    a = s.generate_ik_solutions(P, latent=L)
    assert torch.equal(a, s.generate_ik_solutions(P, latent=L)) and bool(torch.isfinite(a).all())
    lo = torch.tensor([l[0] for l in O(robot).actuated_joints_limits], device=DEV)
    hi = torch.tensor([l[1] for l in O(robot).actuated_joints_limits], device=DEV)
    assert bool(((a >= lo) & (a <= hi)).all())
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses[:128], lat[:128])
    assert (a[:128].cpu() - ref).abs().max().item() <= FLOW_TOL
    # config 3: Panda exact IK, B = 4096, 1 mm / 0.01 rad: every row reported valid meets the thresholds (checked with the
    # oracle's FK) and the limits; every other row is exactly 0; valid is a bool vector on the input device
    robot, hp, lay, sd = panda_model()
    s = _solver(robot, hp, sd)
    _, poses = reachable_poses(robot, 4096, 42)
    sol, valid = s.generate_exact_ik_solutions(poses.to(DEV), pos_error_threshold=1e-3, rot_error_threshold=0.01)
    assert sol.shape == (4096, 7) and valid.shape == (4096,) and valid.dtype == torch.bool and sol.device.type == "cuda"
    sol, valid = sol.cpu(), valid.cpu()
    assert torch.equal(sol[~valid], torch.zeros_like(sol[~valid]))
    if int(valid.sum()) > 0:
        pe, re = ko.calculate_pose_error(robot, sol[valid], poses[valid])
        assert (pe < 1e-3 * 1.01).all() and (re < 0.01 * 1.01).all()
        assert torch.equal(sol[valid], ko.clamp_to_joint_limits(robot, sol[valid]))


def test_one_million_poses_in_one_call():
    """BASELINE config 5 on one GPU: 1,000,000 target poses through the engine's 16384-row chunking.  Size-independent
    properties: finite, inside the limits, deterministic; rows around chunk boundaries and at the ends agree with the
    oracle and with the same rows submitted on their own."""
    robot, hp, lay, sd = panda_model()
    s = _solver(robot, hp, sd)
    n = 1_000_000
    g = torch.Generator(device=DEV).manual_seed(5)
    lo = torch.tensor([l[0] for l in O(robot).actuated_joints_limits], device=DEV)
    hi = torch.tensor([l[1] for l in O(robot).actuated_joints_limits], device=DEV)
    q = lo + (hi - lo) * torch.rand((n, 7), generator=g, device=DEV)
    poses = robot.forward_kinematics(q)
    lat = torch.randn((n, lay.dim), generator=g, device=DEV)
    sol = s.generate_ik_solutions(poses, latent=lat)
    assert sol.shape == (n, 7) and bool(torch.isfinite(sol).all())
    assert bool(((sol >= lo) & (sol <= hi)).all())
    assert torch.equal(sol, s.generate_ik_solutions(poses, latent=lat))
    for start in (0, 16384 - 20, 16384 * 31 - 7, n - 40):
        sl = slice(start, start + 40)
        ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses[sl].cpu(), lat[sl].cpu())
        assert (sol[sl].cpu() - ref).abs().max().item() <= FLOW_TOL, f"rows {start}.."
        alone = s.generate_ik_solutions(poses[sl].contiguous(), latent=lat[sl].contiguous())
        assert (sol[sl] - alone).abs().max().item() <= FLOW_TOL


@pytest.mark.parametrize("model_name", ["panda_lite_tpm", "fetch_full_temp_nsc_tpm", "fetch__large__ns183_9.75m"])
def test_every_released_architecture_matches_oracle(model_name):
    """The released architectures not covered above (model_descriptions.yaml: 6-block Panda, 12- and 16-block Fetch with
    D = 8 and the prismatic torso joint): flow and exact-IK entry points against the oracle, seeded weights."""
    from helpers import released_model

    robot, hp, lay, sd = released_model(model_name, seed=4)
    s = _solver(robot, hp, sd)
    for n, precision in ((40, "f32"), (600, "f32"), (600, "f16x3")):
        s.set_precision(precision)
        _, poses = reachable_poses(robot, n, 50)
        lat = latents(n, lay.dim, 51)
        ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat)
        got = s.generate_ik_solutions(poses.to(DEV), latent=lat.to(DEV)).cpu()
        assert got.shape == (n, robot.ndof)
        assert (got - ref).abs().max().item() <= FLOW_TOL, f"{model_name} n={n} {precision}"
    s.set_precision("f32")
    sol, valid = s.generate_exact_ik_solutions(poses[:64].to(DEV))
    assert sol.shape == (64, robot.ndof) and valid.shape == (64,) and valid.dtype == torch.bool
    assert torch.equal(sol[~valid], torch.zeros_like(sol[~valid]))


@pytest.mark.parametrize("which", ["panda", "fetch"])
def test_capsule_self_collision_matches_oracle(which):
    """ikf_self_collision (mechanism of evaluation_utils.calculate_self_collisions) against the float64 oracle: capsules on
    the base, on links behind actuated joints and behind fixed joints (folded on the host), random configurations."""
    from ikflow_amd import evaluation_utils as eu
    from ikflow_amd.robots import get_robot

    robot = get_robot(which)
    names = [j.name for j in robot.joints]
    act = [j.name for j in robot.joints if j.actuated]
    rng = np.random.default_rng(11)
    capsules = [(None, (0.0, 0.0, 0.0), (0.0, 0.0, 0.25), 0.07)]
    for nm in (act[1], act[3], act[5], names[-1]):  # names[-1] is a fixed joint (hand / gripper)
        p0 = tuple(rng.uniform(-0.08, 0.08, 3))
        p1 = tuple(rng.uniform(-0.15, 0.15, 3))
        capsules.append((nm, p0, p1, float(rng.uniform(0.03, 0.08))))
    capsules.append((act[5], (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 0.05))  # a sphere (degenerate segment)
    ignored = [(0, 1)]
    robot.set_collision_capsules(capsules, ignored_pairs=ignored)
    n = 3000
    q = torch.tensor(O(robot).sample_joint_angles(n, 0.0, rng))
    ref = ko.capsule_clearance(robot, capsules, ignored, q)
    dist = robot.self_collision_distances(q.to(DEV)).cpu().double()
    assert (dist - ref).abs().max().item() <= 2e-5, (dist - ref).abs().max().item()
    col = robot.config_self_collides(q.to(DEV)).cpu()
    clear = ref.abs() > 1e-4  # flags must agree wherever the clearance is not within rounding of zero
    assert torch.equal(col[clear], (ref < 0)[clear])
    assert 0 < int(col.sum()) < n  # the random model produces both outcomes
    assert robot.config_self_collides(q[0].to(DEV)) == bool(col[0])
    # evaluate_solutions fills the fourth slot once a collision model is attached
    poses = robot.forward_kinematics(q[:50].to(DEV))
    out = eu.evaluate_solutions(robot, poses, q[:50].to(DEV))
    assert out[3] is not None and torch.equal(out[3].cpu(), col[:50])
    if which == "panda":  # the solver's return_detailed tuple carries the flags too
        _, hp, lay, sd = tiny_model()
        sv = IKFlowSolver(hp, robot)
        sv.load_state_dict_tensors(sd)
        det = sv.generate_ik_solutions(poses, latent=latents(50, lay.dim, 3).to(DEV), return_detailed=True)
        assert det[4] is not None and det[4].dtype == torch.bool and det[4].shape == (50,)
        assert torch.equal(det[4], robot.config_self_collides(det[0]))
    # C-ABI errors: no model set -> BAD_ARGUMENT; bad frame / pair -> BAD_ARGUMENT
    from ikflow_amd import _lib
    from ikflow_amd.engine import Engine, EngineError
    from ikflow_amd.model import FlowLayout

    eng = Engine(FlowLayout(nb_nodes=1, dim=max(robot.ndof, 2), dim_cond=8, width=256, n_hidden=1, clamp=2.5, ndof=robot.ndof), robot, DEV)
    with pytest.raises(EngineError, match="no collision model"):
        eng.self_collision(q[:4].to(DEV))
    with pytest.raises(EngineError, match="frame"):
        eng.set_collision_model([(robot.ndof + 1, (0, 0, 0), (0, 0, 1), 0.1)], [])
    with pytest.raises(EngineError, match="pair"):
        eng.set_collision_model([(0, (0, 0, 0), (0, 0, 1), 0.1)], [(0, 0)])


# ---- round 2: deterministic exact-IK parity (seeds in), subnet depths / widths, the unfused pipeline, guards -----------
def _seed_tables(robot, q_true, rc, seed):
    """Seeds for every (round, repeat, pose): the truth perturbed by a per-pose noise level, clamped.  The levels are
    spread so that some poses converge in round 0, some in a later round, and some never (sigma = 3 rad)."""
    g = torch.Generator().manual_seed(seed)
    n, ndof = q_true.shape
    sigma = torch.tensor([0.01, 0.05, 0.2, 0.5, 1.0, 3.0])[torch.randint(0, 6, (n,), generator=g)]
    tables = []
    for R in rc:
        noise = torch.randn(R, n, ndof, generator=g) * sigma[None, :, None]
        tables.append(ko.clamp_to_joint_limits(robot, (q_true[None] + noise).reshape(R * n, ndof)).reshape(R, n, ndof))
    return tables


@pytest.mark.parametrize("which,pos_thr,rot_thr", [("panda", 1e-3, 0.01), ("fetch", 1e-3, 0.1), ("fetch_arm", 5e-4, 0.05)])
def test_exact_ik_seeded_is_row_exact_against_the_oracle(which, pos_thr, rot_thr):
    """BASELINE config 3 size (n = 4096, repeat_counts (1, 3, 10)) with IDENTICAL seeds on both sides
    (ikf_generate_exact_seeded): everything behind the flow - LM iterations, validity, highest-valid-repeat-wins, slot
    order, ordered compaction, retry rounds (ikflow_solver.py:190-247, 383-408) - compared pose by pose.

    Required: identical `valid` flags and |dq| <= 5e-6 (vs the oracle with its LM step in fp64, what the kernel does) on
    every pose outside the threshold band.  Band = poses for which some evaluated (round, iteration, repeat) had an error
    within rounding of its threshold: the two sides evaluate FK in fp32 with different operation orders (|d pos| ~ 2e-7;
    |d dot| ~ 2e-7 and rot = 2 acos(dot) => |d rot| ~ 2 |d dot| / rot), so
        band_pos = 1e-4 * thr + 5e-7,    band_rot = 1e-4 * thr + 6e-7 / thr."""
    from ikflow_amd.engine import kinematics_engine_for
    from ikflow_amd.robots import get_robot

    robot = get_robot(which)
    n, rc = 4096, (1, 3, 10)
    q_true, poses = reachable_poses(robot, n, 61)
    tables = _seed_tables(robot, q_true, rc, 62)

    def seed_cpu(rnd, idx):
        return tables[rnd][:, idx, :].reshape(-1, robot.ndof).contiguous()

    ref_sol, ref_valid, margins = ko.generate_exact_ik_solutions_seeded(
        robot, seed_cpu, poses, rc, pos_thr, rot_thr, lm_dtype=torch.float64, return_margins=True)
    eng = kinematics_engine_for(robot, DEV)  # needs no flow weights
    dev_tables = [t.to(DEV) for t in tables]

    def seed_dev(rnd, idx, repeat):
        assert repeat == rc[rnd]
        return dev_tables[rnd][:, idx, :].reshape(-1, robot.ndof).contiguous()

    sol, valid, stats = eng.generate_exact(poses.to(DEV), rc, pos_thr, rot_thr, seed_fn=seed_dev, return_stats=True)
    sol, valid = sol.cpu(), valid.cpu()
    band = (margins[:, 0] <= 1e-4 * pos_thr + 5e-7) | (margins[:, 1] <= 1e-4 * rot_thr + 6e-7 / rot_thr)
    clear = ~band
    n_ref = [int(ref_valid.sum())]
    print(f"{which}: valid {int(valid.sum())}/{n} (oracle {n_ref[0]}), band {int(band.sum())}, stats {stats.tolist()}")
    assert int(band.sum()) <= 0.03 * n
    assert 0.3 * n < int(ref_valid.sum()) < n and stats[1, 0] > 0 and stats[2, 0] > 0  # all three rounds ran
    assert torch.equal(valid[clear], ref_valid[clear])
    both = clear & valid & ref_valid
    d = (sol[both] - ref_sol[both]).abs().max(1).values
    print(f"   |hip - oracle| on {int(both.sum())} both-valid poses: max {d.max().item():.2e}")
    # 5e-6 on every pose, plus - pose by pose - what the pose's OWN conditioning allows: the oracle is run a second time with every iterate that
    # leaves an LM step moved by one fp32 ulp; |twin - oracle| on a pose is what a last-bit difference in an intermediate q does to ITS result
    # (the step map amplifies along the arm's self-motion by up to |e| |d2x/dq2| / lambda).  The two sides differ by such last-bit amounts (the
    # kernel's chain constants are fp32, the oracle's fp64), so a pose may sit as far from the oracle as SENS_FACTOR of its own twins do.
    twin_sol, twin_valid = ko.generate_exact_ik_solutions_seeded(robot, seed_cpu, poses, rc, pos_thr, rot_thr, lm_dtype=torch.float64, q_ulps=1)
    sens = (twin_sol - ref_sol).abs().max(1).values[both]
    sens[~twin_valid[both]] = float("inf")   # (a pose whose flag a last-bit nudge flips is a band pose in all but name)
    SENS_FACTOR = 8.0
    allowed = 5e-6 + SENS_FACTOR * sens
    print(f"   poses beyond 5e-6: {int((d > 5e-6).sum())}; max d / allowed {float((d / allowed).max()):.2f}; twin sensitivity max {float(sens[torch.isfinite(sens)].max()):.2e}")
    assert bool((d <= allowed).all()), (float(d.max()), float((d / allowed).max()), int((d > 5e-6).sum()))
    assert torch.equal(sol[~valid], torch.zeros_like(sol[~valid]))
    assert int(stats[0, 0]) == n and int(valid.sum()) == int(stats[:, 3].sum())
    # per-round bookkeeping agrees with the oracle's up to the band poses
    assert abs(int(stats[:, 3].sum()) - n_ref[0]) <= int(band.sum())
    # ... and how far that sits from the REFERENCE-precision loop (ikflow_solver.py:199-211: jrl solves the step in fp32; the kernel
    # and the oracle above solve it in fp64): flags, and the distribution of |q_hip - q_cpu32| on poses valid on both sides, next to
    # the fp32 loop's own distance from the fp64 loop.  fp32 normal equations at cond(J^T J) ~ 1e5 make the fp32 loop the noisy one:
    # the HIP result must be no further from it than it is from its own fp64 twin.
    ref32_sol, ref32_valid = ko.generate_exact_ik_solutions_seeded(robot, seed_cpu, poses, rc, pos_thr, rot_thr)
    agree = float((valid == ref32_valid).float().mean())
    agree_twins = float((ref_valid == ref32_valid).float().mean())

    def dist(a, b, m):
        d_ = (a[m] - b[m]).abs().max(1).values
        return float(d_.median()), float(d_.quantile(0.99)), float(d_.max())

    d32 = dist(sol, ref32_sol, valid & ref32_valid)
    o32 = dist(ref_sol, ref32_sol, ref_valid & ref32_valid)
    print(f"   vs the fp32-LM loop: flags agree {agree:.4f} (fp64 twin vs fp32 loop {agree_twins:.4f}); |q_hip - q_cpu32| median {d32[0]:.2e} "
          f"p99 {d32[1]:.2e} max {d32[2]:.2e}; |q_f64 - q_cpu32| median {o32[0]:.2e} p99 {o32[1]:.2e} max {o32[2]:.2e}")
    assert agree >= 0.95 and agree >= agree_twins - 0.005
    assert d32[0] <= 1.5 * o32[0] + 1e-6 and d32[1] <= 1.5 * o32[1] + 1e-5


def test_exact_ik_in_the_reference_lm_arithmetic():
    """BASELINE config 3 in the reference's OWN arithmetic (ikf_set_lm_precision 0): fp32 chain walk, Jacobian, J^T J + 1e-4 I
    and an LU / partial-pivoting solve - what jrl's step does on fp32 tensors (ikflow/config.py:8, ikflow_solver.py:199-211, torch.linalg.solve
    = LAPACK sgesv).  n = 4096, repeat_counts (1, 3, 10), identical seeds, compared pose by pose with the oracle's fp32 loop.

    What can be asked of two fp32 evaluations of this step: J^T J of a 7-joint arm in a 6-D task has rank 6, so cond(J^T J + 1e-4 I) =
    sigma_max^2 / 1e-4 + 1 >= 1e4 on EVERY pose (measured 3.5e4 +- 10 %, profiles/r06_lm_precision.json) - each side carries cond x 2^-24 of its
    step as rounding noise (the fp32 reference itself sits 2.6e-5 median / 3.5e-4 max from the fp64 evaluation of ONE step).  So:
      * every pose with cond < 1e3 must agree to 1e-5 (there are none on these arms: the count is printed and asserted on, not assumed);
      * on the rest the HIP fp32 loop must be statistically the reference loop: as close to the oracle's fp32 loop as the oracle's fp64 loop is
        (median and p99 within 1.5x), flags agreeing at least as often."""
    from ikflow_amd.engine import kinematics_engine_for
    from ikflow_amd.robots import get_robot

    robot = get_robot("panda")
    n, rc, pos_thr, rot_thr = 4096, (1, 3, 10), 1e-3, 0.01
    q_true, poses = reachable_poses(robot, n, 61)
    tables = _seed_tables(robot, q_true, rc, 62)

    def seed_cpu(rnd, idx):
        return tables[rnd][:, idx, :].reshape(-1, robot.ndof).contiguous()

    ref32_sol, ref32_valid = ko.generate_exact_ik_solutions_seeded(robot, seed_cpu, poses, rc, pos_thr, rot_thr, lm_dtype=torch.float32)
    ref64_sol, ref64_valid = ko.generate_exact_ik_solutions_seeded(robot, seed_cpu, poses, rc, pos_thr, rot_thr, lm_dtype=torch.float64)
    eng = kinematics_engine_for(robot, DEV)
    assert eng.lm_precision == "f64"
    eng.set_lm_precision("f32")
    assert eng.lm_precision == "f32"
    dev_tables = [t.to(DEV) for t in tables]

    def seed_dev(rnd, idx, repeat):
        return dev_tables[rnd][:, idx, :].reshape(-1, robot.ndof).contiguous()

    sol, valid, stats = eng.generate_exact(poses.to(DEV), rc, pos_thr, rot_thr, seed_fn=seed_dev, return_stats=True)
    again, valid_again = eng.generate_exact(poses.to(DEV), rc, pos_thr, rot_thr, seed_fn=seed_dev)
    assert torch.equal(sol, again) and torch.equal(valid, valid_again)   # deterministic
    sol, valid = sol.cpu(), valid.cpu()
    # every returned solution meets the thresholds (fp32 FK, as the reference checks it) and unsolved rows are zero
    pe, re = ko.calculate_pose_error(robot, sol[valid], poses[valid])
    assert float(pe.max()) < pos_thr + 5e-7 and float(re.max()) < rot_thr + 6e-7 / rot_thr
    assert torch.equal(sol[~valid], torch.zeros_like(sol[~valid]))
    both = valid & ref32_valid
    d = (sol[both] - ref32_sol[both]).abs().max(1).values
    J = ko.jacobian(robot, ref32_sol[both].double())
    cond = torch.linalg.cond(J.transpose(1, 2) @ J + 1e-4 * torch.eye(robot.ndof, dtype=torch.float64))
    well = cond < 1e3
    print(f"fp32 LM mode: valid {int(valid.sum())}/{n} (oracle fp32 loop {int(ref32_valid.sum())}, fp64 loop {int(ref64_valid.sum())}); cond min {float(cond.min()):.3g} "
          f"median {float(cond.median()):.3g} max {float(cond.max()):.3g}; poses with cond < 1e3: {int(well.sum())}")
    if bool(well.any()):
        assert float(d[well].max()) <= 1e-5
    assert float(cond.min()) >= 1e3, "a rank-6 J^T J + 1e-4 I cannot be better conditioned than sigma_max^2 / 1e-4"

    def dist(a, b, m):
        d_ = (a[m] - b[m]).abs().max(1).values
        return float(d_.median()), float(d_.quantile(0.99)), float(d_.max())

    h32 = dist(sol, ref32_sol, both)
    o64 = dist(ref64_sol, ref32_sol, ref64_valid & ref32_valid)
    agree, agree_twins = float((valid == ref32_valid).float().mean()), float((ref64_valid == ref32_valid).float().mean())
    for lo, hi in ((1e3, 1e4), (1e4, 2e4), (2e4, 3e4), (3e4, 4e4), (4e4, 1e5), (1e5, float("inf"))):
        mk = (cond >= lo) & (cond < hi)
        if bool(mk.any()):
            print(f"   cond [{lo:.0e}, {hi:.0e}): {int(mk.sum())} poses, |q_hip32 - q_cpu32| median {float(d[mk].median()):.2e} p99 {float(d[mk].quantile(0.99)):.2e} max {float(d[mk].max()):.2e}")
    print(f"   |q_hip32 - q_cpu32| median {h32[0]:.2e} p99 {h32[1]:.2e} max {h32[2]:.2e}; |q_cpu64 - q_cpu32| median {o64[0]:.2e} p99 {o64[1]:.2e} max {o64[2]:.2e}; "
          f"flags agree {agree:.4f} (fp64 loop vs fp32 loop {agree_twins:.4f})")
    assert h32[0] <= 1.5 * o64[0] + 1e-6 and h32[1] <= 1.5 * o64[1] + 1e-5
    assert agree >= agree_twins - 0.005
    eng.set_lm_precision("f64")


def test_lm_step_in_the_reference_arithmetic_carries_the_reference_noise():
    """One LM step on 4096 perturbed seeds, three evaluations: the oracle in fp32 (the reference's arithmetic), the kernel in fp32 mode, and fp64
    truth.  In units of cond x 2^-24 x |dq| - the rounding noise of ANY fp32 evaluation of (J^T J + 1e-4 I)^-1 J^T e - the kernel's fp32 step is
    as close to truth as the oracle's, and the two agree with one another to that noise; the default fp64-inside step is ~300x closer."""
    from ikflow_amd.engine import kinematics_engine_for
    from ikflow_amd.robots import get_robot

    robot = get_robot("panda")
    n = 4096
    q_true, poses = reachable_poses(robot, n, 0)
    seeds = ko.clamp_to_joint_limits(robot, q_true + 0.05 * torch.randn(q_true.shape, generator=torch.Generator().manual_seed(3)))
    J = ko.jacobian(robot, seeds.double())
    cond = torch.linalg.cond(J.transpose(1, 2) @ J + 1e-4 * torch.eye(robot.ndof, dtype=torch.float64))
    ref32 = ko.lm_step(robot, poses, seeds)
    ref64 = ko.lm_step(robot, poses.double(), seeds.double())
    unit = cond * 2.0 ** -24 * (ref64 - seeds.double()).abs().max(1).values.clamp_min(1e-3)
    eng = kinematics_engine_for(robot, DEV)
    out = {}
    for mode in ("f32", "f64"):
        eng.set_lm_precision(mode)
        got = eng.lm_step(poses.to(DEV), seeds.to(DEV)).cpu().double()
        out[mode] = ((got - ref64).abs().max(1).values / unit, (got - ref32.double()).abs().max(1).values / unit)
    eng.set_lm_precision("f64")
    o32 = (ref32.double() - ref64).abs().max(1).values / unit
    q = lambda t: (float(t.median()), float(t.quantile(0.99)), float(t.max()))
    print(f"LM step, errors in units of cond eps |dq| (median, p99, max): oracle fp32 vs truth {q(o32)}; hip fp32 vs truth {q(out['f32'][0])}, vs oracle fp32 "
          f"{q(out['f32'][1])}; hip fp64 vs truth {q(out['f64'][0])}")
    assert float(cond.min()) >= 1e3
    assert q(out["f32"][0])[0] <= 1.5 * q(o32)[0] and q(out["f32"][0])[1] <= 1.5 * q(o32)[1]     # as close to truth as the reference's arithmetic
    assert q(out["f32"][1])[1] <= 2.5 * q(o32)[1]                                               # and to the reference itself, to that noise
    # no pose far outside the noise model - the poses next to a singularity (two small eigenvalues) included: with fused multiply-adds in the LU
    # elimination the kernel's worst pose sat at 9.5 units against the oracle's 1.5 (r06; solve_lu_pivot is compiled without contraction since)
    assert float(out["f32"][0].max()) <= 4.0 and float(o32.max()) <= 4.0
    assert q(out["f64"][0])[2] <= 0.05                                                          # fp64 inside: two orders below it


def _random_exact_configs(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        rc = tuple(int(v) for v in rng.integers(1, 13, size=int(rng.integers(1, 5))))
        out.append(dict(which=str(rng.choice(["panda", "fetch", "fetch_arm"])), n=int(rng.choice([1, 2, 50, 333, 1000, 2500])), rc=rc,
                        pos_thr=float(rng.choice([5e-4, 1e-3, 5e-3])), rot_thr=float(rng.choice([0.01, 0.05, 0.1])), seed=int(rng.integers(0, 1000))))
    return out


@pytest.mark.parametrize("cfg", _random_exact_configs(int(os.environ.get("IKF_FUZZ_EXACT_COUNT", "16")), int(os.environ.get("IKF_FUZZ_SEED", "20260928"))),
                         ids=lambda c: "-".join(str(v) for v in c.values()).replace(" ", ""))
def test_exact_ik_seeded_random_schedules(cfg):
    """The seeded exact-IK comparison over random robots, pose counts, repeat_counts schedules (1..4 rounds of 1..12 repeats; the
    reference accepts any tuple, ikflow_solver.py:351) and thresholds - same band rule as the n = 4096 test above."""
    from ikflow_amd.engine import kinematics_engine_for
    from ikflow_amd.robots import get_robot

    robot = get_robot(cfg["which"])
    n, rc, pos_thr, rot_thr = cfg["n"], cfg["rc"], cfg["pos_thr"], cfg["rot_thr"]
    q_true, poses = reachable_poses(robot, n, cfg["seed"])
    tables = _seed_tables(robot, q_true, rc, cfg["seed"] + 1)
    ref_sol, ref_valid, margins = ko.generate_exact_ik_solutions_seeded(
        robot, lambda rnd, idx: tables[rnd][:, idx, :].reshape(-1, robot.ndof).contiguous(), poses, rc, pos_thr, rot_thr,
        lm_dtype=torch.float64, return_margins=True)
    eng = kinematics_engine_for(robot, DEV)
    dev_tables = [t.to(DEV) for t in tables]
    sol, valid, stats = eng.generate_exact(poses.to(DEV), rc, pos_thr, rot_thr, return_stats=True,
                                           seed_fn=lambda rnd, idx, repeat: dev_tables[rnd][:, idx, :].reshape(-1, robot.ndof).contiguous())
    sol, valid = sol.cpu(), valid.cpu()
    clear = ~((margins[:, 0] <= 1e-4 * pos_thr + 5e-7) | (margins[:, 1] <= 1e-4 * rot_thr + 6e-7 / rot_thr))
    assert int((~clear).sum()) <= max(2, 0.03 * n)
    assert torch.equal(valid[clear], ref_valid[clear])
    both = clear & valid & ref_valid
    if bool(both.any()):
        # Kernel and oracle both evaluate an LM step in fp64 and round q to fp32 after it.  The two fp64 results differ by ~1e-15 (LU against
        # Cholesky, another FK association), so now and then (about 1e-4 of the roundings) they fall on different sides of an fp32 rounding
        # boundary - one ulp, 2.4e-7 - and the NEXT step multiplies that by its sensitivity, which near a singular configuration and far from
        # the target (these schedules accept up to 5 mm / 0.1 rad) reaches tens.  Neither side is wrong; the reference's own fp32 loop sits a
        # hundred times further out.  So: beyond 5e-6 only stragglers (seed 31337, 120 schedules: 1 element at 6.4e-6 and 2
        # at <= 9.6e-6 out of 13.5 k), never beyond 1e-4.
        d = (sol[both] - ref_sol[both]).abs()
        assert d.max().item() <= 1e-4 and int((d > 5e-6).sum()) <= max(2, 5e-4 * d.numel()), (d.max().item(), int((d > 5e-6).sum()), d.numel())
    assert torch.equal(sol[~valid], torch.zeros_like(sol[~valid]))
    assert int(stats[0, 0]) == n and int(valid.sum()) == int(stats[: len(rc), 3].sum())


def test_refine_exact_one_round_and_large_compaction():
    """ikf_refine_exact = one _generate_exact_ik_solutions call (:119-247) given its flow output; then the multi-workgroup
    ordered compaction (n > 32768) through a 100k-pose seeded call whose second round must see exactly the unsolved poses
    in ascending order."""
    from ikflow_amd.engine import kinematics_engine_for
    from ikflow_amd.robots import Panda

    robot = Panda()
    eng = kinematics_engine_for(robot, DEV)
    n, R = 777, 3
    q_true, poses = reachable_poses(robot, n, 71)
    seeds = _seed_tables(robot, q_true, (R,), 72)[0].reshape(R * n, 7)
    ref_sol, ref_valid = ko.exact_round(robot, seeds, poses, R, 1e-3, 0.05, lm_dtype=torch.float64)
    sol, valid = eng.refine_exact(poses.to(DEV), seeds.to(DEV), R, 1e-3, 0.05)
    agree = (valid.cpu() == ref_valid).float().mean().item()
    assert agree >= 0.99, agree
    both = valid.cpu() & ref_valid
    d1 = (sol.cpu()[both] - ref_sol[both]).abs()
    assert d1.max().item() <= 5e-6 or (int((d1 > 5e-6).sum()) <= 2 and d1.max().item() <= 1e-4), d1.max().item()
    # 100k poses: round 0 solves the even poses (seed = truth), round 1 must be handed exactly the odd ones, in order
    n = 100_000
    eng.reserve_exact(n, 2)
    q_true, poses = reachable_poses(robot, n, 73)
    far = ko.clamp_to_joint_limits(robot, q_true + 2.5)
    seen = {}

    def seed_dev(rnd, idx, repeat):
        seen[rnd] = idx.clone()
        base = q_true.to(DEV)[idx] if rnd == 1 else torch.where((idx % 2 == 0)[:, None], q_true.to(DEV)[idx], far.to(DEV)[idx])
        return base.repeat((repeat, 1))

    sol, valid, stats = eng.generate_exact(poses.to(DEV), (1, 2), 1e-3, 0.01, seed_fn=seed_dev, return_stats=True)
    solved0 = int(stats[0, 3])
    assert solved0 >= n // 2 and stats[1, 0] == n - solved0
    assert bool((seen[1] % 2 == 1).all()) and seen[1].numel() == n - solved0  # only odd poses are left after round 0
    assert bool((seen[1][1:] > seen[1][:-1]).all())  # ascending = ordered compaction
    assert int(valid.sum()) >= 0.99 * n


@pytest.mark.parametrize("kw", [
    dict(nb_nodes=2, dim=7, n_hidden=1, width=256),
    dict(nb_nodes=2, dim=9, n_hidden=1, width=768),
    dict(nb_nodes=2, dim=7, n_hidden=4, width=256),
    dict(nb_nodes=2, dim=8, n_hidden=4, width=512, robot_name="fetch"),
    dict(nb_nodes=2, dim=10, n_hidden=3, width=768, robot_name="fetch_arm"),
    dict(nb_nodes=2, dim=7, n_hidden=2, width=512),
    dict(nb_nodes=2, dim=7, n_hidden=3, width=300),    # not a multiple of 256: zero-padded to 512, exact
    dict(nb_nodes=2, dim=7, n_hidden=2, width=64),
    dict(nb_nodes=1, dim=7, n_hidden=2, width=1280),   # wider than any released model
])
def test_flow_every_subnet_depth_and_width(kw):
    """coeff_fn_config 1..4 and coeff_fn_internal_size other than 256 / 1024 (ikflow/model.py:51-96 accepts any): HIP vs the
    oracle at row counts on both sides of the tile-picker boundaries, both precisions where the depth allows."""
    robot, hp, lay, sd = custom_model(seed=9, gain=1.5, **kw)
    s = _solver(robot, hp, sd)
    n_max = 1400
    _, poses = reachable_poses(robot, n_max, 81)
    lat = latents(n_max, lay.dim, 82)
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat, clamp=False)
    scale = torch.clamp(ref.abs(), min=1.0)
    for prec in (("f32", "f16x3") if lay.n_hidden >= 2 else ("f32",)):
        s.set_precision(prec)
        for n in (1, 37, 300, 700, 1400):
            got = s.generate_ik_solutions(poses[:n].to(DEV), n=(1 if n == 1 else None), latent=lat[:n].to(DEV), clamp_to_joint_limits=False).cpu()
            err = ((got - ref[:n]).abs() / scale[:n]).max().item()
            assert err <= FLOW_TOL, f"{kw} {prec} n={n}: {err:.2e}"
    s.set_precision("f32")
    clamped = s.generate_ik_solutions(poses[:64].to(DEV), latent=lat[:64].to(DEV)).cpu()
    assert (clamped - fo.generate_ik_solutions_torch(sd, lay, robot, poses[:64], lat[:64])).abs().max().item() <= FLOW_TOL


def _random_flow_configs(count, seed):
    rng = np.random.default_rng(seed)
    widths = [1, 16, 100, 255, 256, 257, 300, 512, 640, 768, 1000, 1024, 1100, 1280, 1536, 2048]
    out = []
    for _ in range(count):
        robot_name = str(rng.choice(["panda", "fetch", "fetch_arm"]))
        ndof = O(robot_name).ndof
        out.append(dict(nb_nodes=int(rng.integers(1, 5)), dim=int(rng.integers(ndof, 17)), n_hidden=int(rng.integers(1, 5)),
                        width=int(rng.choice(widths)), robot_name=robot_name, softflow=bool(rng.integers(0, 2)),
                        sigmoid=bool(rng.integers(0, 4) == 0), seed=int(rng.integers(0, 1000)), gain=float(rng.choice([1.0, 1.5, 2.5])),
                        n=int(rng.choice([1, 2, 31, 33, 65, 100, 128, 129, 200, 256, 257, 513, 600, 1025, 1300])), clamp=bool(rng.integers(0, 2))))
    return out


_FUZZ_STATS = {"runs": 0, "noise_branch": []}


@pytest.mark.parametrize("cfg", _random_flow_configs(int(os.environ.get("IKF_FUZZ_COUNT", "48")), int(os.environ.get("IKF_FUZZ_SEED", "20260928"))), ids=lambda c: "-".join(str(v) for v in c.values()))
def test_flow_random_configurations(cfg):
    """Seeded random draws over everything IkflowModelParameters / glow_cNF_model (ikflow/model.py:17-41,300-354) can express within
    the boundary's limits - robot, nb_nodes, dim_latent_space up to 16, coeff_fn_config 1..4, any coeff_fn_internal_size, softflow on /
    off, sigmoid_on_output - at row counts around the tile boundaries: HIP vs the oracle, both precisions where the depth allows."""
    cfg = dict(cfg)
    n, clamp = cfg.pop("n"), cfg.pop("clamp")
    robot, hp, lay, sd = custom_model(**cfg)
    if cfg["sigmoid"] and cfg["softflow"]:   # ikflow_solver.py:43-44 refuses the combination; so does the drop-in
        with pytest.raises(AssertionError):
            _solver(robot, hp, sd)
        return
    s = _solver(robot, hp, sd)
    _, poses = reachable_poses(robot, n, cfg["seed"] + 1)
    lat = latents(n, lay.dim, cfg["seed"] + 2)
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat, clamp=clamp)
    cond = (torch.cat([poses, torch.zeros(n, 1)], 1) if cfg["softflow"] else poses).numpy()
    ref64 = torch.from_numpy(fo.run_inference_f64(sd, lay, robot, lat.numpy(), cond, clamp)).float()
    scale = torch.clamp(ref64.abs(), min=1.0)
    # random weights with output gain 2.5 can be ill-conditioned in fp32 (exp of large s): where the fp32 CPU path itself is further
    # than 1e-5 from the fp64 evaluation, the HIP path must be no further from fp64 than 4x what the fp32 CPU path is
    cpu_noise = ((ref - ref64).abs() / scale).max().item()
    for prec in (("f32", "f16x3") if lay.n_hidden >= 2 else ("f32",)):
        s.set_precision(prec)
        got = s.generate_ik_solutions(poses.to(DEV), n=(1 if n == 1 else None), latent=lat.to(DEV), clamp_to_joint_limits=clamp).cpu()
        assert got.shape == ref.shape
        err = ((got - ref).abs() / scale).max().item()
        err64 = ((got - ref64).abs() / scale).max().item()
        assert err <= FLOW_TOL or err64 <= 4 * cpu_noise, f"{cfg} {prec} n={n}: {err:.2e} vs fp32, {err64:.2e} vs fp64 (cpu {cpu_noise:.2e})"
        _FUZZ_STATS["runs"] += 1
        if err > FLOW_TOL:  # passed on the ill-conditioned-case branch only: counted, and bounded by the test below
            _FUZZ_STATS["noise_branch"].append(f"{'-'.join(str(v) for v in cfg.values())} {prec} n={n}: {err:.2e} vs fp32 cpu, {err64:.2e} vs fp64, cpu noise {cpu_noise:.2e}")


def _random_resident_configs(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        robot_name = str(rng.choice(["panda", "fetch", "fetch_arm"]))
        ndof = O(robot_name).ndof
        out.append(dict(nb_nodes=int(rng.integers(1, 4)), dim=int(rng.integers(ndof, 15)), n_hidden=3, width=int(rng.choice([1024, 1024, 1000, 900])),
                        robot_name=robot_name, softflow=bool(rng.integers(0, 2)), sigmoid=bool(rng.integers(0, 4) == 0), seed=int(rng.integers(0, 1000)),
                        gain=float(rng.choice([1.0, 1.5, 2.0])),
                        n=int(rng.choice([7, 8, 15, 16, 17, 120, 129, 255, 300, 511, 513, 777, 1024, 1500, 2047, 2049, 2600, 3327, 3500, 4096, 4111, 4500, 8200])),
                        clamp=bool(rng.integers(0, 2)), single_pose=bool(rng.integers(0, 5) == 0), soft=float(rng.choice([0.0, 0.0, 0.3]))))
    return out


@pytest.mark.parametrize("cfg", _random_resident_configs(int(os.environ.get("IKF_FUZZ_RESIDENT_COUNT", "32")), int(os.environ.get("IKF_FUZZ_SEED", "20260928"))),
                         ids=lambda c: "-".join(str(v) for v in c.values()))
def test_resident_row_forms_random_configurations(cfg):
    """Seeded random draws over what the row-owner launch and the cluster form accept (width padded to 1024, coeff_fn_config 3; any robot, depth,
    dim_latent_space up to 14 - i.e. up to 7 x inputs per subnet -, softflow on / off with a non-zero entry, sigmoid_on_output, the single-pose
    form) at row counts on both sides of every plan boundary (8 / 128 / 256 / 512 / 1024 / 2048 / ~3300 / 4096 rows): the default plan
    (whatever mix of forms it is) against the oracle, every row."""
    cfg = dict(cfg)
    n, clamp, single, soft = cfg.pop("n"), cfg.pop("clamp"), cfg.pop("single_pose"), cfg.pop("soft")
    if cfg["sigmoid"] and cfg["softflow"]:
        cfg["softflow"] = False   # (the reference refuses the combination, tested in test_flow_random_configurations)
    robot, hp, lay, sd = custom_model(**cfg)
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    _, poses = reachable_poses(robot, n, cfg["seed"] + 1)
    if single:
        poses = poses[:1].expand(n, 7).contiguous()
    lat = latents(n, lay.dim, cfg["seed"] + 2)
    soft = soft if lay.dim_cond == 8 else 0.0
    cond = torch.cat([poses, torch.full((n, 1), soft)], 1) if lay.dim_cond == 8 else poses
    ref = fo.run_inference_torch(sd, lay, robot, lat, cond, clamp)
    ref64 = torch.from_numpy(fo.run_inference_f64(sd, lay, robot, lat.numpy(), cond.numpy(), clamp)).float()
    scale = torch.clamp(ref64.abs(), min=1.0)
    cpu_noise = ((ref - ref64).abs() / scale).max().item()
    got = eng.generate_approx((poses[0] if single else poses).to(DEV), lat.to(DEV), clamp, softflow_scale=soft).cpu()
    assert got.shape == ref.shape and bool(torch.isfinite(got).all())
    err = ((got - ref).abs() / scale).max().item()
    err64 = ((got - ref64).abs() / scale).max().item()
    plan = eng.plan(n)
    assert plan != f"perlayer:{n}", plan   # (these shapes are the resident-row forms' domain)
    assert err <= FLOW_TOL or err64 <= 4 * cpu_noise, f"{cfg} n={n} plan {plan}: {err:.2e} vs fp32, {err64:.2e} vs fp64 (cpu {cpu_noise:.2e})"


def test_flow_random_configurations_share_of_ill_conditioned_cases():
    """The fuzz cases above may pass on `err64 <= 4 * cpu_noise` when the fp32 CPU path itself is further than 1e-5 from fp64 (gain-2.5
    random weights).  That branch must stay the exception: report every case that took it, fail above 10 % of the runs."""
    runs, noisy = _FUZZ_STATS["runs"], _FUZZ_STATS["noise_branch"]
    print(f"fuzz: {len(noisy)} of {runs} (configuration, precision) runs left the 1e-5 contract against the fp32 CPU path and passed on the fp64 bound")
    for line in noisy:
        print("   ", line)
    if runs == 0:
        pytest.skip("the fuzz cases did not run in this session")
    assert len(noisy) <= 0.10 * runs, noisy


@pytest.mark.parametrize("n_hidden", [1, 2, 3])
def test_unfused_pipeline_every_gemm_variant(n_hidden):
    """The four-kernel pipeline of csrc/flow_kernels.hip (k_first_layer / k_gemm_lrelu[_p3] / k_last_layer_coupling): taken
    automatically for coeff_fn_config = 1 and on request (ikf_set_gemm_variant 0..8) - every contraction variant against
    the oracle, partial tiles included."""
    robot, hp, lay, sd = custom_model(nb_nodes=2, dim=9, n_hidden=n_hidden, width=256, seed=3, gain=1.5)
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    n_max = 700
    _, poses = reachable_poses(robot, n_max, 91)
    lat = latents(n_max, lay.dim, 92)
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat)
    for variant in range(9):
        eng.set_gemm_variant(variant)
        for n in (1, 100, 129, 700):
            got = s.generate_ik_solutions(poses[:n].to(DEV), n=(1 if n == 1 else None), latent=lat[:n].to(DEV)).cpu()
            assert (got - ref[:n]).abs().max().item() <= FLOW_TOL, f"variant {variant} n={n}"
    eng.set_gemm_variant(-1)
    got = s.generate_ik_solutions(poses.to(DEV), latent=lat.to(DEV)).cpu()
    assert (got - ref).abs().max().item() <= FLOW_TOL


def test_f16x3_range_guard():
    """f16 holds |a| <= 65504.  With the guard on (default) a call whose hidden activations leave that range is re-run on
    the f32 path (result = the f32 path's, event counted); with it off the flag is readable; weights out of range refuse
    the mode.  Overflow is provoked in the entry kernel (first Linear) and in a contraction epilogue (second Linear), with
    the next layer scaled back so the f32 path stays well conditioned."""
    for hot_layer in (0, 1):
        robot, hp, lay, sd = custom_model(nb_nodes=2, dim=7, n_hidden=3, width=256, seed=4)
        g = lay.glow_module(1)
        big = 4.0e5
        sd = dict(sd)
        sd[f"module_list.{g}.subnet2.{2 * hot_layer}.weight"] = sd[f"module_list.{g}.subnet2.{2 * hot_layer}.weight"] * big
        sd[f"module_list.{g}.subnet2.{2 * hot_layer}.bias"] = sd[f"module_list.{g}.subnet2.{2 * hot_layer}.bias"] * big
        sd[f"module_list.{g}.subnet2.{2 * hot_layer + 2}.weight"] = sd[f"module_list.{g}.subnet2.{2 * hot_layer + 2}.weight"] / big
        n = 900
        _, poses = reachable_poses(robot, n, 5)
        lat = latents(n, lay.dim, 6)
        P, L = poses.to(DEV), lat.to(DEV)
        ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat)
        s = _solver(robot, hp, sd)
        f32 = s.generate_ik_solutions(P, latent=L)
        assert (f32.cpu() - ref).abs().max().item() <= 5 * FLOW_TOL  # 4e5-scaled layer: looser absolute tolerance
        s.set_precision("f16x3")
        eng = s.engine(DEV)
        assert eng.split_fallback_count == 0
        got = s.generate_ik_solutions(P, latent=L)
        assert eng.split_fallback_count == 1, f"hot layer {hot_layer}: overflow not detected"
        assert torch.equal(got, f32)
        sol, valid = s.generate_exact_ik_solutions(P[:50])  # the exact path re-runs its flow rounds too
        assert eng.split_fallback_count >= 2 and bool(torch.isfinite(sol).all())
        eng.set_split_guard(False)
        unguarded = s.generate_ik_solutions(P, latent=L)
        assert eng.split_overflow_pending() and not eng.split_overflow_pending()  # read-and-clear
        assert eng.split_fallback_count >= 2 and unguarded.shape == got.shape
        eng.set_split_guard(True)
    # in-range weights: no flag, no fallback
    robot, hp, lay, sd = tiny_model()
    s = _solver(robot, hp, sd)
    s.set_precision("f16x3")
    _, poses = reachable_poses(robot, 600, 7)
    s.generate_ik_solutions(poses.to(DEV), latent=latents(600, lay.dim, 8).to(DEV))
    assert s.engine(DEV).split_fallback_count == 0 and not s.engine(DEV).split_overflow_pending()
    # a weight beyond the f16 range: the mode is refused, the handle stays on f32
    from ikflow_amd.engine import EngineError

    sd2 = dict(sd)
    k = f"module_list.{lay.glow_module(0)}.subnet1.2.weight"
    sd2[k] = sd2[k].copy()
    sd2[k][3, 5] = 1.0e5
    s2 = _solver(robot, hp, sd2)
    s2.engine(DEV)
    with pytest.raises(EngineError, match="f16 range"):
        s2.set_precision("f16x3")
    assert s2.engine(DEV).precision == "f32"


def test_reload_with_out_of_range_weight_in_f16x3_mode_keeps_every_batch_size_on_the_new_weights():
    """A handle already in f16x3 mode is given a state dict with a hidden weight beyond the f16 range: the split images are
    refused and the handle falls back to f32 - and the f32 path, at EVERY batch size (the <= 512-row kernels read the
    fragment-major weight image, larger batches the row-major one), must run the NEW weights.  Also: an activation-overflow
    bit left in the shared range word by an unguarded call must not make a later set_precision refuse valid weights."""
    from ikflow_amd.engine import EngineError

    robot, hp, lay, sd = tiny_model()
    k = f"module_list.{lay.glow_module(0)}.subnet1.2.weight"
    sd_bad = dict(sd)
    sd_bad[k] = sd_bad[k].copy()
    sd_bad[k][3, 5] = 1.0e5
    k_next = f"module_list.{lay.glow_module(0)}.subnet1.4.weight"  # the next Linear scales that unit back: the f32 path stays well conditioned
    sd_bad[k_next] = sd_bad[k_next].copy()
    sd_bad[k_next][:, 3] *= 1.0e-5
    _, poses = reachable_poses(robot, 700, 31)
    lat = latents(700, lay.dim, 32)
    want_new = fo.generate_ik_solutions_torch(sd_bad, lay, robot, poses, lat)
    want_old = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat)
    assert (want_new - want_old).abs().max().item() > 1e-3  # the two weight sets are distinguishable
    for first_load_in_split_mode in (False, True):
        s = IKFlowSolver(hp, robot)
        if first_load_in_split_mode:  # the very first load of a handle that is already in f16x3 mode
            s.load_state_dict_tensors(sd_bad)
            s._precision = "f16x3"
            with pytest.raises(EngineError, match="f16 range"):
                s.engine(DEV)
            assert s._precision == "f32"  # the solver follows the engine's refusal
        else:
            s.load_state_dict_tensors(sd)
            s.set_precision("f16x3")
            s.generate_ik_solutions(poses[:64].to(DEV), latent=lat[:64].to(DEV))
            with pytest.raises(EngineError, match="f16 range"):
                s.load_state_dict_tensors(sd_bad)
            assert s._precision == "f32"
        eng = s.engine(DEV)
        assert eng.precision == "f32" and eng.weights_loaded
        for n in (5, 200, 256, 300, 512, 700):  # 32x32 tiles, 32x64 tiles, large tiles
            got = eng.generate_approx(poses[:n].to(DEV), lat[:n].to(DEV), True).cpu()
            err_new, err_old = (got - want_new[:n]).abs().max().item(), (got - want_old[:n]).abs().max().item()
            assert err_new <= FLOW_TOL, (first_load_in_split_mode, n, err_new, err_old)
    # a pending activation-overflow bit does not poison the weight check of a later mode switch
    robot, hp, lay, sd = custom_model(nb_nodes=2, dim=7, n_hidden=3, width=256, seed=4)
    g = lay.glow_module(1)
    hot = dict(sd)
    hot[f"module_list.{g}.subnet2.0.weight"] = hot[f"module_list.{g}.subnet2.0.weight"] * 4.0e5
    hot[f"module_list.{g}.subnet2.0.bias"] = hot[f"module_list.{g}.subnet2.0.bias"] * 4.0e5
    hot[f"module_list.{g}.subnet2.2.weight"] = hot[f"module_list.{g}.subnet2.2.weight"] / 4.0e5
    s = _solver(robot, hp, hot)
    s.set_precision("f16x3")
    eng = s.engine(DEV)
    eng.set_split_guard(False)
    _, p2 = reachable_poses(robot, 300, 33)
    s.generate_ik_solutions(p2.to(DEV), latent=latents(300, lay.dim, 34).to(DEV))  # leaves the overflow bit set
    s.load_state_dict_tensors(sd)  # in-range weights, mode f16x3: must be accepted
    assert eng.precision == "f16x3"
    assert eng.split_overflow_pending()  # ... and the pending bit still belongs to its reader
    eng.set_split_guard(True)
    got = s.generate_ik_solutions(p2.to(DEV), latent=latents(300, lay.dim, 34).to(DEV)).cpu()
    assert (got - fo.generate_ik_solutions_torch(sd, lay, robot, p2, latents(300, lay.dim, 34))).abs().max().item() <= FLOW_TOL
    assert eng.split_fallback_count == 0


def test_exact_ik_row_state_grows_per_round_and_error_exits_keep_the_stream_contract():
    """(1) Without ikf_reserve_exact the row state of a call is what its rounds need; a schedule whose worst case is beyond the
    up-front bound starts with round 0's rows and grows between rounds - results equal those of a reserved handle.
    (2) A call that fails after work was enqueued (a latent callback that gives up in round 1) still records the handle's
    tail event: a following call on ANOTHER stream waits for the first call's kernels (shared scratch) and is correct."""
    robot, hp, lay, sd = tiny_model()
    n = 3000
    _, poses = reachable_poses(robot, n, 41)
    P = poses.to(DEV)
    rc = (1, 2, 4)
    lats = [latents(n * r, lay.dim, 50 + i).to(DEV) for i, r in enumerate(rc)]
    a = _solver(robot, hp, sd)
    a.engine(DEV).reserve_exact(n, 4)
    sol_a, val_a = a.engine(DEV).generate_exact(P, rc, 1e-3, 0.01, latents=lats)
    b = _solver(robot, hp, sd)  # nothing reserved, and no up-front worst case: the row state grows between the rounds
    b.engine(DEV).set_exact_upfront_rows(0)
    sol_b, val_b = b.engine(DEV).generate_exact(P, rc, 1e-3, 0.01, latents=lats)
    assert torch.equal(val_a, val_b) and torch.equal(sol_a, sol_b)
    assert 0 < int(val_a.sum()) < n  # rounds 1 and 2 really ran

    class GiveUp(Exception):
        pass

    eng = b.engine(DEV)
    short = [lats[0], lats[1][:5]]  # round 1's latent is too short: the callback raises after round 0 was enqueued
    s1, s2 = torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)
    with torch.cuda.stream(s1):
        with pytest.raises(AssertionError, match="must be at least"):
            eng.generate_exact(P, rc[:2], 1e-3, 0.01, latents=short)
    with torch.cuda.stream(s2):  # no host synchronisation in between
        got = b.generate_ik_solutions(P, latent=lats[0])
    torch.cuda.synchronize()
    want = a.generate_ik_solutions(P, latent=lats[0])
    assert torch.equal(got, want)


def test_one_handle_called_from_two_streams():
    """The per-handle scratch is shared: a call arriving on another stream waits for the previous call's work
    (hipStreamWaitEvent).  Alternating streams without any host synchronisation must give the single-stream results."""
    robot, hp, lay, sd = tiny_model()
    s = _solver(robot, hp, sd)
    n = 3000
    _, poses = reachable_poses(robot, n, 15)
    P = poses.to(DEV)
    lats = [latents(n, lay.dim, 20 + i).to(DEV) for i in range(6)]
    want = [s.generate_ik_solutions(P, latent=l) for l in lats]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)]
    got = []
    for i, l in enumerate(lats):
        with torch.cuda.stream(streams[i % 2]):
            got.append(s.generate_ik_solutions(P, latent=l))
    torch.cuda.synchronize()
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    assert torch.cuda.current_device() == 0



def test_panda_approximate_capsule_model():
    """Robot.use_approximate_collision_model(): this repository's own capsule approximation of the Panda (not jrl's geometry,
    no parity claim - SURVEY 8 f-3 stays unpinned).  Sanity on the GPU: natural postures are free, a fully folded arm
    collides, the kernel agrees with the float64 oracle on the same capsules, and uniformly random configurations collide
    at a plausible rate (the reference quotes 3-6 % self-colliding solutions for its released models, yaml:3-4)."""
    from ikflow_amd.robots import PANDA_APPROX_CAPSULES, PANDA_APPROX_IGNORED, Panda

    robot = Panda().use_approximate_collision_model()
    named = torch.tensor([[0, -np.pi / 4, 0, -3 * np.pi / 4, 0, np.pi / 2, np.pi / 4], [0, 0, 0, -1.5708, 0, 1.8675, 0],
                          [0.5, 0.3, -0.4, -2.0, 0.2, 2.2, 1.0], [0, -1.76, 0, -3.07, 0, 0.0, 0]], dtype=torch.float32)
    col = robot.config_self_collides(named.to(DEV)).cpu()
    assert col.tolist() == [False, False, False, True]
    q = torch.tensor(O(robot).sample_joint_angles(5000, 0.0, np.random.default_rng(3)))
    dist = robot.self_collision_distances(q.to(DEV)).cpu().double()
    frac = float((dist < 0).float().mean())
    assert 0.01 < frac < 0.15, frac
    # the oracle walks the same capsule list with its own chain and closest-point search; it tests every pair on different
    # frames except the ignored ones, so hand it the adjacent-frame pairs as ignored too
    folded, pairs = robot._collision_model
    all_pairs = {(a, b) for a in range(len(folded)) for b in range(a + 1, len(folded)) if folded[a][0] != folded[b][0]}
    ignored = sorted(all_pairs - set(pairs))
    ref = ko.capsule_clearance(robot, PANDA_APPROX_CAPSULES, ignored, q[:1500])
    assert (dist[:1500] - ref).abs().max().item() <= 2e-5
    _, hp, lay, sd = tiny_model()
    s = IKFlowSolver(hp, robot)
    s.load_state_dict_tensors(sd)
    det = s.generate_ik_solutions(robot.forward_kinematics(q[:40].to(DEV)), latent=latents(40, lay.dim, 3).to(DEV), return_detailed=True)
    assert det[4] is not None and det[4].dtype == torch.bool and det[4].shape == (40,)
    assert not Panda().has_collision_model  # opt-in: the default robot still answers None in the self-collision slot


def test_softflow_scale_and_zero_row_calls():
    """The 8th conditional entry (softflow scale; always 0.0 at inference in the reference, ikflow_solver.py:335-338) is a
    C-ABI argument: a non-zero value must equal the oracle run on the full 8-entry conditional.  Plus the n = 0 forms of
    every entry point."""
    robot, hp, lay, sd = tiny_model(seed=6, gain=1.5)
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    n = 300
    _, poses = reachable_poses(robot, n, 17)
    lat = latents(n, lay.dim, 18)
    for scale in (0.0, 0.37, -1.2):
        cond = torch.cat([poses, torch.full((n, 1), scale)], dim=1)
        ref = fo.run_inference_torch(sd, lay, robot, lat, cond, True)
        got = eng.generate_approx(poses.to(DEV), lat.to(DEV), True, softflow_scale=scale).cpu()
        assert (got - ref).abs().max().item() <= FLOW_TOL, scale
    eng.set_gemm_variant(4)  # the unfused pipeline reads the softflow column too
    cond = torch.cat([poses, torch.full((n, 1), 0.37)], dim=1)
    got = eng.generate_approx(poses.to(DEV), lat.to(DEV), True, softflow_scale=0.37).cpu()
    assert (got - fo.run_inference_torch(sd, lay, robot, lat, cond, True)).abs().max().item() <= FLOW_TOL
    eng.set_gemm_variant(-1)
    empty_q = torch.zeros(0, 7, device=DEV)
    empty_p = torch.zeros(0, 7, device=DEV)
    assert eng.forward_kinematics(empty_q).shape == (0, 7) and eng.lm_step(empty_p, empty_q).shape == (0, 7)
    assert eng.pose_error(empty_q, empty_p)[0].shape == (0,) and eng.joint_limits_exceeded(empty_q).shape == (0,)
    sol, valid = eng.refine_exact(empty_p, empty_q, 3, 1e-3, 0.1)
    assert sol.shape == (0, 7) and valid.shape == (0,)
    sol, valid = eng.generate_exact(empty_p, (1, 3), 1e-3, 0.1, seed_fn=lambda r, idx, rep: empty_q)
    assert sol.shape == (0, 7) and valid.shape == (0,)


@pytest.mark.parametrize("kw", [
    dict(nb_nodes=3, dim=7, n_hidden=3, width=1024),                       # the released shape
    dict(nb_nodes=2, dim=9, n_hidden=2, width=256),                        # TINY's: the one-launch form is also the LAST contraction
    dict(nb_nodes=2, dim=10, n_hidden=4, width=512, robot_name="fetch_arm"),
    dict(nb_nodes=2, dim=8, n_hidden=2, width=1024, robot_name="fetch"),
])
def test_small_batch_one_launch_form_equals_two_launches(kw):
    """k_entry_gemm_skinny (entry kernel + first hidden contraction in one launch, <= 256 tiles; the first Linear on the matrix
    pipe, accumulating from the bias with k ascending) against the two-launch form it replaces (ikf_set_gemm_variant 110 =
    off, 112 = forced for every batch it supports; 111 = the default): the f32 MFMA is an fmaf chain, so with the 32-row tiles the two forms
    must give identical bits; the 16-row head (<= 128 rows) adds the partial sums in another fixed order and agrees to rounding; all match the oracle."""
    robot, hp, lay, sd = custom_model(seed=12, gain=1.5, **kw)
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    n_max = 512
    _, poses = reachable_poses(robot, n_max, 95)
    lat = latents(n_max, lay.dim, 96)
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat, clamp=False)
    for n in (1, 31, 32, 33, 200, 256, 257, 400, 512):
        P, L = poses[:n].to(DEV), lat[:n].to(DEV)
        kw_n = dict(n=(1 if n == 1 else None), latent=L, clamp_to_joint_limits=False)
        eng.set_gemm_variant(110)
        two = s.generate_ik_solutions(P, **kw_n)
        eng.set_gemm_variant(112)
        one = s.generate_ik_solutions(P, **kw_n)
        if n > 128:
            assert torch.equal(one, two), f"{kw} n={n}: max diff {(one - two).abs().max().item():.3e}"
        else:  # the 16-row head sums the partial-sum slots per wave first (pending16_issue): another fixed order, equal to rounding
            assert (one - two).abs().max().item() <= 5e-6, f"{kw} n={n}: max diff {(one - two).abs().max().item():.3e}"
            assert torch.equal(one, s.generate_ik_solutions(P, **kw_n)), "and the same bits every time"
        err = ((one.cpu() - ref[:n]).abs() / torch.clamp(ref[:n].abs(), min=1.0)).max().item()
        assert err <= FLOW_TOL, f"{kw} n={n}: {err:.2e}"
    # softflow column and the exact path (tile-major pose gather through pose_idx) go through it too
    cond = torch.cat([poses[:100], torch.full((100, 1), 0.4)], dim=1)
    if lay.dim_cond == 8:
        got = eng.generate_approx(poses[:100].to(DEV), lat[:100].to(DEV), False, softflow_scale=0.4).cpu()
        assert (got - fo.run_inference_torch(sd, lay, robot, lat[:100], cond, False)).abs().max().item() <= 10 * FLOW_TOL
    res = []
    for variant in (110, 112):
        eng.set_gemm_variant(variant)
        torch.manual_seed(5)
        res.append(s.generate_exact_ik_solutions(poses[:100].to(DEV), pos_error_threshold=0.05, rot_error_threshold=0.5))
    # (rounds of <= 128 rows take the 16-row head: seeds equal to rounding, refined to the same solutions)
    assert torch.equal(res[0][1], res[1][1]) and (res[0][0] - res[1][0]).abs().max().item() <= 1e-3
    eng.set_gemm_variant(111)


@pytest.mark.parametrize("kw", [
    dict(nb_nodes=3, dim=7, n_hidden=3, width=1024),                                    # the released Panda shape (L1 = 3, L2 = 4)
    dict(nb_nodes=2, dim=10, n_hidden=3, width=1024, robot_name="fetch_arm"),           # FetchArm's (L1 = L2 = 5: 13 first-Linear inputs)
    dict(nb_nodes=2, dim=8, n_hidden=3, width=1024, robot_name="fetch"),                # Fetch's (8 dof)
    dict(nb_nodes=2, dim=7, n_hidden=3, width=1000),                                    # a width the engine pads to 1024
    dict(nb_nodes=2, dim=7, n_hidden=3, width=1024, softflow=False, sigmoid=True),      # sigmoid_on_output graph, 7-entry conditional
])
def test_row_owner_form_matches_oracle_and_the_per_layer_kernels(kw):
    """k_flow_rowowner (one launch per call: a workgroup keeps 16 rows on chip through every subnet, weights streamed past them) forced
    for every batch (ikf_set_gemm_variant 182) against the oracle and against the per-layer kernels (180): ragged row counts on both
    sides of the 16-row workgroup boundary, more than one round of 256 workgroups, the single-pose broadcast, a non-zero softflow
    entry, unclamped outputs, and the exact path's tile-major pose gather.  Another summation order than the per-layer kernels:
    equal to rounding, each within 1e-5 of the oracle."""
    robot, hp, lay, sd = custom_model(seed=21, gain=1.5, **kw)
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    n_max = 4096 + 37
    _, poses = reachable_poses(robot, n_max, 131)
    lat = latents(n_max, lay.dim, 132)
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat, clamp=False)
    for n in (1, 15, 16, 17, 100, 1000, n_max):
        P, L = poses[:n].to(DEV), lat[:n].to(DEV)
        kw_n = dict(n=(1 if n == 1 else None), latent=L, clamp_to_joint_limits=False)
        eng.set_gemm_variant(182)
        ro = s.generate_ik_solutions(P, **kw_n)
        assert torch.equal(ro, s.generate_ik_solutions(P, **kw_n)), "the same bits every time"
        eng.set_gemm_variant(180)
        layered = s.generate_ik_solutions(P, **kw_n)
        scale = torch.clamp(ref[:n].abs(), min=1.0)
        err = ((ro.cpu() - ref[:n]).abs() / scale).max().item()
        dif = ((ro - layered).cpu().abs() / scale).max().item()
        assert err <= FLOW_TOL, f"{kw} n={n}: {err:.2e} from the oracle"
        assert dif <= FLOW_TOL, f"{kw} n={n}: {dif:.2e} from the per-layer kernels"
    eng.set_gemm_variant(182)
    # clamped (the API default) and the single-pose form y.expand(n, 7)
    n = 333
    got = s.generate_ik_solutions(poses[5].to(DEV), n=n, latent=lat[:n].to(DEV)).cpu()
    want = fo.generate_ik_solutions_torch(sd, lay, robot, poses[5:6].expand(n, 7), lat[:n], clamp=True)
    assert (got - want).abs().max().item() <= FLOW_TOL
    if lay.dim_cond == 8:
        cond = torch.cat([poses[:100], torch.full((100, 1), 0.4)], dim=1)
        got = eng.generate_approx(poses[:100].to(DEV), lat[:100].to(DEV), False, softflow_scale=0.4).cpu()
        assert (got - fo.run_inference_torch(sd, lay, robot, lat[:100], cond, False)).abs().max().item() <= 10 * FLOW_TOL
    # exact IK: the retry rounds' flow rows (poses gathered through the active-pose list, tile-major repeats) through the row-owner launch
    res = []
    for variant in (180, 182):
        eng.set_gemm_variant(variant)
        torch.manual_seed(5)
        res.append(s.generate_exact_ik_solutions(poses[:100].to(DEV), pos_error_threshold=0.05, rot_error_threshold=0.5))
    assert torch.equal(res[0][1], res[1][1]) and (res[0][0] - res[1][0]).abs().max().item() <= 1e-3
    eng.set_gemm_variant(181)


@pytest.mark.parametrize("kw", [
    dict(nb_nodes=3, dim=7, n_hidden=3, width=1024),                                    # the released Panda shape
    dict(nb_nodes=2, dim=10, n_hidden=3, width=1024, robot_name="fetch_arm"),           # FetchArm's (13 first-Linear inputs)
    dict(nb_nodes=2, dim=8, n_hidden=3, width=1024, robot_name="fetch"),
    dict(nb_nodes=2, dim=7, n_hidden=3, width=1024, softflow=False, sigmoid=True),
])
def test_cluster_form_matches_oracle_and_the_row_owner_form(kw):
    """k_flow_cluster<G> (G = 8 / 4 / 2 workgroups per 16-row tile split the hidden columns and exchange h2 and the last Linear's partial
    sums inside the launch) forced wherever its grid fits (ikf_set_gemm_variant 187) against the oracle, the per-layer kernels (185 + 180)
    and the row-owner launch (182): every G, ragged row counts, one tile, the single-pose form, a softflow entry, the exact path.  The
    members' rotated k order is fixed per output column: the same bits every time; equal to the other forms to rounding."""
    robot, hp, lay, sd = custom_model(seed=23, gain=1.5, **kw)
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    n_max = 2048
    _, poses = reachable_poses(robot, n_max, 141)
    lat = latents(n_max, lay.dim, 142)
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat, clamp=False)
    for n in (1, 16, 17, 100, 128, 129, 200, 256, 257, 300, 512, 513, 1000, 1024, 1025, 2000, 2048):   # G = 32 up to 128 rows (k split 4 ways),
        # 16 up to 256 (2 ways), 8 up to 512, 4 up to 1024, 2 up to 2048
        P, L = poses[:n].to(DEV), lat[:n].to(DEV)
        kw_n = dict(n=(1 if n == 1 else None), latent=L, clamp_to_joint_limits=False)
        eng.set_gemm_variant(187)
        assert "cluster" in eng.dominant_kernel_name(n)
        cl = s.generate_ik_solutions(P, **kw_n)
        for _ in range(3):
            assert torch.equal(cl, s.generate_ik_solutions(P, **kw_n)), "the same bits every time"
        eng.set_gemm_variant(185)
        eng.set_gemm_variant(182)
        ro = s.generate_ik_solutions(P, **kw_n)
        eng.set_gemm_variant(180)
        layered = s.generate_ik_solutions(P, **kw_n)
        eng.set_gemm_variant(181)
        scale = torch.clamp(ref[:n].abs(), min=1.0)
        err = ((cl.cpu() - ref[:n]).abs() / scale).max().item()
        assert err <= FLOW_TOL, f"{kw} n={n}: {err:.2e} from the oracle"
        assert ((cl - ro).cpu().abs() / scale).max().item() <= FLOW_TOL and ((cl - layered).cpu().abs() / scale).max().item() <= FLOW_TOL
    eng.set_gemm_variant(187)
    n = 333
    got = s.generate_ik_solutions(poses[5].to(DEV), n=n, latent=lat[:n].to(DEV)).cpu()
    want = fo.generate_ik_solutions_torch(sd, lay, robot, poses[5:6].expand(n, 7), lat[:n], clamp=True)
    assert (got - want).abs().max().item() <= FLOW_TOL
    if lay.dim_cond == 8:
        cond = torch.cat([poses[:100], torch.full((100, 1), 0.4)], dim=1)
        got = eng.generate_approx(poses[:100].to(DEV), lat[:100].to(DEV), False, softflow_scale=0.4).cpu()
        assert (got - fo.run_inference_torch(sd, lay, robot, lat[:100], cond, False)).abs().max().item() <= 10 * FLOW_TOL
    res = []
    for variant in (185, 187):
        eng.set_gemm_variant(variant)
        torch.manual_seed(5)
        res.append(s.generate_exact_ik_solutions(poses[:100].to(DEV), pos_error_threshold=0.05, rot_error_threshold=0.5))
    assert torch.equal(res[0][1], res[1][1]) and (res[0][0] - res[1][0]).abs().max().item() <= 1e-3
    eng.set_gemm_variant(186)


def test_cluster_form_default_split_and_many_calls_in_flight():
    """Defaults: 512 rows = one cluster launch (G = 8), 4096 + 512 = the row-owner launch + a cluster launch for the rest; 200 calls of
    mixed sizes queued back to back without a synchronisation in between (the epoch words are re-zeroed by a memset in front of every
    launch, the exchange buffers are reused) reproduce the first results bit for bit."""
    robot, hp, lay, sd = panda_model()
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    n = 4096 + 512
    _, poses = reachable_poses(robot, n, 17)
    lat = latents(n, lay.dim, 18)
    P, L = poses.to(DEV), lat.to(DEV)
    assert "cluster" in eng.dominant_kernel_name(512) and "rowowner" in eng.dominant_kernel_name(n) and "cluster" in eng.dominant_kernel_name(1)
    assert "cluster" in eng.dominant_kernel_name(128) and "cluster" in eng.dominant_kernel_name(16)
    eng.profile_begin()
    full = s.generate_ik_solutions(P, latent=L)
    n_launch, _ = eng.profile_end()
    assert n_launch == 2, "one row-owner launch + one cluster launch"
    eng.set_gemm_variant(182)
    ro = s.generate_ik_solutions(P, latent=L)
    eng.set_gemm_variant(181)
    assert torch.equal(full[:4096], ro[:4096]) and (full[4096:] - ro[4096:]).abs().max().item() <= FLOW_TOL
    sizes = [512, 300, 1024, 2048, 700, 512, 100, 16, 200]
    first = {k: s.generate_ik_solutions(P[:k], latent=L[:k]).clone() for k in set(sizes)}
    outs = []
    for i in range(200):
        k = sizes[i % len(sizes)]
        outs.append((k, s.generate_ik_solutions(P[:k], latent=L[:k])))
    torch.cuda.synchronize()
    for k, o in outs:
        assert torch.equal(o, first[k]), k
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses[:512], lat[:512])
    assert (first[512].cpu() - ref).abs().max().item() <= FLOW_TOL


def test_cluster_form_repair_launch_when_a_peer_never_arrives():
    """A cluster launch one workgroup short (ikf_set_gemm_variant 188): the members of that workgroup's row tile wait in vain, the wait
    runs out (bounded), every other wait ends through the abort word, and the predicated row-owner launch queued behind it recomputes the
    rows - the caller's results are the row-owner form's, bit for bit, with no error; the handle counts the repair and stops using the form."""
    robot, hp, lay, sd = panda_model()
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    n = 512
    _, poses = reachable_poses(robot, n, 27)
    lat = latents(n, lay.dim, 28)
    P, L = poses.to(DEV), lat.to(DEV)
    eng.set_gemm_variant(182)
    ro = s.generate_ik_solutions(P, latent=L)
    eng.set_gemm_variant(181)
    good = s.generate_ik_solutions(P, latent=L)
    assert "cluster" in eng.dominant_kernel_name(n) and eng.cluster_repairs == 0
    eng.set_gemm_variant(188)
    out = torch.full_like(good, float("nan"))
    out.copy_(s.generate_ik_solutions(P, latent=L))
    torch.cuda.synchronize()
    assert torch.equal(out, ro), "the repair launch's rows"
    assert eng.cluster_repairs == 1 and "cluster" not in eng.dominant_kernel_name(n)
    again = s.generate_ik_solutions(P, latent=L)   # the form sits out the next calls
    assert (again - good).abs().max().item() <= FLOW_TOL and eng.cluster_repairs == 1


@pytest.mark.gpu
def test_cluster_form_is_tried_again_after_a_pause_that_doubles():
    """A wait that ran out means another tenant held CUs at that moment - not for ever.  After a give-up the cluster form sits out 16
    calls (ikf_cluster_backoff counts them down), then runs again at its usual speed with the cluster form's own bits; a second give-up
    pauses 32 calls.  Results are valid throughout (repair launch / the other forms), ikf_cluster_repairs keeps counting, and reloading
    the weights forgets the history."""
    robot, hp, lay, sd = panda_model()
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    n = 512
    _, poses = reachable_poses(robot, n, 127)
    lat = latents(n, lay.dim, 128)
    P, L = poses.to(DEV), lat.to(DEV)
    good = s.generate_ik_solutions(P, latent=L).clone()
    assert eng.plan(n) == "cluster8:512" and eng.cluster_backoff == 0
    for round_, pause in ((1, 16), (2, 32)):
        eng.set_gemm_variant(188)              # the next cluster launch is one workgroup short: its tile's waits run out
        out = s.generate_ik_solutions(P, latent=L).clone()
        torch.cuda.synchronize()
        assert (out - good).abs().max().item() <= FLOW_TOL
        assert eng.cluster_repairs == round_ and eng.cluster_backoff == pause
        assert "cluster" not in eng.plan(n), eng.plan(n)
        for k in range(pause):
            assert eng.cluster_backoff == pause - k
            out = s.generate_ik_solutions(P, latent=L)
            assert (out - good).abs().max().item() <= FLOW_TOL
        assert eng.cluster_backoff == 0 and eng.plan(n) == "cluster8:512"
        back = s.generate_ik_solutions(P, latent=L).clone()
        torch.cuda.synchronize()
        assert torch.equal(back, good), "the cluster form again, bit for bit"
        assert eng.cluster_repairs == round_
    # ... and at its usual speed: faster than the per-layer kernels that replaced it during the pause, timed in the same run on the same GPU
    # (0.48 against 0.71 ms on an idle MI355X; no absolute bound - a shared or down-clocked GPU is exactly where the pause matters)
    def ms_per_call():
        for _ in range(20):
            s.generate_ik_solutions(P, latent=L)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            s.generate_ik_solutions(P, latent=L)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 50 * 1e3

    ms = ms_per_call()
    eng.set_gemm_variant(185)
    assert "cluster" not in eng.plan(n)
    ms_per_layer = ms_per_call()
    eng.set_gemm_variant(186)
    assert ms < ms_per_layer, f"{ms:.3f} ms per 512-row call after the pause against {ms_per_layer:.3f} on the per-layer kernels"
    s.load_state_dict_tensors(sd)
    assert s.engine(DEV).cluster_backoff == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n", [130, 256, 300, 512, 1000, 1024, 1536, 2048, 2600])
def test_cluster_form_tagged_hand_over_against_epoch_words_and_the_oracle(n):
    """The default hand-over of the cluster form with 2 .. 16 members (r05): every exchanged float carries its subnet's parity in the last
    mantissa bit - no drain, no epoch word, no memset in front of the launch; a consumer re-reads what still shows the other parity.  The
    tag moves an activation by at most one ulp, so the two hand-overs agree to rounding (not bit for bit), each is reproducible bit for bit,
    both sit inside the tolerance against the oracle at coupling coefficients of O(1), and XCD-local and spread placement still give the
    same bits (same tagged values, another memory path)."""
    robot, hp, lay, sd = panda_model(gain=2.0)
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    _, poses = reachable_poses(robot, n, 141)
    lat = latents(n, lay.dim, 142)
    P, L = poses.to(DEV), lat.to(DEV)
    tagged = [s.generate_ik_solutions(P, latent=L, clamp_to_joint_limits=False).clone() for _ in range(4)]
    eng.set_gemm_variant(189)
    tagged_spread = s.generate_ik_solutions(P, latent=L, clamp_to_joint_limits=False).clone()
    eng.set_gemm_variant(190)
    eng.set_gemm_variant(192)
    words = s.generate_ik_solutions(P, latent=L, clamp_to_joint_limits=False).clone()
    eng.set_gemm_variant(193)
    tagged.append(s.generate_ik_solutions(P, latent=L, clamp_to_joint_limits=False).clone())
    torch.cuda.synchronize()
    assert all(torch.equal(tagged[0], t) for t in tagged[1:]) and torch.equal(tagged[0], tagged_spread)
    scale = torch.clamp(words.abs(), min=1.0)
    assert ((tagged[0] - words).abs() / scale).max().item() <= 8e-6   # (an ulp on an activation, carried through 24 subnets at gain 2)
    ref = torch.tensor(fo.flow_inverse_f64(sd, lay, lat.numpy(), torch.cat([poses, torch.zeros(n, 1)], 1).numpy())[:, : lay.ndof])
    for got in (tagged[0], words):
        assert ((got.cpu().double() - ref).abs() / torch.clamp(ref.abs(), min=1.0)).max().item() <= FLOW_TOL
    assert eng.cluster_repairs == 0


@pytest.mark.gpu
def test_cluster_form_tagged_buffers_after_a_give_up_with_calls_already_queued():
    """The tagged hand-over keeps an invariant between calls - every float of its exchange buffers has parity 1 - that a launch which gave up
    breaks.  The host only learns of a give-up when it plans a later call, so launches may already be queued behind the broken one: each of
    them finds the abort word still set, returns at once and leaves its rows to its own repair launch; the first call planned after the host
    has seen the give-up re-creates the buffers.  Here: one launch a workgroup short, five more calls queued behind it without a
    synchronisation, every result inside the tolerance, and after the pause the form is back with its own bits."""
    robot, hp, lay, sd = panda_model()
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    sets = []
    for k, n in enumerate((512, 300, 1024, 512, 2048, 200)):
        _, poses = reachable_poses(robot, n, 150 + k)
        sets.append((poses.to(DEV), latents(n, lay.dim, 160 + k).to(DEV)))
    good = [s.generate_ik_solutions(P, latent=L).clone() for P, L in sets]
    torch.cuda.synchronize()
    eng.set_gemm_variant(188)
    outs = [s.generate_ik_solutions(P, latent=L).clone() for P, L in sets]     # no synchronisation in between
    torch.cuda.synchronize()
    for g, o in zip(good, outs):
        assert (g - o).abs().max().item() <= FLOW_TOL
    assert eng.cluster_repairs == 1 and eng.cluster_backoff > 0
    while eng.cluster_backoff > 0:
        s.generate_ik_solutions(*sets[0][:1], latent=sets[0][1])
    again = [s.generate_ik_solutions(P, latent=L).clone() for P, L in sets]
    torch.cuda.synchronize()
    assert all(torch.equal(a, g) for a, g in zip(again, good)) and eng.cluster_repairs == 1


@pytest.mark.gpu
def test_small_batch_weight_image_is_built_on_first_use_only():
    """The fragment-major image of the small-batch per-layer kernels (+ 201 MB for Panda, 48 pack launches) is no longer part of
    ikf_load_weights: a handle whose small batches run the cluster form does not build it by itself; the first <= 512-row chunk that does take
    the per-layer path builds it (same results as before), and ikf_reserve ALWAYS does so ahead of time (r06) - a cluster-form handle falls back
    to these kernels while another process holds CUs, the worst moment for an allocation and a device-wide synchronisation inside a call."""
    robot, hp, lay, sd = panda_model()
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    n = 200
    _, poses = reachable_poses(robot, n, 131)
    lat = latents(n, lay.dim, 132)
    P, L = poses.to(DEV), lat.to(DEV)
    assert eng.load_time_ms > 0.0 and eng.frag_image_time_ms == 0.0
    by_plan = s.generate_ik_solutions(P, latent=L).clone()
    assert eng.plan(n).startswith("cluster") and eng.frag_image_time_ms == 0.0
    eng.set_gemm_variant(180)
    eng.set_gemm_variant(185)
    assert eng.plan(n) == f"perlayer:{n}"
    per_layer = s.generate_ik_solutions(P, latent=L).clone()
    torch.cuda.synchronize()
    assert eng.frag_image_time_ms > 0.0
    assert (per_layer - by_plan).abs().max().item() <= FLOW_TOL
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat)
    assert (per_layer.cpu() - ref).abs().max().item() <= FLOW_TOL
    # ikf_reserve builds the image before the first call - on a default handle (cluster form allowed) too
    s2 = _solver(robot, hp, sd)
    e2 = s2.engine(DEV)
    assert e2.plan(n).startswith("cluster") and e2.frag_image_time_ms == 0.0
    e2.reserve(512)
    built = e2.frag_image_time_ms
    assert built > 0.0
    e2.set_gemm_variant(180)
    e2.set_gemm_variant(185)
    assert torch.equal(s2.generate_ik_solutions(P, latent=L), per_layer) and e2.frag_image_time_ms == built


@pytest.mark.parametrize("n", [129, 200, 256, 257, 300, 500, 512, 700, 1000, 1024])
def test_cluster_form_xcd_local_hand_over_equals_the_spread_form(n):
    """G = 4 / 8 / 16: the default form keeps a row tile's members on ONE XCD and hands activations over through that XCD's L2 (plain stores);
    ikf_set_gemm_variant 189 spreads the members over the XCDs (write-through stores, the form of the other member counts).  Same
    arithmetic, another memory path: bit-for-bit the same results, ragged last tiles and grids padded to groups of 8 row tiles included,
    and no repair - the placement check inside the launch found every member where the grid mapping expects it."""
    robot, hp, lay, sd = panda_model()
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    _, poses = reachable_poses(robot, n, 61)
    lat = latents(n, lay.dim, 62)
    P, L = poses.to(DEV), lat.to(DEV)
    assert eng.plan(n) == f"cluster{16 if n <= 256 else (8 if n <= 512 else 4)}:{n}"
    assert eng.cluster_local, "the placement census at load: workgroups b and b + 8 k of a grid share an XCD on an MI355X"
    local = [s.generate_ik_solutions(P, latent=L).clone() for _ in range(3)]
    eng.set_gemm_variant(189)
    spread = s.generate_ik_solutions(P, latent=L).clone()
    eng.set_gemm_variant(190)
    local.append(s.generate_ik_solutions(P, latent=L).clone())
    torch.cuda.synchronize()
    assert all(torch.equal(o, spread) for o in local) and eng.cluster_repairs == 0 and eng.cluster_local
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat)
    assert (spread.cpu() - ref).abs().max().item() <= FLOW_TOL


def test_cluster_form_xcd_local_placement_check_falls_back_to_the_spread_form():
    """Nothing about placement is assumed: every member publishes its XCC_ID in the top byte of its epoch words, and a consumer that meets
    another XCD's id gives up BEFORE it reads that peer's payload (which would sit in an L2 it cannot see).  ikf_set_gemm_variant 191 makes
    workgroup 0 of the next XCD-local launch publish a wrong id: its peers give up (host word 2), the repair launch recomputes the rows - the
    caller gets the row-owner form's results bit for bit -, the handle counts it and goes on with the SPREAD cluster form, not without the
    cluster form."""
    robot, hp, lay, sd = panda_model()
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    n = 512
    _, poses = reachable_poses(robot, n, 63)
    lat = latents(n, lay.dim, 64)
    P, L = poses.to(DEV), lat.to(DEV)
    eng.set_gemm_variant(182)
    ro = s.generate_ik_solutions(P, latent=L).clone()
    eng.set_gemm_variant(181)
    eng.set_gemm_variant(192)   # the epoch-word hand-over: the tagged one (default) validates the payload itself and publishes no ids
    good = s.generate_ik_solutions(P, latent=L).clone()
    assert eng.cluster_repairs == 0 and eng.cluster_local
    eng.set_gemm_variant(191)
    out = torch.full_like(good, float("nan"))
    out.copy_(s.generate_ik_solutions(P, latent=L))
    torch.cuda.synchronize()
    assert torch.equal(out, ro), "the repair launch's rows"
    assert eng.cluster_repairs == 1 and "cluster" in eng.dominant_kernel_name(n) and eng.plan(n) == "cluster8:512" and not eng.cluster_local
    again = s.generate_ik_solutions(P, latent=L)   # the spread form from now on: same bits as the local form gave
    torch.cuda.synchronize()
    assert torch.equal(again, good) and eng.cluster_repairs == 1
    eng.set_gemm_variant(191)                       # (no XCD-local launch any more: the hook stays unused, nothing gives up)
    assert torch.equal(s.generate_ik_solutions(P, latent=L), good) and eng.cluster_repairs == 1


def test_cluster_form_tagged_xcd_local_placement_is_classified_when_a_wait_runs_out():
    """The DEFAULT hand-over (tagged payload) in its XCD-local form assumes nothing about placement either (r06).  It has no epoch words; a member on
    another XCD would keep its plain payload stores in ITS L2, its peers would see the old parity, re-read, and run out of patience (2 - 4 ms) - never a
    wrong value.  So every member writes (launch number, XCC_ID) into a word of its own at the start of every launch, and the HOST, when it folds the
    give-up, reads the words of that launch: members of one row tile that ran it on different XCDs make it a placement failure (the handle goes on with
    the SPREAD form, without a pause) instead of "a peer is not resident" (pause).  Variant 191 makes workgroup 0 such a member - wrong id, payload stores dropped: the
    repair launch recomputes the rows, and the spread form gives the same bits as before."""
    robot, hp, lay, sd = panda_model()
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    n = 512
    _, poses = reachable_poses(robot, n, 163)
    lat = latents(n, lay.dim, 164)
    P, L = poses.to(DEV), lat.to(DEV)
    eng.set_gemm_variant(182)
    ro = s.generate_ik_solutions(P, latent=L).clone()
    eng.set_gemm_variant(181)
    good = [s.generate_ik_solutions(P, latent=L).clone() for _ in range(3)]   # (consecutive launches: another launch number each)
    assert all(torch.equal(g, good[0]) for g in good) and eng.cluster_repairs == 0 and eng.cluster_local and eng.plan(n) == "cluster8:512"
    eng.set_gemm_variant(191)
    out = torch.full_like(ro, float("nan"))
    out.copy_(s.generate_ik_solutions(P, latent=L))
    torch.cuda.synchronize()
    assert torch.equal(out, ro), "the repair launch's rows"
    assert eng.cluster_repairs == 1 and not eng.cluster_local and eng.cluster_backoff == 0 and eng.plan(n) == "cluster8:512"
    again = s.generate_ik_solutions(P, latent=L)
    torch.cuda.synchronize()
    assert torch.equal(again, good[0]) and eng.cluster_repairs == 1   # spread placement: the same tagged values through another memory path


@pytest.mark.parametrize("n", [512, 4096, 5000])
def test_approximate_call_can_be_captured_into_a_hip_graph(n):
    """After ikf_load_weights + ikf_reserve an approximate call allocates nothing and never synchronises (include/ikflow_amd.h), so it can be
    captured into a HIP graph (torch.cuda.CUDAGraph) - cluster form, row-owner launch and a two-chunk plan - and the replay writes the call's bits."""
    robot, hp, lay, sd = panda_model()
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    eng.reserve(8192)
    _, poses = reachable_poses(robot, n, 171)
    lat = latents(n, lay.dim, 172)
    P, L = poses.to(DEV), lat.to(DEV)
    ref = s.generate_ik_solutions(P, latent=L).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            s.generate_ik_solutions(P, latent=L)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = s.generate_ik_solutions(P, latent=L)
    for _ in range(3):
        with torch.inference_mode():
            out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
    assert eng.cluster_repairs == 0
    assert torch.equal(s.generate_ik_solutions(P, latent=L), ref)   # (and the handle is as usable as before)


def test_plan_of_a_call_by_batch_size():
    """plan_flow's decisions at representative sizes (released Panda shape, 256 CUs): what DESIGN.md section 4.0 tabulates.  Other shapes
    (TINY: width 256) and the f16x3 mode stay on the per-layer kernels whatever the size."""
    robot, hp, lay, sd = panda_model()
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    want = {1: "cluster32:1", 8: "cluster32:8", 16: "cluster32:16", 128: "cluster32:128", 200: "cluster16:200", 256: "cluster16:256", 300: "cluster8:300",
            512: "cluster8:512", 600: "cluster8:512 cluster32:88", 1024: "cluster4:1024", 1536: "cluster8:512 cluster4:1024", 2048: "cluster2:2048",
            2560: "cluster8:512 cluster2:2048", 3072: "cluster4:1024 cluster2:2048", 3400: "rowowner:3400", 4096: "rowowner:4096",
            4096 + 200: "rowowner:4096 cluster16:200", 8192: "rowowner:8192", 3 * 4096 + 3500: "rowowner:15788",
            1_000_000 // 8: "rowowner:122880 cluster2:2048 cluster32:72",
            # a last row next to another form stays on the resident-row forms (one weight image in the Infinity Cache, not two)
            513: "cluster8:512 cluster32:1", 1025: "cluster4:1024 cluster32:1", 2049: "cluster2:2048 cluster32:1", 4097: "rowowner:4096 cluster32:1"}
    import time

    t0 = time.perf_counter()
    for n in range(1, 4096, 37):   # the plan of a call is made on the host in front of every call: a few microseconds, whatever the tail
        eng.plan(n)
    assert (time.perf_counter() - t0) / 111 < 2e-3
    got = {n: eng.plan(n) for n in want}
    assert got == want, {n: (got[n], want[n]) for n in want if got[n] != want[n]}
    eng.set_gemm_variant(185)
    assert eng.plan(512) == "perlayer:512" and eng.plan(4096 + 200) == "rowowner:4096 perlayer:200"
    eng.set_gemm_variant(180)
    assert eng.plan(4096) == "perlayer:4096"
    eng.set_gemm_variant(181); eng.set_gemm_variant(186)
    s.set_precision("f16x3")
    assert s.engine(DEV).plan(4096) == "perlayer:4096"
    r2, h2, l2, sd2 = tiny_model()
    assert _solver(r2, h2, sd2).engine(DEV).plan(4096) == "perlayer:4096"


def test_row_owner_form_is_what_the_baseline_batch_runs():
    """By default a batch's full rounds of (CUs x 16) rows (and a last partial round when that is the cheapest plan) take the row-owner
    launch, the rest the cheapest mix of cluster launches and per-layer kernels (plan_flow): 4096 rows = one launch; 4096 + 200 = one launch
    + a 200-row cluster chunk (G = 16); with the cluster form off (185) + the per-layer kernels for the 200 rows.  Results are those of the
    forced forms row for row (identical bits with the form that ran them)."""
    robot, hp, lay, sd = panda_model()
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    n = 4096 + 200
    _, poses = reachable_poses(robot, n, 7)
    lat = latents(n, lay.dim, 8)
    P, L = poses.to(DEV), lat.to(DEV)
    eng.set_gemm_variant(182)
    ro = s.generate_ik_solutions(P, latent=L)
    eng.set_gemm_variant(180); eng.set_gemm_variant(185)
    layered_tail = s.generate_ik_solutions(P[4096:], latent=L[4096:])
    eng.set_gemm_variant(187)
    cluster_tail = s.generate_ik_solutions(P[4096:], latent=L[4096:])
    eng.set_gemm_variant(181); eng.set_gemm_variant(186)
    auto = s.generate_ik_solutions(P, latent=L)
    assert torch.equal(auto[:4096], ro[:4096])
    assert torch.equal(auto[4096:], cluster_tail)
    eng.set_gemm_variant(185)
    auto_no_cluster = s.generate_ik_solutions(P, latent=L)
    assert torch.equal(auto_no_cluster[:4096], ro[:4096]) and torch.equal(auto_no_cluster[4096:], layered_tail)
    eng.set_gemm_variant(186)
    eng.profile_begin()
    s.generate_ik_solutions(P[:4096], latent=L[:4096])
    n_launch, _ = eng.profile_end()
    assert n_launch == 1, "4096 rows = one row-owner launch"
    assert "rowowner" in eng.dominant_kernel_name(4096) and "cluster" in eng.dominant_kernel_name(1)   # (one weight image for every size)


@pytest.mark.parametrize("kw", [
    dict(nb_nodes=3, dim=7, n_hidden=3, width=1024),                       # the released shape
    dict(nb_nodes=2, dim=9, n_hidden=2, width=256),                        # TINY's: the contraction with the tail is the subnet's only one
    dict(nb_nodes=2, dim=10, n_hidden=4, width=512, robot_name="fetch_arm"),
    dict(nb_nodes=3, dim=8, n_hidden=2, width=1024, robot_name="fetch"),
    dict(nb_nodes=2, dim=7, n_hidden=3, width=768, softflow=False, sigmoid=True),  # padded width, sigmoid graph (dim_cond 7)
])
def test_in_launch_entry_phase_equals_entry_launches(kw):
    """The next subnet's entry phase in the tail of the last hidden contraction (TailSync: write-through partial sums, row-tile
    arrival counter, agent-scope reads; ikf_set_gemm_variant 121 - an opt-in: it measured slower) against the default
    form with a k_subnet_entry launch between the subnets (120): same arithmetic in the same order, so identical bits - on both kernels that carry the tail (128x128
    tiles, 32x64 small-batch tiles), at the largest batches they take it for, with ragged last row tiles, repeated (the counters
    are re-zeroed by every call), through the exact path (pose gather), and both match the oracle."""
    robot, hp, lay, sd = custom_model(seed=21, gain=1.5, **kw)
    s = _solver(robot, hp, sd, flavour="probes")
    eng = s.engine(DEV)
    n_max = 4096
    _, poses = reachable_poses(robot, n_max, 97)
    lat = latents(n_max, lay.dim, 98)
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat, clamp=False)
    for tile_cfg, sizes in ((100, (257, 300, 511, 512, 4096)), (101, (1, 100, 128, 129, 1000, 3999, 4096)), (105, (33, 512))):
        for n in sizes:
            P, L = poses[:n].to(DEV), lat[:n].to(DEV)
            kw_n = dict(n=(1 if n == 1 else None), latent=L, clamp_to_joint_limits=False)
            eng.set_gemm_variant(tile_cfg)  # 100: tile by batch size; 101: 128x128 tiles forced; 105: 32x64 small-batch tiles forced
            eng.set_gemm_variant(120)
            two = s.generate_ik_solutions(P, **kw_n)
            eng.set_gemm_variant(121)
            one = s.generate_ik_solutions(P, **kw_n)
            again = s.generate_ik_solutions(P, **kw_n)
            assert torch.equal(one, two), f"{kw} cfg {tile_cfg} n={n}: max diff {(one - two).abs().max().item():.3e}"
            assert torch.equal(one, again)
            err = ((one.cpu() - ref[:n]).abs() / torch.clamp(ref[:n].abs(), min=1.0)).max().item()
            assert err <= FLOW_TOL, f"{kw} cfg {tile_cfg} n={n}: {err:.2e}"
    eng.set_gemm_variant(100)
    if lay.dim_cond == 8:  # softflow column
        cond = torch.cat([poses[:300], torch.full((300, 1), 0.4)], dim=1)
        got = eng.generate_approx(poses[:300].to(DEV), lat[:300].to(DEV), False, softflow_scale=0.4).cpu()
        assert (got - fo.run_inference_torch(sd, lay, robot, lat[:300], cond, False)).abs().max().item() <= 10 * FLOW_TOL
    res = []
    for variant in (120, 121):  # exact path: 400 poses, rounds of 400 / <= 1200 / <= 4000 rows (pose gather through pose_idx)
        eng.set_gemm_variant(variant)
        torch.manual_seed(5)
        torch.cuda.manual_seed(5)
        res.append(s.generate_exact_ik_solutions(poses[:400].to(DEV), pos_error_threshold=0.05, rot_error_threshold=0.5))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    eng.set_gemm_variant(121)
    # single-pose form (pose broadcast) at a size that takes the tail
    one = s.generate_ik_solutions(poses[0].to(DEV), n=400, latent=lat[:400].to(DEV))
    eng.set_gemm_variant(120)
    assert torch.equal(one, s.generate_ik_solutions(poses[0].to(DEV), n=400, latent=lat[:400].to(DEV)))


@pytest.mark.parametrize("kw", [
    dict(nb_nodes=3, dim=7, n_hidden=3, width=1024),                              # the released shape
    dict(nb_nodes=2, dim=8, n_hidden=3, width=1024, robot_name="fetch"),          # prismatic joint, softflow column
    dict(nb_nodes=3, dim=7, n_hidden=3, width=1024, softflow=False, sigmoid=True),
    dict(nb_nodes=2, dim=9, n_hidden=2, width=256),                               # not a shape the chain takes: the switch must be a no-op
])
def test_one_launch_chain_equals_per_layer_launches(kw):
    """The whole subnet chain of a <= 128-row call in ONE launch (k_flow_chain16, ikf_set_gemm_variant 171 - an opt-in: it measured slower,
    DESIGN.md section 4): 256 persistent workgroups, (XCC_ID, ticket) = (row tile, column tile), arrival counters in the XCD's L2 between the
    layers, sibling-written operands read past the L1.  Against the per-layer launches on the same 16 x 32 tiles (170 + 158): the same bodies
    in the same order, so identical bits - at every row-tile count 1 .. 8, ragged last tiles, repeated (the last workgroup out re-zeroes the
    control words), interleaved with batches the chain does not take, single-pose form, exact path; and it matches the oracle."""
    robot, hp, lay, sd = custom_model(seed=77, gain=1.5, **kw)
    s = _solver(robot, hp, sd, flavour="probes")
    eng = s.engine(DEV)
    n_max = 300
    _, poses = reachable_poses(robot, n_max, 131)
    lat = latents(n_max, lay.dim, 132)
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat, clamp=False)
    eng.set_gemm_variant(158)  # 16 x 32 tiles for every batch of <= 128 rows, as the chain uses
    for n in (1, 2, 15, 16, 17, 33, 64, 100, 127, 128, 129, 300, 5):
        P, L = poses[:n].to(DEV), lat[:n].to(DEV)
        kw_n = dict(n=(1 if n == 1 else None), latent=L, clamp_to_joint_limits=False)
        eng.set_gemm_variant(170)
        per_layer = s.generate_ik_solutions(P, **kw_n)
        eng.set_gemm_variant(171)
        chain = s.generate_ik_solutions(P, **kw_n)
        again = s.generate_ik_solutions(P, **kw_n)
        assert torch.equal(chain, per_layer), f"{kw} n={n}: max diff {(chain - per_layer).abs().max().item():.3e}"
        assert torch.equal(chain, again)
        err = ((chain.cpu() - ref[:n]).abs() / torch.clamp(ref[:n].abs(), min=1.0)).max().item()
        assert err <= FLOW_TOL, f"{kw} n={n}: {err:.2e}"
    one = s.generate_ik_solutions(poses[0].to(DEV), n=100, latent=lat[:100].to(DEV))  # pose broadcast
    eng.set_gemm_variant(170)
    assert torch.equal(one, s.generate_ik_solutions(poses[0].to(DEV), n=100, latent=lat[:100].to(DEV)))
    if lay.dim_cond == 8:  # softflow column
        eng.set_gemm_variant(171)
        cond = torch.cat([poses[:100], torch.full((100, 1), 0.4)], dim=1)
        got = eng.generate_approx(poses[:100].to(DEV), lat[:100].to(DEV), False, softflow_scale=0.4).cpu()
        assert (got - fo.run_inference_torch(sd, lay, robot, lat[:100], cond, False)).abs().max().item() <= 10 * FLOW_TOL
    res = []
    for variant in (170, 171):  # exact path: 60 poses, rounds of 60 / <= 180 / <= 600 rows (pose gather through pose_idx, row offsets)
        eng.set_gemm_variant(variant)
        torch.manual_seed(5)
        torch.cuda.manual_seed(5)
        res.append(s.generate_exact_ik_solutions(poses[:60].to(DEV), pos_error_threshold=0.05, rot_error_threshold=0.5))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    eng.set_gemm_variant(170)
    eng.set_gemm_variant(159)


@pytest.mark.parametrize("kw", [
    dict(nb_nodes=3, dim=7, n_hidden=3, width=1024),                          # the released shape
    dict(nb_nodes=2, dim=9, n_hidden=2, width=256),                           # TINY's: head and last contraction are one launch
    dict(nb_nodes=2, dim=10, n_hidden=4, width=768, robot_name="fetch_arm"),  # a middle contraction (stores its activation)
    dict(nb_nodes=2, dim=8, n_hidden=3, width=1280, robot_name="fetch"),      # too wide for the one-launch head: entry kernel + 16-row contractions
    dict(nb_nodes=2, dim=7, n_hidden=3, width=512, softflow=False, sigmoid=True),
])
def test_sixteen_row_tiles_for_small_batches(kw):
    """<= 128 rows run on 16 x 32 tiles and <= 64 rows on 16 x 16 tiles (v_mfma_f32_16x16x4_f32; ikf_set_gemm_variant 151 / 159, the
    defaults) instead of 32 x 32 (150): twice / four times the workgroups, a half / a quarter of the matrix-pipe time each.  Another tile shape = another summation order, so the two agree to rounding, not bit
    for bit; each matches the oracle to 1e-5.  Row counts around the 16-row tile edges, the one-launch head and the two-launch form
    (110 / 111), the forced configuration beyond 128 rows (160), the softflow column and the exact path."""
    robot, hp, lay, sd = custom_model(seed=51, gain=1.5, **kw)
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    n_max = 200
    _, poses = reachable_poses(robot, n_max, 103)
    lat = latents(n_max, lay.dim, 104)
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat, clamp=False)
    scale = torch.clamp(ref.abs(), min=1.0)
    try:
        for n in (1, 2, 15, 16, 17, 31, 33, 64, 100, 127, 128):
            P, L = poses[:n].to(DEV), lat[:n].to(DEV)
            kw_n = dict(n=(1 if n == 1 else None), latent=L, clamp_to_joint_limits=False)
            outs = {}
            for tiles in (150, 151, 158):  # 32x32 tiles / the default (16x16 up to 64 rows, 16x32 up to 128) / 16x32 only
                for head in (110, 111):
                    eng.set_gemm_variant(151); eng.set_gemm_variant(159)
                    eng.set_gemm_variant(tiles); eng.set_gemm_variant(head)
                    outs[(tiles, head)] = s.generate_ik_solutions(P, **kw_n).cpu()
                    err = ((outs[(tiles, head)] - ref[:n]).abs() / scale[:n]).max().item()
                    assert err <= FLOW_TOL, f"{kw} n={n} tiles {tiles} head {head}: {err:.2e}"
            for other in ((150, 111), (151, 110), (158, 111), (158, 110)):
                assert ((outs[(151, 111)] - outs[other]).abs() / scale[:n]).max().item() <= 4e-6, other
            eng.set_gemm_variant(151); eng.set_gemm_variant(159); eng.set_gemm_variant(111)
            assert torch.equal(outs[(151, 111)], s.generate_ik_solutions(P, **kw_n).cpu())  # deterministic
        eng.set_gemm_variant(151); eng.set_gemm_variant(111)
        for forced in (160, 161):  # 16x32 / 16x16 tiles forced for a batch that would not pick them
            eng.set_gemm_variant(forced)
            got = s.generate_ik_solutions(poses.to(DEV), latent=lat.to(DEV), clamp_to_joint_limits=False).cpu()
            assert ((got - ref).abs() / scale).max().item() <= FLOW_TOL, forced
        # the 32x32-on-16x16x4 tiles (164: priced and rejected) live in the probes library only; the product refuses the code
        with pytest.raises(Exception, match="probes library"):
            eng.set_gemm_variant(164)
        sp = _solver(robot, hp, sd, flavour="probes")
        sp.engine(DEV).set_gemm_variant(164)
        got = sp.generate_ik_solutions(poses.to(DEV), latent=lat.to(DEV), clamp_to_joint_limits=False).cpu()
        assert ((got - ref).abs() / scale).max().item() <= FLOW_TOL, 164
        eng.set_gemm_variant(100)
        if lay.dim_cond == 8:  # softflow column
            cond = torch.cat([poses[:100], torch.full((100, 1), 0.4)], dim=1)
            got = eng.generate_approx(poses[:100].to(DEV), lat[:100].to(DEV), False, softflow_scale=0.4).cpu()
            assert (got - fo.run_inference_torch(sd, lay, robot, lat[:100], cond, False)).abs().max().item() <= 10 * FLOW_TOL
        # exact path at a size whose first round takes the 16-row tiles (pose gather through pose_idx): thresholds hold
        sol, valid = s.generate_exact_ik_solutions(poses[:60].to(DEV), pos_error_threshold=0.05, rot_error_threshold=0.5)
        if bool(valid.any()):
            pe, re = ko.calculate_pose_error(robot, sol[valid].cpu(), poses[:60][valid.cpu()])
            assert (pe < 0.05 * 1.001).all() and (re < 0.5 * 1.001).all()
    finally:
        eng.set_gemm_variant(100); eng.set_gemm_variant(151); eng.set_gemm_variant(159); eng.set_gemm_variant(111)


def test_activation_store_policy_does_not_change_results():
    """ikf_set_gemm_variant 130..134: the hidden activations leave the contractions / the entry kernel with write-back or
    write-through (sc1) stores - a cache policy, so every setting must give identical bits at every tile configuration."""
    robot, hp, lay, sd = custom_model(nb_nodes=3, dim=7, n_hidden=3, width=1024, seed=41)
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    _, poses = reachable_poses(robot, 4096, 101)
    lat = latents(4096, lay.dim, 102)
    try:
        for n in (5, 200, 300, 512, 700, 1024, 2048, 2049, 4096):
            P, L = poses[:n].to(DEV), lat[:n].to(DEV)
            outs = []
            for code in (130, 131, 132, 133, 134):
                eng.set_gemm_variant(code)
                outs.append(s.generate_ik_solutions(P, latent=L))
            assert all(torch.equal(o, outs[0]) for o in outs[1:]), n
    finally:
        eng.set_gemm_variant(134)


def test_hip_path_against_the_reference_statement_fixtures():
    """tests/golden/ref_exact_loop.npz = outputs of the reference's own `generate_ik_solutions` / `_run_inference` and
    `generate_exact_ik_solutions` / `_generate_exact_ik_solutions` statements (executed over the oracle's flow and kinematics by
    tests/golden/make_ref_exact_loop.py).  The HIP path through the drop-in API on the same inputs: approximate IK within 1e-5
    (batch, single-pose, [1 x 7], unclamped forms); exact IK with the recorded per-round latents - flags agree except where
    a flow-seed rounding difference (1e-6) is amplified across a threshold, solved values agree where the flags do."""
    import os

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_exact_loop.npz"))
    robot, hp, lay, _ = tiny_model()
    sd = fo.make_state_dict(lay, "panda", seed=3, output_gain=1.5)
    s = _solver(robot, hp, sd)
    P, L = torch.from_numpy(z["ik_poses"]).to(DEV), torch.from_numpy(z["ik_latent"]).to(DEV)
    n = P.shape[0]
    assert np.abs(s.generate_ik_solutions(P, latent=L).cpu().numpy() - z["ik_batch_clamped"]).max() <= FLOW_TOL
    got = s.generate_ik_solutions(P, latent=L, clamp_to_joint_limits=False).cpu().numpy()
    assert (np.abs(got - z["ik_batch_unclamped"]) / np.maximum(1.0, np.abs(z["ik_batch_unclamped"]))).max() <= FLOW_TOL
    assert np.abs(s.generate_ik_solutions(P[3].contiguous(), n=n, latent=L).cpu().numpy() - z["ik_single_pose"]).max() <= FLOW_TOL
    assert np.abs(s.generate_ik_solutions(P[3:4].contiguous(), n=n, latent=L).cpu().numpy() - z["ik_single_pose_1x7"]).max() <= FLOW_TOL
    torch.manual_seed(4321)  # the shim draws the latent with the reference's call; on the GPU generator the values differ from the
    drawn = s.generate_ik_solutions(P[5].contiguous(), n=6, latent_scale=0.5)  # CPU fixture, so only the shape / finiteness here
    assert drawn.shape == (6, 7) and bool(torch.isfinite(drawn).all())
    # exact IK (weights seed 2 as in the fixture)
    sd2 = fo.make_state_dict(lay, "panda", seed=2)
    s2 = _solver(robot, hp, sd2)
    agree, total = 0, 0
    for tag in ("a", "b", "c"):
        poses = torch.from_numpy(z[f"{tag}_poses"])
        pos_thr, rot_thr = (float(v) for v in z[f"{tag}_thresholds"])
        lats = [torch.from_numpy(z[f"{tag}_latent_{i}"]).to(DEV) for i in range(int(z[f"{tag}_n_rounds"]))]
        sol, valid = s2.generate_exact_ik_solutions(poses.to(DEV), repeat_counts=(1, 3, 10), pos_error_threshold=pos_thr,
                                                    rot_error_threshold=rot_thr, latents=lats)
        ref_valid = torch.from_numpy(z[f"{tag}_valid"])
        agree += int((valid.cpu() == ref_valid).sum())
        total += ref_valid.numel()
        assert torch.equal(sol.cpu()[~valid.cpu()], torch.zeros_like(sol.cpu()[~valid.cpu()]))
    assert agree >= 0.93 * total, (agree, total)


def test_sample_joint_angles_and_poses_like_the_reference_tests_use_it():
    """jrl.Robot.sample_joint_angles_and_poses as the reference's tests call it (tests/ikflow_solver_test.py:73-75): numpy in
    the limits, poses = FK(q) (checked with the oracle), colliding samples redrawn when asked."""
    from ikflow_amd.robots import Panda

    robot = Panda()
    q, poses = robot.sample_joint_angles_and_poses(500, rng=np.random.default_rng(0))
    assert isinstance(q, np.ndarray) and q.shape == (500, 7) and poses.shape == (500, 7)
    lim = np.array(O(robot).actuated_joints_limits)
    assert (q >= lim[:, 0]).all() and (q <= lim[:, 1]).all()
    ref = ko.forward_kinematics(robot, torch.from_numpy(q)).numpy()
    assert np.abs(poses[:, :3] - ref[:, :3]).max() <= 2e-6
    robot.use_approximate_collision_model()
    q2, _ = robot.sample_joint_angles_and_poses(2000, only_non_self_colliding=True, tqdm_enabled=False, return_torch=True,
                                                rng=np.random.default_rng(1))
    assert q2.is_cuda and not bool(robot.config_self_collides(q2).any())


def test_exact_ik_one_million_poses():
    """BASELINE config 5's size through the exact path on one GPU: 1,000,000 target poses, repeat_counts (1, 3, 10) with
    seeded random weights = the schedule's worst case (~14 M flow rows in 16384-row chunks, 42 M LM row-iterations, the
    multi-workgroup ordered compaction at every round, 280 MB of LM state sized once).  Size-independent properties."""
    robot, hp, lay, sd = panda_model()
    s = _solver(robot, hp, sd)
    eng = s.engine(DEV)
    n = 1_000_000
    g = torch.Generator(device=DEV).manual_seed(9)
    lo = torch.tensor([l[0] for l in O(robot).actuated_joints_limits], device=DEV)
    hi = torch.tensor([l[1] for l in O(robot).actuated_joints_limits], device=DEV)
    q = lo + (hi - lo) * torch.rand((n, 7), generator=g, device=DEV)
    poses = robot.forward_kinematics(q)
    sol, valid, stats = eng.generate_exact(poses, (1, 3, 10), 1e-3, 0.01, return_stats=True)
    assert sol.shape == (n, 7) and valid.shape == (n,) and valid.dtype == torch.bool
    assert int(stats[0, 0]) == n and int(stats[1, 0]) == n - int(stats[0, 3]) and int(stats[2, 0]) == int(stats[1, 0]) - int(stats[1, 3])
    assert int(valid.sum()) == int(stats[:, 3].sum()) and int(stats[:, 1].sum()) > 13_000_000
    assert bool((sol[~valid] == 0).all())  # unsolved rows stay 0 (ikflow_solver.py:197)
    if int(valid.sum()) > 0:
        pe, re = eng.pose_error(sol[valid], poses[valid])
        assert bool((pe < 1e-3).all()) and bool((re < 0.01).all())
        assert bool(((sol[valid] >= lo) & (sol[valid] <= hi)).all())
    # the solved poses, checked independently of the kernels that flagged them (oracle FK on the CPU)
    idx = torch.nonzero(valid)[:200, 0].cpu()
    if idx.numel():
        pe_ref, re_ref = ko.calculate_pose_error(robot, sol[idx.to(DEV)].cpu(), poses[idx.to(DEV)].cpu())
        assert (pe_ref < 1e-3 * 1.01).all() and (re_ref < 0.01 * 1.05).all()


def test_bench_two_ranks_on_one_gpu():
    """bench.py's N>1 path with the REAL engine: two ranks under torch.distributed.run, both on cuda:0, gloo in place of RCCL
    (IKF_BENCH_TEST_BACKEND; RCCL refuses two ranks on one device) - row shards, side-stream all_gather_into_tensor behind an event,
    fence, max over ranks, own shard at its rank offset (asserted inside bench.py), one JSON line from rank 0."""
    import json
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, IKF_BENCH_TEST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "1024"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = lines[0]
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2048 and out["scaling"] == "weak" and "test_backend" in out
    assert out["value"] > 0 and abs(out["value"] - 2048 * 4 / (out["ms_per_step"] * 4e-3)) <= 1e-6 * out["value"]
    assert "cpu_baseline" not in out and "cells" not in out["extra"]  # rank-0-at-N=1-only legs stay out of an N>1 line


def test_repeated_interleaved_calls_are_bit_identical():
    """tools/determinism_soak.py, short form: the same call repeated while other batch sizes run in between (the engine's buffers are
    shared between sizes) returns identical bits - approximate IK on every tile path in both precisions, and exact IK."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "determinism_soak.py"), "6"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_library_loaded_before_torch_still_sees_the_gpu():
    """`__graft_entry__.build()` then `smoke()` in ONE process: the ctypes binding is loaded before anything has imported torch.
    torch ships its own copy of the HIP runtime; `_lib.load()` imports torch first so the process holds one runtime, not two (with
    two, `ikf_create` saw no device although torch.cuda.is_available())."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("from ikflow_amd import _lib; _lib.load()\n"
            "import __graft_entry__ as g\n"
            "g.build(); g.smoke()\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
