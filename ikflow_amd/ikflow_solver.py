"""IKFlowSolver - drop-in for ``ikflow.ikflow_solver.IKFlowSolver`` (ikflow/ikflow_solver.py:32-441) whose compute runs on
the MI355X engine (libikflow_amd.so) instead of FrEIA / jrl / torch ops.

Same method names, argument meaning, return shapes/dtypes and assertion behaviour as the reference:
  generate_ik_solutions        ikflow_solver.py:254-343
  generate_exact_ik_solutions  ikflow_solver.py:345-411
  load_state_dict              ikflow_solver.py:413-441
  _run_inference / _calculate_pose_error   ikflow_solver.py:85-117
Differences, all documented in DESIGN.md:
  * there is no ``nn_model`` torch module - weights live packed in HBM inside the engine;
  * ``compile_model`` is accepted and ignored (nothing to trace); ``run_lma_on_cpu`` is accepted and ignored
    (LM runs on the GPU, resident with the batch);
  * ``generate_exact_ik_solutions`` takes an optional ``latents=`` (one tensor per retry round) so that
    "identical (pose, latent)" parity is testable - the reference draws them internally (:187);
  * ``return_detailed=True`` returns ``None`` in the self-collision slot (Klampt check, SURVEY 8 f-3, out of scope).
"""
from __future__ import annotations

import pickle
import warnings
from time import time
from typing import Dict, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from ikflow_amd import config
from ikflow_amd.model import (
    IkflowModelParameters,
    layout_from,
    random_state_dict,
    state_dict_to_numpy,
    validate_state_dict,
)
from ikflow_amd.robots import Robot


def mm_to_m(x: float) -> float:
    return x / 1000.0


def draw_latent(latent_distribution: str, latent_scale: float, shape: Tuple[int, int], device):
    """Draw a sample from the latent noise distribution (ikflow_solver.py:16-29; torch's global generator)."""
    assert latent_distribution in ["gaussian", "uniform"]
    assert latent_scale > 0
    assert len(shape) == 2
    if latent_distribution == "gaussian":
        return latent_scale * torch.randn(shape, device=device)
    if latent_distribution == "uniform":
        return 2 * latent_scale * torch.rand(shape, device=device) - latent_scale


class IKFlowSolver:
    def __init__(self, hyper_parameters: IkflowModelParameters, robot: Robot, compile_model: Optional[Dict] = None):
        """Initialize an IKFlowSolver (ikflow_solver.py:33-68)."""
        assert isinstance(
            hyper_parameters, IkflowModelParameters
        ), f"hyper_parameters should be a IkflowModelParameters type, is {type(hyper_parameters)}"
        assert isinstance(robot, Robot), f"robot should be a Robot type, is {type(robot)}"
        assert isinstance(compile_model, (type(None), dict))

        if not hasattr(hyper_parameters, "sigmoid_on_output"):
            hyper_parameters.sigmoid_on_output = False
        if hyper_parameters.softflow_enabled:
            assert not hyper_parameters.sigmoid_on_output, (
                "sigmoid_on_output and softflow are incompatible, disable one or the other"
            )
        self._robot = robot
        self.dim_cond = 7
        if hyper_parameters.softflow_enabled:
            self.dim_cond = 8  # [x, ... q3, softflow_scale]   (softflow_scale should be 0 for inference)
        self._network_width = hyper_parameters.dim_latent_space
        self._hyper_parameters = hyper_parameters
        self._layout = layout_from(hyper_parameters, robot)
        if compile_model is not None:
            warnings.warn("compile_model is ignored: the MI355X engine runs hand-written HIP kernels, nothing is traced.")
        self._model_weights_loaded = False
        self._engine = None  # created on first use, on the device of the inputs
        # tests / tools only: "probes" binds lib/libikflow_amd_probes.so (the product + the priced-and-rejected forms of rounds 2 - 3);
        # set before the first call
        self.library_flavour = ""
        self._precision = "f32"
        self._state_dict_np: Optional[Dict[str, np.ndarray]] = None
        self.ndof = self.robot.ndof

    # -- properties (ikflow_solver.py:68-83) ---------------------------------------------------------
    @property
    def robot(self) -> Robot:
        return self._robot

    @property
    def network_width(self) -> int:
        return self._network_width

    @property
    def conditional_size(self) -> int:
        return self.dim_cond

    @property
    def layout(self):
        return self._layout

    # -- engine --------------------------------------------------------------------------------------
    def engine(self, device=None):
        """The C-ABI handle for `device` (created lazily; weights are uploaded when it is created)."""
        from ikflow_amd.engine import Engine  # imports the ctypes binding: fails loudly without the built library

        from ikflow_amd.engine import _dev_index

        device = torch.device(config.DEVICE if device is None else device)
        # an index-less "cuda" means torch's current device (what Engine resolves it to)
        if self._engine is None or self._engine.device != torch.device("cuda", _dev_index(device)):
            eng = Engine(self._layout, self._robot, device, self.library_flavour)
            if self._state_dict_np is not None:
                eng.load_state_dict(self._state_dict_np)
            if self._precision != "f32":
                try:
                    eng.set_precision(self._precision)
                except Exception:
                    # the engine refused the mode for these weights (e.g. a hidden weight beyond the f16 range) and stays
                    # on f32: the solver follows it, so solver and engine never disagree, and reports the refusal once
                    self._precision = "f32"
                    self._engine = eng
                    raise
            self._engine = eng
        return self._engine

    def set_precision(self, mode: str):
        """Arithmetic of the hidden Linear contractions: "f32" (exact f32 MFMA, default) or "f16x3" (error-compensated
        f16 split, measured at least as accurate against fp64; see include/ikflow_amd.h ikf_set_precision)."""
        assert mode in ("f32", "f16x3"), mode
        if self._engine is not None:
            self._engine.set_precision(mode)  # EngineError when the mode is refused: the solver then keeps its previous mode
        self._precision = mode

    def _ensure_initialized(self, allow_uninitialized: bool):
        """The reference runs its randomly initialised nn_model when allow_uninitialized=True
        (tests/ikflow_solver_test.py:89-117); the engine needs explicit weights, so draw the same kind
        (nn.Linear default init) once."""
        if self._state_dict_np is None and allow_uninitialized:
            self._state_dict_np = random_state_dict(self._layout, self._robot, seed=0)
            self._engine = None

    # -- inner path (ikflow_solver.py:85-117) ----------------------------------------------------------
    def _run_inference(self, latent: torch.Tensor, y: torch.Tensor, t0: float, clamp_to_joint_limits: bool, return_detailed: bool):
        t0 = time()
        eng = self.engine(latent.device)
        solutions = eng.generate_approx(y, latent, clamp_to_joint_limits)
        if return_detailed:
            n = solutions.shape[0]
            targets = y.reshape(1, 7).expand(n, 7).contiguous() if y.numel() == 7 else y
            pos_errors, rot_errors = eng.pose_error(solutions, targets)
            joint_limits_exceeded = eng.joint_limits_exceeded(solutions)
            # evaluation_utils.evaluate_solutions' fourth slot: filled when the robot carries a capsule model (jrl's
            # collision geometry is not available here), None otherwise
            self_colliding = self._robot.config_self_collides(solutions) if self._robot.has_collision_model else None
            return solutions, pos_errors, rot_errors, joint_limits_exceeded, self_colliding, time() - t0
        return solutions

    def _calculate_pose_error(self, qs: torch.Tensor, target_poses: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.engine(qs.device).pose_error(qs, target_poses)

    # -- public methods ----------------------------------------------------------------------------------
    def generate_ik_solutions(
        self,
        y: torch.Tensor,
        n: Optional[int] = None,
        latent: Optional[torch.Tensor] = None,
        latent_distribution: str = "gaussian",
        latent_scale: float = 1.0,
        clamp_to_joint_limits: bool = True,
        refine_solutions: bool = False,
        return_detailed: bool = False,
        allow_uninitialized: bool = False,
    ):
        """Run the flow in reverse to generate samples conditioned on a pose y (ikflow_solver.py:254-343).

        y: [7] single target pose (then `n` solutions are drawn) or [batch x 7]; latent: optional [n x network_width].
        Returns [n x ndof], or the 6-tuple (solutions, pos_errors, rot_errors, joint_limits_exceeded, self_colliding,
        runtime) when return_detailed.
        """
        t0 = time()
        if not allow_uninitialized:
            assert self._model_weights_loaded, "Model weights have not been loaded. Call load_state_dict(...)"
        assert isinstance(y, torch.Tensor), f"y must be a torch.Tensor (got {type(y)})."
        if y.numel() == 7:
            assert isinstance(n, int)
            assert n > 0
        else:
            assert y.shape[1] == 7, f"y must be of shape [7] or [n x 7], got {y.shape}"
        assert isinstance(latent_distribution, str)
        assert isinstance(latent_scale, float)
        assert isinstance(latent, torch.Tensor) or (
            latent is None
        ), f"latent must either be a torch.Tensor or None (got {type(latent)})."
        assert not refine_solutions, "refine_solutions is deprecated, use generate_exact_ik_solutions() instead"
        if "cuda" in str(config.DEVICE):
            assert "cpu" not in str(y.device), f"Cuda is available ('{config.DEVICE}'), but target_poses are on {y.device}"
        self._ensure_initialized(allow_uninitialized)

        n = y.shape[0] if n is None else n
        device = y.device
        with torch.inference_mode():
            # the conditional [y, 0] is assembled inside the first-layer kernel (ikflow_solver.py:333-338)
            if latent is None:
                latent = draw_latent(latent_distribution, latent_scale, (n, self._network_width), device)
            assert latent.shape[0] == n, f"{len(latent)} != {n}"
            return self._run_inference(latent, y, t0, clamp_to_joint_limits, return_detailed)

    def generate_exact_ik_solutions(
        self,
        target_poses: torch.Tensor,
        repeat_counts: Tuple[int] = (1, 3, 10),
        pos_error_threshold: float = mm_to_m(1),
        rot_error_threshold: float = 0.1,
        verbosity: int = 0,
        run_lma_on_cpu: bool = True,
        return_detailed: bool = False,
        latents: Optional[Sequence[torch.Tensor]] = None,
    ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Same as generate_ik_solutions() but refines the flow's seeds with Levenberg-Marquardt
        (ikflow_solver.py:345-411). Returns (solutions [n x ndof], valids [n] bool); unsolved rows are 0."""
        assert target_poses.shape[1] == 7, f"target_poses must be of shape [n x 7], got {target_poses.shape}"
        assert isinstance(repeat_counts, tuple), f"repeat_counts must be a tuple, got {type(repeat_counts)}"
        assert not return_detailed, "return_detailed is not currently supported for generate_exact_ik_solutions()"
        assert self._model_weights_loaded, "Model weights have not been loaded. Call load_state_dict(...)"
        t0 = time()
        n_opt_steps_max = 3
        with torch.inference_mode():
            eng = self.engine(target_poses.device)
            out = eng.generate_exact(
                target_poses, repeat_counts, pos_error_threshold, rot_error_threshold, latents=latents,
                n_lm_steps=n_opt_steps_max, return_stats=verbosity > 0,
            )
        if verbosity > 0:
            solutions, valids, stats = out
            for r, (n_in, rows, lm_rows, solved) in enumerate(stats.tolist()):
                print(f"  round {r}: repeat={repeat_counts[r]} poses={n_in} flow_rows={rows} lm_row_iters<={lm_rows} solved={solved}")
            print(f"  {int(valids.sum().item())}/{valids.numel()} valid ({time() - t0} seconds)")
            return solutions, valids
        return out

    # -- weights -------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict_filename: str):
        """Set the model's weights from a pickled state_dict (ikflow_solver.py:413-441) or a .npz with the same keys."""
        if str(state_dict_filename).endswith(".npz"):
            with np.load(state_dict_filename) as z:
                state_dict = {k: z[k] for k in z.files}
        else:
            with open(state_dict_filename, "rb") as f:
                try:
                    state_dict = pickle.load(f)
                except pickle.UnpicklingError as e:
                    print(f"Error loading state dict from {state_dict_filename}: {e}")
                    raise e
        self.load_state_dict_tensors(state_dict)

    def load_state_dict_tensors(self, state_dict: Dict[str, Union[torch.Tensor, np.ndarray]]):
        """Extension: take the {key: tensor} mapping directly (what pickle.load returns in the reference)."""
        sd = state_dict_to_numpy(state_dict)
        sd = {(k[len("nn_model."):] if k.startswith("nn_model.") else k): v for k, v in sd.items()}
        sd = {(k[len("_orig_mod."):] if k.startswith("_orig_mod.") else k): v for k, v in sd.items()}  # torch.compile'd modules (:421-427)
        # older FrEIA releases named the two coupling subnets s1/s2 instead of subnet1/subnet2
        sd = {k.replace(".s1.", ".subnet1.").replace(".s2.", ".subnet2."): v for k, v in sd.items()}
        validate_state_dict(self._layout, sd)  # RuntimeError like nn.Module.load_state_dict on a bad file
        self._state_dict_np = sd
        if self._engine is not None:
            try:
                self._engine.load_state_dict(sd)
            finally:  # an f16x3 handle refuses out-of-range weights by falling back to f32: the solver follows its engine
                self._precision = self._engine.precision
        self._model_weights_loaded = True
