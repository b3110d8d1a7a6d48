"""Python face of the C-ABI engine: owns an ``ikf_model`` handle, passes torch device pointers + the current HIP
stream through ctypes.  torch is used here only for device memory and streams."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ikflow_amd import _lib
from ikflow_amd.model import FlowLayout, LEAKY_RELU_SLOPE
from ikflow_amd.robots import JOINT_FIXED, Robot, rpy_to_matrix


class EngineError(RuntimeError):
    pass


def _raise(code: int, lib=None):
    msg = _lib.last_error(lib)
    if code == _lib.IKF_ERR_NOT_LOADED:
        raise AssertionError(msg)  # the reference asserts (ikflow_solver.py:310-311)
    if code == _lib.IKF_ERR_MISSING_TENSOR:
        raise RuntimeError("Error(s) in loading state_dict: " + msg)
    raise EngineError(f"libikflow_amd status {code}: {msg}")


def _check(code: int, lib=None):
    if code != _lib.IKF_OK:
        _raise(code, lib)


def fold_chain(robot: Robot) -> Tuple[List[Tuple[int, np.ndarray, np.ndarray]], np.ndarray]:
    """Fold the fixed URDF transforms into the actuated joints: [(kind, axis, pre 3x4)], tool 3x4 (float64 math)."""
    joints = []
    T = np.eye(4)
    for j in robot.joints:
        F = np.eye(4)
        F[:3, :3] = rpy_to_matrix(j.origin_rpy)
        F[:3, 3] = j.origin_xyz
        T = T @ F
        if j.kind == JOINT_FIXED:
            continue
        ax = np.asarray(j.axis, dtype=np.float64)
        ax = ax / np.linalg.norm(ax)
        joints.append((int(j.kind), ax, T[:3, :4].copy()))
        T = np.eye(4)
    return joints, T[:3, :4].copy()


def _make_desc(layout: FlowLayout, robot: Robot) -> _lib.ikf_model_desc:
    d = _lib.ikf_model_desc()
    d.abi_version = _lib.IKF_ABI_VERSION
    d.nb_nodes, d.dim, d.dim_cond = layout.nb_nodes, layout.dim, layout.dim_cond
    d.width, d.n_hidden = layout.width, layout.n_hidden
    d.clamp, d.leaky_slope = layout.clamp, LEAKY_RELU_SLOPE
    d.ndof = robot.ndof
    d.sigmoid_on_output = 1 if getattr(layout, "sigmoid_on_output", False) else 0
    if robot.ndof > _lib.IKF_MAX_DOF:
        raise EngineError(f"robot has {robot.ndof} dof; the engine supports at most {_lib.IKF_MAX_DOF}")
    for i, (lo, hi) in enumerate(robot.actuated_joints_limits):
        d.joint_lo[i], d.joint_hi[i] = lo, hi
    joints, tool = fold_chain(robot)
    for i, (kind, ax, pre) in enumerate(joints):
        d.chain[i].kind = kind
        for k in range(3):
            d.chain[i].axis[k] = float(ax[k])
        for k, v in enumerate(pre.reshape(-1)):
            d.chain[i].pre[k] = float(v)
    for k, v in enumerate(tool.reshape(-1)):
        d.tool[k] = float(v)
    return d


def _dev_index(device) -> int:
    dev = torch.device(device)
    if dev.type != "cuda":
        raise EngineError(f"ikflow_amd runs on the GPU only; got device '{dev}' (there is no CPU path)")
    return dev.index if dev.index is not None else torch.cuda.current_device()


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor (got {type(t)})")
    if t.device.type != "cuda":
        raise EngineError(f"{name} is on '{t.device}'; ikflow_amd runs on the GPU only (there is no CPU path)")
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t.contiguous()


class Engine:
    """One ikf_model handle on one device."""

    def __init__(self, layout: FlowLayout, robot: Robot, device, flavour: str = ""):
        self.lib = _lib.load(flavour)
        self.flavour = flavour
        self.layout = layout
        self.robot = robot
        self.device = torch.device("cuda", _dev_index(device))
        self._h = C.c_void_p()
        desc = _make_desc(layout, robot)
        self._ck(self.lib.ikf_create(C.byref(desc), self.device.index, C.byref(self._h)))

    def _ck(self, code: int) -> None:
        _check(code, self.lib)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self.lib.ikf_destroy(h)
            except Exception:
                pass
            self._h = C.c_void_p()

    # -- helpers -------------------------------------------------------------------------------------
    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _on_device(self, t: torch.Tensor, name: str) -> torch.Tensor:
        t = _f32(t, name)
        if t.device != self.device:
            raise EngineError(f"{name} is on {t.device} but the engine lives on {self.device}")
        return t

    # -- weights -------------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, np.ndarray]) -> None:
        """sd: {FrEIA key: numpy array (float32 / int64)} - see ikflow_amd.model for the key names."""
        items = list(sd.items())
        arr = (_lib.ikf_tensor * len(items))()
        keep = []
        for i, (k, v) in enumerate(items):
            v = np.ascontiguousarray(v)
            if v.dtype.kind == "f":
                v = v.astype(np.float32, copy=False)
                dt = 0
            elif v.dtype.kind in "iu":
                v = v.astype(np.int64, copy=False)
                dt = 1
            else:
                continue
            if v.ndim > 4:
                continue
            name = k.encode("utf-8")
            keep.append((name, v))
            arr[i].name = name
            arr[i].h_data = v.ctypes.data
            arr[i].dtype = dt
            arr[i].ndim = v.ndim
            for a, s in enumerate(v.shape):
                arr[i].shape[a] = s
        with torch.cuda.device(self.device):
            self._ck(self.lib.ikf_load_weights(self._h, arr, len(items)))
        del keep

    @property
    def weights_loaded(self) -> bool:
        return bool(self.lib.ikf_weights_loaded(self._h))

    def reserve(self, max_rows: int) -> None:
        self._ck(self.lib.ikf_reserve(self._h, int(max_rows)))

    def reserve_exact(self, max_poses: int, max_repeat: int = 10) -> None:
        """Pre-size the exact-IK state (max_poses * max_repeat LM rows) so that generate_exact allocates nothing."""
        self._ck(self.lib.ikf_reserve_exact(self._h, int(max_poses), int(max_repeat)))

    def set_exact_upfront_rows(self, max_rows: int) -> None:
        """Largest worst-case exact-IK row state a call may reserve up front (0: always grow per retry round)."""
        self._ck(self.lib.ikf_set_exact_upfront_rows(self._h, int(max_rows)))

    PRECISIONS = {"f32": 0, "f16x3": 1}

    def set_precision(self, mode: str) -> None:
        """"f32": hidden contractions on the exact-f32 MFMA. "f16x3": error-compensated three-product f16 split."""
        self._ck(self.lib.ikf_set_precision(self._h, self.PRECISIONS[mode]))

    @property
    def precision(self) -> str:
        return {v: k for k, v in self.PRECISIONS.items()}[self.lib.ikf_get_precision(self._h)]

    LM_PRECISIONS = {"f32": 0, "f64": 1}

    def set_lm_precision(self, mode: str) -> None:
        """Arithmetic of the LM step (include/ikflow_amd.h ikf_set_lm_precision): "f64" (default) = fp64 inside the step, Cholesky; "f32" =
        the reference's own arithmetic (fp32 throughout, LU with partial pivoting as torch.linalg.solve, ikflow_solver.py:205,208)."""
        self._ck(self.lib.ikf_set_lm_precision(self._h, self.LM_PRECISIONS[mode]))

    @property
    def lm_precision(self) -> str:
        return {v: k for k, v in self.LM_PRECISIONS.items()}[self.lib.ikf_get_lm_precision(self._h)]

    def set_split_guard(self, on: bool) -> None:
        """f16x3 range guard (include/ikflow_amd.h ikf_set_split_guard): on (default) = one 4-byte flag read per call and an
        automatic f32 re-run when a hidden activation left the f16 range; off = no synchronisation, no re-run."""
        self._ck(self.lib.ikf_set_split_guard(self._h, 1 if on else 0))

    @property
    def split_fallback_count(self) -> int:
        return int(self.lib.ikf_split_fallback_count(self._h))

    def split_overflow_pending(self) -> bool:
        return bool(self.lib.ikf_split_overflow_pending(self._h, self._stream()))

    def set_gemm_variant(self, variant: int) -> None:
        self._ck(self.lib.ikf_set_gemm_variant(self._h, int(variant)))

    # -- approximate IK ------------------------------------------------------------------------------
    def generate_approx(self, poses: torch.Tensor, latent: torch.Tensor, clamp: bool, softflow_scale: float = 0.0) -> torch.Tensor:
        """poses [n x 7] or [7] (broadcast); latent [n x D] -> [n x ndof]."""
        latent = self._on_device(latent, "latent")
        poses = self._on_device(poses, "y")
        n = latent.shape[0]
        assert latent.ndim == 2 and latent.shape[1] == self.layout.dim, f"latent must be [n x {self.layout.dim}], got {tuple(latent.shape)}"
        broadcast = poses.numel() == 7
        if not broadcast:
            assert poses.ndim == 2 and poses.shape[1] == 7 and poses.shape[0] == n, f"{poses.shape[0]} != {n}"
        out = torch.empty((n, self.layout.ndof), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._ck(
                self.lib.ikf_generate_approx(
                    self._h, poses.data_ptr(), 1 if broadcast else 0, latent.data_ptr(), n, 1 if clamp else 0,
                    float(softflow_scale), out.data_ptr(), self._stream(),
                )
            )
        return out

    # -- kinematics ----------------------------------------------------------------------------------
    def _q(self, q: torch.Tensor) -> torch.Tensor:
        q = self._on_device(q, "q")
        assert q.ndim == 2 and q.shape[1] == self.layout.ndof, f"q must be [n x {self.layout.ndof}], got {tuple(q.shape)}"
        return q

    def forward_kinematics(self, q: torch.Tensor) -> torch.Tensor:
        q = self._q(q)
        out = torch.empty((q.shape[0], 7), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib.ikf_forward_kinematics(self._h, q.data_ptr(), q.shape[0], out.data_ptr(), self._stream()))
        return out

    def pose_error(self, q: torch.Tensor, target_poses: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        q = self._q(q)
        tp = self._on_device(target_poses, "target_poses")
        assert tp.shape == (q.shape[0], 7)
        pe = torch.empty(q.shape[0], dtype=torch.float32, device=self.device)
        re = torch.empty_like(pe)
        with torch.cuda.device(self.device):
            self._ck(self.lib.ikf_pose_error(self._h, q.data_ptr(), tp.data_ptr(), q.shape[0], pe.data_ptr(), re.data_ptr(), self._stream()))
        return pe, re

    def lm_step(self, target_poses: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
        q = self._q(q)
        tp = self._on_device(target_poses, "target_poses")
        assert tp.shape == (q.shape[0], 7)
        out = torch.empty_like(q)
        with torch.cuda.device(self.device):
            self._ck(self.lib.ikf_lm_step(self._h, tp.data_ptr(), q.data_ptr(), q.shape[0], out.data_ptr(), self._stream()))
        return out

    def jacobian(self, q: torch.Tensor) -> torch.Tensor:
        q = self._q(q)
        out = torch.empty((q.shape[0], 6, self.layout.ndof), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib.ikf_jacobian(self._h, q.data_ptr(), q.shape[0], out.data_ptr(), self._stream()))
        return out

    def clamp_to_joint_limits(self, q: torch.Tensor) -> torch.Tensor:
        q = self._q(q)
        out = torch.empty_like(q)
        with torch.cuda.device(self.device):
            self._ck(self.lib.ikf_clamp_to_joint_limits(self._h, q.data_ptr(), q.shape[0], out.data_ptr(), self._stream()))
        return out

    def joint_limits_exceeded(self, q: torch.Tensor) -> torch.Tensor:
        q = self._q(q)
        out = torch.empty(q.shape[0], dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib.ikf_joint_limits_exceeded(self._h, q.data_ptr(), q.shape[0], out.data_ptr(), self._stream()))
        return out.to(torch.bool)

    # -- capsule self-collision (evaluation_utils.calculate_self_collisions mechanism) ------------------------
    def set_collision_model(self, capsules, pairs) -> None:
        """capsules: [(frame, p0[3], p1[3], radius)] in engine frames (0 = base, j + 1 = after actuated joint j);
        pairs: [(a, b)] capsule indices."""
        arr = (_lib.ikf_capsule * max(len(capsules), 1))()
        for c, (frame, p0, p1, r) in zip(arr, capsules):
            c.frame, c.radius = int(frame), float(r)
            for k in range(3):
                c.p0[k], c.p1[k] = float(p0[k]), float(p1[k])
        flat = (C.c_int32 * max(2 * len(pairs), 1))(*[int(v) for ab in pairs for v in ab])
        self._ck(self.lib.ikf_set_collision_model(self._h, C.cast(arr, C.c_void_p), len(capsules), C.cast(flat, C.c_void_p), len(pairs)))
        self._has_collision_model = True

    def self_collision(self, q: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """[n x ndof] -> (signed clearance of the closest pair [n] f32, colliding [n] bool)."""
        q = self._q(q)
        dist = torch.empty(q.shape[0], dtype=torch.float32, device=self.device)
        col = torch.empty(q.shape[0], dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib.ikf_self_collision(self._h, q.data_ptr(), q.shape[0], dist.data_ptr(), col.data_ptr(), self._stream()))
        return dist, col.to(torch.bool)

    # -- exact IK ------------------------------------------------------------------------------------
    def generate_exact(
        self,
        target_poses: torch.Tensor,
        repeat_counts: Sequence[int],
        pos_error_threshold: float,
        rot_error_threshold: float,
        latents: Optional[Sequence[torch.Tensor]] = None,
        n_lm_steps: int = 3,
        return_stats: bool = False,
        seed_fn=None,
    ):
        """Returns (solutions [n x ndof] f32, valids [n] bool) (+ stats [rounds x 4] if asked).

        ``latents``: optional per-round latent tensors ([>= n_r*R_r x D], tile-major) for parity runs; when None each
        round draws ``torch.randn((n_tiled, D), device=...)`` exactly like draw_latent() (ikflow_solver.py:187).
        ``seed_fn(round, active_idx [n_active] int64 device tensor, repeat) -> [n_active * repeat x ndof]``: parity runs with
        the flow taken out (ikf_generate_exact_seeded) - the returned seeds replace the flow's output of that round."""
        tp = self._on_device(target_poses, "target_poses")
        assert tp.ndim == 2 and tp.shape[1] == 7, f"target_poses must be of shape [n x 7], got {tuple(tp.shape)}"
        n = tp.shape[0]
        D = self.layout.dim
        nr = len(repeat_counts)
        sols = torch.empty((n, self.layout.ndof), dtype=torch.float32, device=self.device)
        valid = torch.empty(n, dtype=torch.uint8, device=self.device)
        keep: List[torch.Tensor] = []
        err: List[BaseException] = []

        def _latent_cb(_user, rnd, rows, dim):
            try:
                if latents is not None:
                    lt = self._on_device(latents[rnd], f"latents[{rnd}]")
                    assert lt.ndim == 2 and lt.shape[1] == dim and lt.shape[0] >= rows, (
                        f"latents[{rnd}] must be at least [{rows} x {dim}], got {tuple(lt.shape)}"
                    )
                else:
                    lt = 1.0 * torch.randn((rows, dim), device=self.device)
                keep.append(lt)
                return lt.data_ptr()
            except BaseException as e:  # never let an exception cross the C boundary
                err.append(e)
                return 0

        def _seed_cb(_user, rnd, n_active, repeat, _d_idx, ndof):
            try:
                # the still-unsolved poses, ascending: the same list the engine built (its stream is idle here)
                idx = torch.arange(n, device=self.device) if rnd == 0 else torch.nonzero(valid == 0)[:, 0]
                assert idx.numel() == n_active, (idx.numel(), n_active)
                sq = self._on_device(seed_fn(rnd, idx, repeat), f"seeds of round {rnd}")
                assert sq.shape == (n_active * repeat, ndof), f"seeds of round {rnd} must be [{n_active * repeat} x {ndof}], got {tuple(sq.shape)}"
                keep.append(sq)
                return sq.data_ptr()  # produced on the stream the engine enqueues its copy on: ordered
            except BaseException as e:
                err.append(e)
                return 0

        rc = (C.c_int32 * nr)(*[int(r) for r in repeat_counts])
        stats = (C.c_int64 * (4 * nr))()
        with torch.cuda.device(self.device):
            if seed_fn is not None:
                cb = _lib.SEED_FN(_seed_cb)
                code = self.lib.ikf_generate_exact_seeded(
                    self._h, tp.data_ptr(), n, rc, nr, int(n_lm_steps), float(pos_error_threshold),
                    float(rot_error_threshold), cb, None, sols.data_ptr(), valid.data_ptr(),
                    stats if return_stats else None, self._stream(),
                )
            else:
                cb = _lib.LATENT_FN(_latent_cb)
                code = self.lib.ikf_generate_exact(
                    self._h, tp.data_ptr(), n, rc, nr, int(n_lm_steps), float(pos_error_threshold),
                    float(rot_error_threshold), cb, None, sols.data_ptr(), valid.data_ptr(),
                    stats if return_stats else None, self._stream(),
                )
        if err:
            raise err[0]
        self._ck(code)
        # latents must outlive the enqueued kernels
        if keep:
            torch.cuda.current_stream(self.device).synchronize()
        out = (sols, valid.to(torch.bool))
        if return_stats:
            return out + (np.array(list(stats), dtype=np.int64).reshape(nr, 4),)
        return out

    def refine_exact(self, target_poses: torch.Tensor, seeds_q: torch.Tensor, repeat: int, pos_error_threshold: float,
                     rot_error_threshold: float, n_lm_steps: int = 3) -> Tuple[torch.Tensor, torch.Tensor]:
        """One round of _generate_exact_ik_solutions (ikflow_solver.py:119-247) given its flow output: seeds_q
        [n * repeat x ndof] tile-major -> (solutions [n x ndof], valids [n] bool)."""
        tp = self._on_device(target_poses, "target_poses")
        sq = self._q(seeds_q)
        n = tp.shape[0]
        assert tp.ndim == 2 and tp.shape[1] == 7 and sq.shape[0] == n * repeat, (tuple(tp.shape), tuple(sq.shape), repeat)
        sols = torch.empty((n, self.layout.ndof), dtype=torch.float32, device=self.device)
        valid = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._ck(self.lib.ikf_refine_exact(self._h, tp.data_ptr(), n, int(repeat), sq.data_ptr(), int(n_lm_steps),
                                         float(pos_error_threshold), float(rot_error_threshold), sols.data_ptr(),
                                         valid.data_ptr(), self._stream()))
        return sols, valid.to(torch.bool)

    # -- measurement ---------------------------------------------------------------------------------
    def time_gemm(self, rows: int, iters: int) -> float:
        ms = C.c_float(0.0)
        with torch.cuda.device(self.device):
            self._ck(self.lib.ikf_time_gemm(self._h, int(rows), int(iters), C.byref(ms), self._stream()))
        return float(ms.value)

    def profile_begin(self) -> None:
        self._ck(self.lib.ikf_profile_begin(self._h))

    def profile_end(self):
        """-> (number of dominant-kernel launches since profile_begin, sum of their HIP-event durations in ms)"""
        n, ms = C.c_int64(0), C.c_double(0.0)
        with torch.cuda.device(self.device):
            self._ck(self.lib.ikf_profile_end(self._h, C.byref(n), C.byref(ms), self._stream()))
        self.last_event_overhead_ms = float(self.lib.ikf_profile_event_overhead_ms(self._h))
        return int(n.value), float(ms.value)

    def plan(self, rows: int) -> str:
        """The chunks a call of `rows` rows is cut into, e.g. "rowowner:4096 cluster16:200" (DESIGN.md section 4.3)."""
        buf = C.create_string_buffer(256)
        self._ck(self.lib.ikf_plan_describe(self._h, int(rows), buf, 256))
        return buf.value.decode()

    @property
    def cluster_local(self) -> bool:
        """Cluster launches with 4 / 8 / 16 members hand over through one XCD's L2 (placement census at load + a check in every launch)."""
        return bool(self.lib.ikf_cluster_local(self._h))

    @property
    def cluster_repairs(self) -> int:
        """Calls whose cluster-form launch gave up waiting for a peer workgroup (recomputed by the repair launch; the form then sits out
        `cluster_backoff` calls and is tried again)."""
        return int(self.lib.ikf_cluster_repairs(self._h))

    @property
    def cluster_backoff(self) -> int:
        """Calls the cluster form still sits out after a give-up (16 after the first, doubling up to 65536; 0: in use)."""
        return int(self.lib.ikf_cluster_backoff(self._h))

    @property
    def load_time_ms(self) -> float:
        """Host wall time of the last ikf_load_weights on this handle (packing, upload, device-side images of the resident-row forms)."""
        return float(self.lib.ikf_load_time_ms(self._h))

    @property
    def frag_image_time_ms(self) -> float:
        """... of building the small-batch per-layer kernels' weight image (0 until a <= 512-row chunk on that path or ikf_reserve needed it)."""
        return float(self.lib.ikf_frag_image_time_ms(self._h))

    def dominant_kernel_name(self, rows: Optional[int] = None) -> str:
        """Name (as in a rocprofv3 kernel trace) of the kernel that carries a batch of `rows` rows; None: the per-layer contraction."""
        if rows is not None:
            return self.lib.ikf_dominant_kernel_for(self._h, int(rows)).decode()
        if self.precision == "f16x3":
            return self.lib.ikf_split_kernel_name().decode()
        return self.lib.ikf_dominant_kernel_name().decode()


# ---------------------------------------------------------------------------------------------------
# kinematics-only engines for Robot.forward_kinematics & co (no flow weights needed)
# ---------------------------------------------------------------------------------------------------
_KIN_CACHE: Dict[Tuple, Engine] = {}


def _robot_key(robot: Robot) -> Tuple:
    """Two robots share an engine only when their chains and limits are the same, whatever their names."""
    return (robot.name, tuple((j.kind, j.origin_xyz, j.origin_rpy, j.axis, j.limits) for j in robot.joints))


def kinematics_engine_for(robot: Robot, device=None) -> Engine:
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if not torch.cuda.is_available():
        raise EngineError("no GPU visible: ikflow_amd kinematics run on the MI355X only (there is no CPU path)")
    key = (_robot_key(robot), _dev_index(dev))
    if key not in _KIN_CACHE:
        lay = FlowLayout(nb_nodes=1, dim=max(robot.ndof, 2), dim_cond=8, width=256, n_hidden=1, clamp=2.5, ndof=robot.ndof)
        _KIN_CACHE[key] = Engine(lay, robot, dev)
    return _KIN_CACHE[key]
