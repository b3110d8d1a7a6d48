"""Flow description for the hot path: hyper-parameters, the conditional-flow layout, weight tables.

Mirrors what ``ikflow/model.py`` defines for inference:
  * ``IkflowModelParameters``            ikflow/model.py:17-41   (attribute bag; inference reads 7 fields)
  * ``TINY_MODEL_PARAMS``                ikflow/model.py:45-48
  * graph layout of ``glow_cNF_model``   ikflow/model.py:291-356 :
        FixedLinearTransform -> [PermuteRandom(seed=i) -> GLOWCouplingBlock(clamp, split=D//2)] x nb_nodes
  * subnet layout ``subnet_constructor`` ikflow/model.py:51-96   : Linear/LeakyReLU(0.01) stacks
  * released-model table                 ikflow/model_descriptions.yaml (hyper-parameters only; the weight
    URLs are not reachable from here and are not part of the hot path)

There is no torch.nn graph here: the "model" is a table of named fp32 arrays (the FrEIA ``GraphINN``
state_dict key names) that the engine packs once into its HBM layout.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

from ikflow_amd.robots import Robot

SIGMOID_SCALING_ABS_MAX = 1.0  # ikflow/config.py:31
LEAKY_RELU_SLOPE = 0.01  # torch.nn.LeakyReLU() default, ikflow/model.py:63-95
ATAN_CLAMP_GAIN = 0.636  # FrEIA GLOWCouplingBlock clamp_activation="ATAN": s = clamp * 0.636 * atan(s)


class IkflowModelParameters:
    """Same attribute bag as ikflow/model.py:17-41 (defaults identical)."""

    def __init__(self):
        self.coupling_layer = "glow"
        self.nb_nodes = 12
        self.dim_latent_space = 9
        self.coeff_fn_config = 3
        self.coeff_fn_internal_size = 1024
        self.permute_random_enabled = True
        self.sigmoid_on_output = False
        self.lambd_predict = 1.0
        self.init_scale = 0.04473500291638653
        self.rnvp_clamp = 2.5
        self.y_noise_scale = 1e-7
        self.zeros_noise_scale = 1e-3
        self.softflow_noise_scale = 0.01
        self.softflow_enabled = True

    def __str__(self) -> str:
        s = "IkflowModelParameters\n"
        for k, v in self.__dict__.items():
            s += f"  {k}: \t{v}\n"
        return s


def _tiny() -> IkflowModelParameters:
    p = IkflowModelParameters()
    p.nb_nodes = 3
    p.coeff_fn_config = 2
    p.coeff_fn_internal_size = 256
    return p


TINY_MODEL_PARAMS = _tiny()

# Hyper-parameters of the released models (ikflow/model_descriptions.yaml:10-17,19-26,33-49,56-63).
MODEL_DESCRIPTIONS: Dict[str, Dict] = {
    "panda__full__lp191_5.25m": dict(nb_nodes=12, dim_latent_space=7, coeff_fn_config=3, coeff_fn_internal_size=1024, rnvp_clamp=2.5, robot_name="panda"),
    "panda_lite_tpm": dict(nb_nodes=6, dim_latent_space=7, coeff_fn_config=3, coeff_fn_internal_size=1024, rnvp_clamp=2.5, robot_name="panda"),
    "fetch_full_temp_nsc_tpm": dict(nb_nodes=12, dim_latent_space=8, coeff_fn_config=3, coeff_fn_internal_size=1024, rnvp_clamp=2.5, robot_name="fetch"),
    "fetch__large__ns183_9.75m": dict(nb_nodes=16, dim_latent_space=8, coeff_fn_config=3, coeff_fn_internal_size=1024, rnvp_clamp=2.5, robot_name="fetch"),
    "fetch_arm__large__mh186_9.25m": dict(nb_nodes=16, dim_latent_space=10, coeff_fn_config=3, coeff_fn_internal_size=1024, rnvp_clamp=2.5, robot_name="fetch_arm"),
    # ikflow/model_descriptions.yaml:90-97.  The robot itself is data: Robot.from_urdf / register_robot_urdf (robots.py)
    "rizon4__snowy-brook-208__global_step=2.75M": dict(nb_nodes=12, dim_latent_space=7, coeff_fn_config=3, coeff_fn_internal_size=1024, rnvp_clamp=2.5, robot_name="rizon4"),
}


def hparams_for(model_name: str) -> IkflowModelParameters:
    assert model_name in MODEL_DESCRIPTIONS, f"Model name '{model_name}' not found in model descriptions"
    p = IkflowModelParameters()
    p.__dict__.update(MODEL_DESCRIPTIONS[model_name])
    return p


@dataclass(frozen=True)
class FlowLayout:
    """Static shape of the conditional flow (everything the engine needs besides the weights)."""

    nb_nodes: int
    dim: int  # D = dim_latent_space (network width)
    dim_cond: int  # 8 with softflow (pose 7 + softflow scale), else 7   ikflow_solver.py:51-53
    width: int  # coeff_fn_internal_size
    n_hidden: int  # coeff_fn_config = number of width x width-or-in x width Linear layers before the output layer
    clamp: float
    ndof: int
    sigmoid_on_output: bool = False  # ikflow/model.py:304-307 graph variant: scaling node + flipped sigmoid in front

    @property
    def module_offset(self) -> int:
        """Index shift of the permutation / coupling modules in GraphINN.module_list: the sigmoid variant has two
        leading modules (IkFlowFixedLinearTransform, InvertibleSigmoidFlipped) instead of one."""
        return 1 if self.sigmoid_on_output else 0

    @property
    def split1(self) -> int:  # ikflow/model.py:336  (old FrEIA rule: D // 2)
        return self.dim // 2

    @property
    def split2(self) -> int:
        return self.dim - self.dim // 2

    def subnet_dims(self, which: int) -> List[Tuple[int, int]]:
        """[(in, out)] per Linear layer of subnet 1 or 2 (FrEIA GLOWCouplingBlock):
        subnet1: (split1 + cond) -> 2*split2 ; subnet2: (split2 + cond) -> 2*split1."""
        cin = (self.split1 if which == 1 else self.split2) + self.dim_cond
        cout = 2 * (self.split2 if which == 1 else self.split1)
        dims = [(cin, self.width)]
        for _ in range(self.n_hidden - 1):
            dims.append((self.width, self.width))
        dims.append((self.width, cout))
        return dims

    def n_weights(self) -> int:
        return self.nb_nodes * sum(i * o for w in (1, 2) for (i, o) in self.subnet_dims(w))

    def flops_per_solution(self) -> int:
        """2 * sum(in*out) over every Linear (Panda 101,572,608; FetchArm 135,725,056)."""
        return 2 * self.n_weights()

    def weight_bytes(self) -> int:
        """fp32 weights + biases, read once per batch (Panda 203,440,800)."""
        nb = self.nb_nodes * sum(o for w in (1, 2) for (_, o) in self.subnet_dims(w))
        return 4 * (self.n_weights() + nb)

    def row_io_bytes(self) -> int:
        return 4 * (7 + self.dim + self.ndof)


def layout_from(hparams: IkflowModelParameters, robot: Robot) -> FlowLayout:
    if not hasattr(hparams, "sigmoid_on_output"):
        hparams.sigmoid_on_output = False  # ikflow_solver.py:43-44
    if hparams.softflow_enabled:
        assert not hparams.sigmoid_on_output, "sigmoid_on_output and softflow are incompatible, disable one or the other"
    assert hparams.coeff_fn_config in (1, 2, 3, 4), "Number of layers `n_layers` must be in [1, ..., 4]"
    dim_cond = 8 if hparams.softflow_enabled else 7
    return FlowLayout(
        nb_nodes=int(hparams.nb_nodes),
        dim=int(hparams.dim_latent_space),
        dim_cond=dim_cond,
        width=int(hparams.coeff_fn_internal_size),
        n_hidden=int(hparams.coeff_fn_config),
        clamp=float(hparams.rnvp_clamp),
        ndof=robot.ndof,
        sigmoid_on_output=bool(hparams.sigmoid_on_output),
    )


# ---------------------------------------------------------------------------------------------------
# state_dict key names (FrEIA GraphINN.module_list; evidence for "module_list.0.M": reference
# scripts/download_model_from_wandb_checkpoint.py:13-28).  Block i: permutation at module 2i+1,
# coupling block at module 2i+2; Linear layers sit at even indices of the nn.Sequential (0,2,4,6).
# ---------------------------------------------------------------------------------------------------
def key_perm(i: int, off: int = 0) -> str:
    return f"module_list.{2 * i + 1 + off}.perm"


def key_perm_inv(i: int, off: int = 0) -> str:
    return f"module_list.{2 * i + 1 + off}.perm_inv"


def key_linear(i: int, subnet: int, layer: int, what: str, off: int = 0) -> str:
    """`off` = FlowLayout.module_offset (1 for the sigmoid_on_output graph, else 0)."""
    return f"module_list.{2 * i + 2 + off}.subnet{subnet}.{2 * layer}.{what}"


def freia_permutation(dim: int, seed: int) -> np.ndarray:
    """FrEIA PermuteRandom(seed): np.random.seed(seed); np.random.permutation(dim)  (ikflow/model.py:339).
    Uses a private legacy RandomState so numpy's global RNG is not reseeded as a side effect."""
    return np.random.RandomState(seed).permutation(dim).astype(np.int64)


def fixed_linear_transform(layout: FlowLayout, robot: Robot) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(M, M_inv, b) exactly as FrEIA FixedLinearTransform stores them for ikflow/model.py:310-316:
    x_invSig = diag(1 / max(|lo_i|, |hi_i|)) in float32, M = x_invSig.t(), M_inv = M.inverse() (fp32 LU), b = 0."""
    import torch

    if layout.sigmoid_on_output:
        # get_pre_sigmoid_scaling_node (ikflow/model.py:241-288): joints [lo, hi] -> [0, 1]; the padding columns
        # [-SIGMOID_SCALING_ABS_MAX, +SIGMOID_SCALING_ABS_MAX] -> [0, 1] (ikflow/config.py:31: 1.0)
        scaling = torch.eye(layout.dim)
        offset = torch.zeros(layout.dim)
        for i in range(layout.dim):
            lo, hi = robot.actuated_joints_limits[i] if i < robot.ndof else (-SIGMOID_SCALING_ABS_MAX, SIGMOID_SCALING_ABS_MAX)
            slope = (1.0 - 0.0) / (hi - lo)
            offset[i] = 0.0 - (slope * lo)
            scaling[i, i] = slope
        M = scaling.t().contiguous()
        M_inv = scaling.t().inverse().contiguous()
        return M.numpy().copy(), M_inv.numpy().copy(), offset.unsqueeze(0).numpy().copy()
    x_inv_sig = torch.eye(layout.dim)
    for i in range(robot.ndof):
        lo, hi = robot.actuated_joints_limits[i]
        x_inv_sig[i, i] = 1.0 / max(abs(lo), abs(hi))
    M = x_inv_sig.t().contiguous()
    M_inv = x_inv_sig.t().inverse().contiguous()
    b = torch.zeros(1, layout.dim)
    return M.numpy().copy(), M_inv.numpy().copy(), b.numpy().copy()


def random_state_dict(
    layout: FlowLayout, robot: Robot, seed: int = 0, output_gain: float = 1.0
) -> Dict[str, np.ndarray]:
    """Synthetic weights with the reference's initialisation (``nn.Linear`` default: U(-1/sqrt(in), 1/sqrt(in))
    for weight and bias - the reference never reads ``init_scale``), drawn from a private torch CPU generator.

    ``output_gain`` scales each subnet's last Linear so the coupling coefficients (s, t) reach O(1) like a
    trained model's do (default init leaves them ~0.05 and the atan/exp path nearly untouched).
    Released weights are remote files (model_descriptions.yaml:17) and not available offline.
    """
    import torch

    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, np.ndarray] = {}
    M, M_inv, b = fixed_linear_transform(layout, robot)
    sd["module_list.0.M"] = M
    sd["module_list.0.M_inv"] = M_inv
    sd["module_list.0.b"] = b
    off = layout.module_offset
    for i in range(layout.nb_nodes):
        perm = freia_permutation(layout.dim, i)
        perm_inv = np.zeros_like(perm)
        perm_inv[perm] = np.arange(layout.dim)
        sd[key_perm(i, off)] = perm
        sd[key_perm_inv(i, off)] = perm_inv
        for subnet in (1, 2):
            dims = layout.subnet_dims(subnet)
            for layer, (cin, cout) in enumerate(dims):
                bound = 1.0 / np.sqrt(cin)
                w = (torch.rand(cout, cin, generator=g) * 2 - 1) * bound
                bb = (torch.rand(cout, generator=g) * 2 - 1) * bound
                if layer == len(dims) - 1:
                    w = w * output_gain
                    bb = bb * output_gain
                sd[key_linear(i, subnet, layer, "weight", off)] = w.numpy().copy()
                sd[key_linear(i, subnet, layer, "bias", off)] = bb.numpy().copy()
    return sd


def validate_state_dict(layout: FlowLayout, sd: Dict[str, np.ndarray]) -> None:
    """Shape check with the error text style of ``nn.Module.load_state_dict`` (missing / mismatched keys)."""
    missing, bad = [], []
    for k in ("module_list.0.M_inv",):
        if k not in sd:
            missing.append(k)
        elif tuple(sd[k].shape) != (layout.dim, layout.dim):
            bad.append((k, tuple(sd[k].shape), (layout.dim, layout.dim)))
    if layout.sigmoid_on_output:  # the scaling node's offset b = -slope * lo is never zero: a file without it is broken
        if "module_list.0.b" not in sd:
            missing.append("module_list.0.b")
        elif int(np.prod(sd["module_list.0.b"].shape)) != layout.dim:
            bad.append(("module_list.0.b", tuple(sd["module_list.0.b"].shape), (1, layout.dim)))
    off = layout.module_offset
    for i in range(layout.nb_nodes):
        k = key_perm_inv(i, off)
        if k not in sd:
            missing.append(k)
        elif tuple(sd[k].shape) != (layout.dim,):
            bad.append((k, tuple(sd[k].shape), (layout.dim,)))
        for subnet in (1, 2):
            for layer, (cin, cout) in enumerate(layout.subnet_dims(subnet)):
                kw, kb = key_linear(i, subnet, layer, "weight", off), key_linear(i, subnet, layer, "bias", off)
                if kw not in sd:
                    missing.append(kw)
                elif tuple(sd[kw].shape) != (cout, cin):
                    bad.append((kw, tuple(sd[kw].shape), (cout, cin)))
                if kb not in sd:
                    missing.append(kb)
                elif tuple(sd[kb].shape) != (cout,):
                    bad.append((kb, tuple(sd[kb].shape), (cout,)))
    if missing or bad:
        msg = "Error(s) in loading state_dict for the flow:"
        if missing:
            msg += f"\n\tMissing key(s) in state_dict: {', '.join(repr(m) for m in missing[:8])}" + (" ..." if len(missing) > 8 else "")
        for k, got, want in bad[:8]:
            msg += f"\n\tsize mismatch for {k}: copying a param with shape {got} from checkpoint, the shape in current model is {want}."
        raise RuntimeError(msg)


def state_dict_to_numpy(state_dict) -> Dict[str, np.ndarray]:
    """Accept {str: torch.Tensor | np.ndarray}; return contiguous numpy arrays (fp32 / int64)."""
    out: Dict[str, np.ndarray] = {}
    for k, v in state_dict.items():
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        v = np.ascontiguousarray(v)
        if v.dtype.kind == "f":
            v = v.astype(np.float32, copy=False)
        elif v.dtype.kind in "iu":
            v = v.astype(np.int64, copy=False)
        out[str(k)] = v
    return out
