"""ORACLE (test infrastructure - never imported by the product path in ikflow_amd/; imports nothing from ikflow_amd).

CPU restatement of the conditional-flow inverse pass that ``IKFlowSolver._run_inference`` executes
(/root/reference/ikflow/ikflow_solver.py:85-110: ``nn_model(latent, c=conditional, rev=True)`` -> ``[:, :ndof]`` -> clamp).

The arithmetic lives in FrEIA==0.2 (reference pyproject.toml:11, uv.lock:533-541), which is NOT in /root/reference and not
installed; what is restated below is FrEIA's published algorithm for the three modules the graph of
``ikflow/model.py:300-354`` contains, in the reverse direction and in FrEIA's op order:

  GraphINN.forward(rev=True)     : modules visited last -> first
  GLOWCouplingBlock (rev)        : x1,x2 = split(x,[D//2, D-D//2])
                                   a1 = subnet1(cat[x1,c]); s1,t1 = a1[:, :L2], a1[:, L2:]
                                   s1 = clamp*(0.636*atan(s1));  y2 = (x2 - t1)*exp(-s1)
                                   a2 = subnet2(cat[y2,c]); s2,t2 = a2[:, :L1], a2[:, L1:]
                                   s2 = clamp*(0.636*atan(s2));  y1 = (x1 - t2)*exp(-s2);  out = cat[y1,y2]
  PermuteRandom (rev)            : x[:, perm_inv]
  FixedLinearTransform (rev)     : (x - b).mm(M_inv)      (in-tree twin of the same algebra: ikflow/model.py:220)
  sigmoid_on_output variant      : InvertibleSigmoidFlipped rev = 1/(1+exp(-x)) (ikflow/model.py:124-127), then the
                                   scaling node's (x - b).mm(M_inv) with M, b of get_pre_sigmoid_scaling_node (:241-288)
  subnet                         : Linear/LeakyReLU(0.01) stack, ikflow/model.py:51-96

Everything the flow needs is defined HERE a second time - constants, split rule, state_dict key names, permutation tables,
the fixed linear transform, the hyper-parameters of the released models and a weight generator that builds real
``torch.nn.Sequential`` subnets - so that a wrong constant / key index / split in the product is not shared by its checker
(tests/test_oracle_independence.py compares the two sets of tables).

``flow_inverse_torch`` uses the same torch CPU ops the reference would run (F.linear -> MKL sgemm, leaky_relu,
atan, exp, cat, index) - it IS the "reference PyTorch-CPU path" arithmetic.  ``flow_inverse_f64`` is a numpy
float64 twin used to arbitrate rounding disputes.

PARITY STATUS.  Pinned by reference-derived vectors: the subnet stack (layer order, LeakyReLU slope, Sequential key
names) for coeff_fn_config 1..4 - tests/golden/ref_vectors.npz holds outputs of the reference's OWN ``subnet_constructor``,
``IkFlowFixedLinearTransform.forward`` and ``InvertibleSigmoidFlipped.forward`` (ikflow/model.py:51-96, 120-146, 191-233,
executed from the reference file by tests/golden/make_ref_vectors.py in the build container); the permutation tables (numpy legacy MT19937 literals); the Panda scale vector; the sigmoid scaling node's known answers
(tests/model_test.py:50-123).  NOT pinned by any reference vector: the coupling arithmetic itself (split order, s|t order,
0.636*atan clamp, perm_inv direction) - "parity unpinned", restated from FrEIA 0.2's published code.  The pin for it is armed:
tests/golden/make_ref_thirdparty.py + tests/test_thirdparty_pin.py compare this file with FrEIA's GraphINN itself wherever FrEIA==0.2 and
jrl are importable (they are not in the build container).  Independent of any recall: ``flow_forward_f64`` runs the graph the other way, and
forward(inverse(z)) = z is checked at full batch size on the GPU path (a flow is a bijection).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle.robot_tables import OracleRobot
from oracle.robot_tables import robot as _robot_by_name

GLOW_ATAN_GAIN = 0.636  # FrEIA GLOWCouplingBlock, clamp_activation="ATAN": s -> clamp * 0.636 * atan(s)
LEAKY_SLOPE = 0.01  # nn.LeakyReLU() default negative_slope (ikflow/model.py:63-95 constructs it without arguments)
SIGMOID_PAD_ABS_MAX = 1.0  # ikflow/config.py:31 SIGMOID_SCALING_ABS_MAX

# ikflow/model_descriptions.yaml (hyper-parameters only) + ikflow/model.py:45-48 (TINY); softflow_enabled defaults True
RELEASED = {
    "panda__full__lp191_5.25m": ("panda", 12, 7, 3, 1024, 2.5),  # yaml:10-17
    "panda_lite_tpm": ("panda", 6, 7, 3, 1024, 2.5),  # yaml:19-26
    "fetch_full_temp_nsc_tpm": ("fetch", 12, 8, 3, 1024, 2.5),  # yaml:33-40
    "fetch__large__ns183_9.75m": ("fetch", 16, 8, 3, 1024, 2.5),  # yaml:42-49
    "fetch_arm__large__mh186_9.25m": ("fetch_arm", 16, 10, 3, 1024, 2.5),  # yaml:56-63
    "tiny": ("panda", 3, 9, 2, 256, 2.5),  # ikflow/model.py:45-48, dim_latent_space default 9
}
# released hyper-parameters whose robot has no chain in oracle/robot_tables.py (the URDF ships with jrl, absent here)
RELEASED_NO_CHAIN = {
    "rizon4__snowy-brook-208__global_step=2.75M": ("rizon4", 12, 7, 3, 1024, 2.5),  # yaml:90-97
}


@dataclass(frozen=True)
class OracleLayout:
    nb_nodes: int
    dim: int  # D = dim_latent_space
    dim_cond: int  # 8 with softflow (pose + scale), 7 without (ikflow_solver.py:51-53)
    width: int  # coeff_fn_internal_size
    n_hidden: int  # coeff_fn_config
    clamp: float  # rnvp_clamp
    ndof: int
    sigmoid_on_output: bool = False

    @staticmethod
    def of(obj) -> "OracleLayout":
        """Copy the plain hyper-parameter fields of any layout-like object (nothing derived is taken over)."""
        if isinstance(obj, OracleLayout):
            return obj
        return OracleLayout(int(obj.nb_nodes), int(obj.dim), int(obj.dim_cond), int(obj.width), int(obj.n_hidden),
                            float(obj.clamp), int(obj.ndof), bool(getattr(obj, "sigmoid_on_output", False)))

    @property
    def len1(self) -> int:  # ikflow/model.py:336: split_dimension = ndim_tot // 2
        return self.dim // 2

    @property
    def len2(self) -> int:
        return self.dim - self.dim // 2

    @property
    def first_block_module(self) -> int:
        """Index in GraphINN.module_list of block 0's PermuteRandom: 1 after the single FixedLinearTransform, 2 after the
        (scaling node, flipped sigmoid) pair of the sigmoid_on_output graph (ikflow/model.py:304-316)."""
        return 2 if self.sigmoid_on_output else 1

    def perm_module(self, block: int) -> int:
        return self.first_block_module + 2 * block

    def glow_module(self, block: int) -> int:
        return self.first_block_module + 2 * block + 1

    def flops_per_solution(self) -> int:
        total = 0
        for cin, cout in ((self.len1 + self.dim_cond, 2 * self.len2), (self.len2 + self.dim_cond, 2 * self.len1)):
            total += cin * self.width + (self.n_hidden - 1) * self.width * self.width + self.width * cout
        return 2 * self.nb_nodes * total


def layout_for(model_name: str, sigmoid_on_output: bool = False, softflow: bool = True) -> OracleLayout:
    rob, nb, dim, cfg, width, clamp = RELEASED[model_name]
    return OracleLayout(nb, dim, 8 if softflow else 7, width, cfg, clamp, _robot_by_name(rob).ndof, sigmoid_on_output)


def _oracle_robot(robot) -> OracleRobot:
    return robot if isinstance(robot, OracleRobot) else _robot_by_name(robot if isinstance(robot, str) else robot.name)


# ---------------------------------------------------------------------------------------------------
# the pieces of the graph, restated
# ---------------------------------------------------------------------------------------------------
def make_subnet(width: int, n_layers: int, ch_in: int, ch_out: int) -> nn.Sequential:
    """ikflow/model.py:51-96: Linear(ch_in, W), LeakyReLU, [Linear(W, W), LeakyReLU] x (n_layers-1), Linear(W, ch_out)."""
    assert n_layers in (1, 2, 3, 4), "Number of layers `n_layers` must be in [1, ..., 4]"
    mods = [nn.Linear(ch_in, width), nn.LeakyReLU()]
    for _ in range(n_layers - 1):
        mods += [nn.Linear(width, width), nn.LeakyReLU()]
    mods.append(nn.Linear(width, ch_out))
    return nn.Sequential(*mods)


def permute_random_tables(dim: int, seed: int):
    """FrEIA PermuteRandom(seed): np.random.seed(seed); perm = np.random.permutation(dim); perm_inv[perm[k]] = k.
    (numpy's global generator is restored afterwards - the literal call would leave it reseeded.)"""
    saved = np.random.get_state()
    try:
        np.random.seed(seed)
        perm = np.random.permutation(dim)
    finally:
        np.random.set_state(saved)
    perm_inv = np.zeros_like(perm)
    for k, p in enumerate(perm):
        perm_inv[p] = k
    return perm.astype(np.int64), perm_inv.astype(np.int64)


def fixed_linear_transform(layout, robot):
    """(M, M_inv, b) as the module stores them.  Plain graph (ikflow/model.py:310-316): x_invSig = diag(1/max|limit|),
    stored M = x_invSig.t(), M_inv = x_invSig.t().inverse(), b = zeros.  sigmoid graph (:241-288): joints [lo, hi] -> [0, 1],
    padding columns [-1, 1] -> [0, 1]; b = offsets."""
    lay = OracleLayout.of(layout)
    limits = _oracle_robot(robot).actuated_joints_limits
    m = torch.eye(lay.dim)
    off = torch.zeros(lay.dim)
    if lay.sigmoid_on_output:
        for i in range(lay.dim):
            lo, hi = limits[i] if i < lay.ndof else (-SIGMOID_PAD_ABS_MAX, SIGMOID_PAD_ABS_MAX)
            slope = (1.0 - 0.0) / (hi - lo)
            off[i] = 0.0 - (slope * lo)
            m[i, i] = slope
    else:
        for i in range(lay.ndof):
            m[i, i] = 1.0 / max(abs(limits[i][0]), abs(limits[i][1]))
    return m.t().contiguous().numpy().copy(), m.t().inverse().contiguous().numpy().copy(), off.unsqueeze(0).numpy().copy()


def make_state_dict(layout, robot, seed: int = 0, output_gain: float = 1.0) -> Dict[str, np.ndarray]:
    """A state_dict with FrEIA GraphINN key names whose Linear weights are those of real ``nn.Linear`` modules created in
    the order the reference creates them (per block: subnet1 then subnet2, GLOWCouplingBlock.__init__) under
    ``torch.manual_seed(seed)`` - i.e. the default initialisation the reference's un-trained model has (``init_scale`` is
    never read).  Key suffixes come from ``nn.Sequential.state_dict()`` itself.  ``output_gain`` scales each subnet's last
    Linear so that s, t reach O(1) like a trained model's."""
    lay = OracleLayout.of(layout)
    sd: Dict[str, np.ndarray] = {}
    M, M_inv, b = fixed_linear_transform(lay, robot)
    sd["module_list.0.M"], sd["module_list.0.M_inv"], sd["module_list.0.b"] = M, M_inv, b
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        for i in range(lay.nb_nodes):
            perm, perm_inv = permute_random_tables(lay.dim, i)
            sd[f"module_list.{lay.perm_module(i)}.perm"] = perm
            sd[f"module_list.{lay.perm_module(i)}.perm_inv"] = perm_inv
            subnets = {
                "subnet1": make_subnet(lay.width, lay.n_hidden, lay.len1 + lay.dim_cond, 2 * lay.len2),
                "subnet2": make_subnet(lay.width, lay.n_hidden, lay.len2 + lay.dim_cond, 2 * lay.len1),
            }
            for name, net in subnets.items():
                last = max(int(k.split(".")[0]) for k in net.state_dict())
                for k, v in net.state_dict().items():
                    v = v.detach().clone()
                    if int(k.split(".")[0]) == last:
                        v = v * output_gain
                    sd[f"module_list.{lay.glow_module(i)}.{name}.{k}"] = v.numpy().copy()
    return sd


def _limits(limits_or_robot):
    """A list of (lo, hi) pairs as given, or - for a robot object / robot name - the ORACLE's own limits table."""
    if isinstance(limits_or_robot, (str, OracleRobot)) or hasattr(limits_or_robot, "name"):
        return _oracle_robot(limits_or_robot).actuated_joints_limits
    return limits_or_robot


def _t(a) -> torch.Tensor:
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))


def subnet_torch(sd: Dict, layout, block: int, which: int, u: torch.Tensor) -> torch.Tensor:
    """ikflow/model.py:51-96 - Linear, LeakyReLU, ..., Linear; Linear layers sit at the even Sequential indices."""
    lay = OracleLayout.of(layout)
    base = f"module_list.{lay.glow_module(block)}.subnet{which}."
    h = u
    for layer in range(lay.n_hidden + 1):
        h = F.linear(h, _t(sd[f"{base}{2 * layer}.weight"]), _t(sd[f"{base}{2 * layer}.bias"]))
        if layer != lay.n_hidden:
            h = F.leaky_relu(h, LEAKY_SLOPE)
    return h


def flow_inverse_torch(sd: Dict, layout, latent: torch.Tensor, conditional: torch.Tensor) -> torch.Tensor:
    """[n x D] latent, [n x dim_cond] conditional -> [n x D] output_rev (before the [:, :ndof] slice)."""
    lay = OracleLayout.of(layout)
    assert latent.dtype == torch.float32 and conditional.dtype == torch.float32
    L1, L2 = lay.len1, lay.len2
    with torch.inference_mode():
        x = latent
        c = conditional
        for i in reversed(range(lay.nb_nodes)):
            x1, x2 = torch.split(x, [L1, L2], dim=1)
            a1 = subnet_torch(sd, lay, i, 1, torch.cat([x1, c], 1))
            s1, t1 = a1[:, :L2], a1[:, L2:]
            s1 = lay.clamp * (GLOW_ATAN_GAIN * torch.atan(s1))
            y2 = (x2 - t1) * torch.exp(-s1)
            a2 = subnet_torch(sd, lay, i, 2, torch.cat([y2, c], 1))
            s2, t2 = a2[:, :L1], a2[:, L1:]
            s2 = lay.clamp * (GLOW_ATAN_GAIN * torch.atan(s2))
            y1 = (x1 - t2) * torch.exp(-s2)
            x = torch.cat((y1, y2), 1)
            x = x[:, _t(sd[f"module_list.{lay.perm_module(i)}.perm_inv"]).long()]
        if lay.sigmoid_on_output:
            x = 1 / (1 + torch.exp(-x))  # InvertibleSigmoidFlipped, rev branch (ikflow/model.py:124-127)
        b = _t(sd["module_list.0.b"]) if "module_list.0.b" in sd else 0.0
        x = (x - b).mm(_t(sd["module_list.0.M_inv"]))
    return x


def run_inference_torch(sd: Dict, layout, limits, latent: torch.Tensor, conditional: torch.Tensor,
                        clamp_to_joint_limits: bool) -> torch.Tensor:
    """ikflow_solver.py:98-102: flow inverse, [:, :ndof], optional clamp."""
    lay = OracleLayout.of(layout)
    out = flow_inverse_torch(sd, lay, latent, conditional)
    sol = out[:, : lay.ndof]
    if clamp_to_joint_limits:
        limits = _limits(limits)
        lo = torch.tensor([l[0] for l in limits], dtype=torch.float32)
        hi = torch.tensor([l[1] for l in limits], dtype=torch.float32)
        sol = torch.max(torch.min(sol, hi), lo)
    return sol.clone()


def generate_ik_solutions_torch(sd, layout, limits, y: torch.Tensor, latent: torch.Tensor, clamp=True, n=None):
    """ikflow_solver.py:328-343 conditional assembly (single-pose broadcast or batch) + inference."""
    lay = OracleLayout.of(layout)
    if y.numel() == 7:
        n = latent.shape[0] if n is None else n
        cond = torch.cat([y.reshape(1, 7).expand((n, 7)), torch.zeros((n, 1))], dim=1)
    else:
        n = y.shape[0]
        cond = torch.cat([y, torch.zeros((n, 1))], dim=1)
    if lay.dim_cond == 7:
        cond = cond[:, :7]
    return run_inference_torch(sd, lay, limits, latent, cond.contiguous(), clamp)


# ---------------------------------------------------------------------------------------------------
# float64 twin
# ---------------------------------------------------------------------------------------------------
def flow_inverse_f64(sd: Dict, layout, latent: np.ndarray, conditional: np.ndarray) -> np.ndarray:
    lay = OracleLayout.of(layout)
    L1, L2 = lay.len1, lay.len2
    clamp = np.float64(np.float32(lay.clamp))
    gain = np.float64(np.float32(GLOW_ATAN_GAIN))
    slope = np.float64(np.float32(LEAKY_SLOPE))

    def subnet(block, which, u):
        base = f"module_list.{lay.glow_module(block)}.subnet{which}."
        h = u
        for layer in range(lay.n_hidden + 1):
            w = np.asarray(sd[f"{base}{2 * layer}.weight"], dtype=np.float64)
            b = np.asarray(sd[f"{base}{2 * layer}.bias"], dtype=np.float64)
            h = h @ w.T + b
            if layer != lay.n_hidden:
                h = np.where(h > 0, h, slope * h)
        return h

    x = np.asarray(latent, dtype=np.float64)
    c = np.asarray(conditional, dtype=np.float64)
    for i in reversed(range(lay.nb_nodes)):
        x1, x2 = x[:, :L1], x[:, L1:]
        a1 = subnet(i, 1, np.concatenate([x1, c], 1))
        s1, t1 = a1[:, :L2], a1[:, L2:]
        y2 = (x2 - t1) * np.exp(-(clamp * (gain * np.arctan(s1))))
        a2 = subnet(i, 2, np.concatenate([y2, c], 1))
        s2, t2 = a2[:, :L1], a2[:, L1:]
        y1 = (x1 - t2) * np.exp(-(clamp * (gain * np.arctan(s2))))
        x = np.concatenate([y1, y2], 1)[:, np.asarray(sd[f"module_list.{lay.perm_module(i)}.perm_inv"], dtype=np.int64)]
    if lay.sigmoid_on_output:
        x = 1.0 / (1.0 + np.exp(-x))
    b = np.asarray(sd["module_list.0.b"], dtype=np.float64) if "module_list.0.b" in sd else 0.0
    return (x - b) @ np.asarray(sd["module_list.0.M_inv"], dtype=np.float64)


def flow_forward_f64(sd: Dict, layout, x: np.ndarray, conditional: np.ndarray) -> np.ndarray:
    """The graph in its FORWARD (training) direction, float64: [n x D] joint-space rows (padded to D) -> [n x D] latent.  The hot path never
    runs it (ikflow_solver.py:98 calls rev=True only); it is here because a flow is a bijection - forward(inverse(z; c); c) = z is a
    size-independent property the tests check at BASELINE.json's full batch sizes without a second run of the inverse.  Node order as built by
    ikflow/model.py:300-354: FixedLinearTransform (x.mm(M) + b, the in-tree twin's forward branch model.py:214-218), then per block i = 0 .. N-1
    PermuteRandom(seed=i) (x[:, perm]) and GLOWCouplingBlock forward [RECALLED, FrEIA 0.2]: r2 = subnet2([x2, c]), y1 = exp(clamp 0.636 atan(s2)) x1 + t2;
    r1 = subnet1([y1, c]), y2 = exp(clamp 0.636 atan(s1)) x2 + t1 - the algebraic inverse of flow_inverse_* above, written independently."""
    lay = OracleLayout.of(layout)
    assert not lay.sigmoid_on_output, "forward of the sigmoid graph is not needed by any test"
    L1, L2 = lay.len1, lay.len2
    clamp = np.float64(np.float32(lay.clamp))
    gain = np.float64(np.float32(GLOW_ATAN_GAIN))
    slope = np.float64(np.float32(LEAKY_SLOPE))

    def subnet(block, which, u):
        base = f"module_list.{lay.glow_module(block)}.subnet{which}."
        h = u
        for layer in range(lay.n_hidden + 1):
            h = h @ np.asarray(sd[f"{base}{2 * layer}.weight"], dtype=np.float64).T + np.asarray(sd[f"{base}{2 * layer}.bias"], dtype=np.float64)
            if layer != lay.n_hidden:
                h = np.where(h > 0, h, slope * h)
        return h

    c = np.asarray(conditional, dtype=np.float64)
    b = np.asarray(sd["module_list.0.b"], dtype=np.float64) if "module_list.0.b" in sd else 0.0
    M = np.asarray(sd["module_list.0.M"], dtype=np.float64) if "module_list.0.M" in sd else np.linalg.inv(np.asarray(sd["module_list.0.M_inv"], dtype=np.float64))
    v = np.asarray(x, dtype=np.float64) @ M + b
    for i in range(lay.nb_nodes):
        perm_inv = np.asarray(sd[f"module_list.{lay.perm_module(i)}.perm_inv"], dtype=np.int64)
        perm = np.empty_like(perm_inv)
        perm[perm_inv] = np.arange(perm_inv.shape[0])      # (perm_inv[perm[k]] = k)
        v = v[:, perm]
        x1, x2 = v[:, :L1], v[:, L1:]
        r2 = subnet(i, 2, np.concatenate([x2, c], 1))
        y1 = np.exp(clamp * (gain * np.arctan(r2[:, :L1]))) * x1 + r2[:, L1:]
        r1 = subnet(i, 1, np.concatenate([y1, c], 1))
        y2 = np.exp(clamp * (gain * np.arctan(r1[:, :L2]))) * x2 + r1[:, L2:]
        v = np.concatenate([y1, y2], 1)
    return v


def run_inference_f64(sd, layout, limits, latent, conditional, clamp_to_joint_limits: bool) -> np.ndarray:
    lay = OracleLayout.of(layout)
    out = flow_inverse_f64(sd, lay, latent, conditional)[:, : lay.ndof]
    if clamp_to_joint_limits:
        limits = _limits(limits)
        lo = np.array([np.float32(l[0]) for l in limits], dtype=np.float64)
        hi = np.array([np.float32(l[1]) for l in limits], dtype=np.float64)
        out = np.clip(out, lo, hi)
    return out
