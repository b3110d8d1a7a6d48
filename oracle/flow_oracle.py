"""ORACLE (test infrastructure - never imported by the product path in ikflow_amd/).

CPU restatement of the conditional-flow inverse pass that ``IKFlowSolver._run_inference`` executes
(``ikflow/ikflow_solver.py:85-110``: ``nn_model(latent, c=conditional, rev=True)`` -> ``[:, :ndof]`` -> clamp).

The arithmetic lives in FrEIA==0.2 (pyproject.toml:11, uv.lock:533-541), which is NOT in /root/reference and
not installed; what is restated below is FrEIA's published algorithm for the three modules the graph of
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

``flow_inverse_torch`` uses the same torch CPU ops the reference would run (F.linear -> MKL sgemm, leaky_relu,
atan, exp, cat, index) - it IS the "reference PyTorch-CPU path" arithmetic.  ``flow_inverse_f64`` is a numpy
float64 twin used to arbitrate rounding disputes.

PARITY STATUS: the coupling-block numerics are pinned by NO reference test or golden vector;
what is pinned - the permutation tables (numpy legacy MT19937) and the Panda scale vector - is checked in
tests/test_oracle_golden.py.  Coupling numerics: "parity unpinned" (restated from FrEIA's published code).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from ikflow_amd.model import ATAN_CLAMP_GAIN, LEAKY_RELU_SLOPE, FlowLayout, key_linear, key_perm_inv


def _t(a) -> torch.Tensor:
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))


def subnet_torch(sd: Dict, layout: FlowLayout, block: int, which: int, u: torch.Tensor) -> torch.Tensor:
    """ikflow/model.py:51-96 - Linear, LeakyReLU, ..., Linear."""
    n_lin = layout.n_hidden + 1
    off = layout.module_offset
    h = u
    for layer in range(n_lin):
        w = _t(sd[key_linear(block, which, layer, "weight", off)])
        b = _t(sd[key_linear(block, which, layer, "bias", off)])
        h = F.linear(h, w, b)
        if layer != n_lin - 1:
            h = F.leaky_relu(h, LEAKY_RELU_SLOPE)
    return h


def flow_inverse_torch(
    sd: Dict, layout: FlowLayout, latent: torch.Tensor, conditional: torch.Tensor
) -> torch.Tensor:
    """[n x D] latent, [n x dim_cond] conditional -> [n x D] output_rev (before the [:, :ndof] slice)."""
    assert latent.dtype == torch.float32 and conditional.dtype == torch.float32
    L1, L2 = layout.split1, layout.split2
    clamp = layout.clamp
    with torch.inference_mode():
        x = latent
        c = conditional
        for i in reversed(range(layout.nb_nodes)):
            x1, x2 = torch.split(x, [L1, L2], dim=1)
            a1 = subnet_torch(sd, layout, i, 1, torch.cat([x1, c], 1))
            s1, t1 = a1[:, :L2], a1[:, L2:]
            s1 = clamp * (ATAN_CLAMP_GAIN * torch.atan(s1))
            y2 = (x2 - t1) * torch.exp(-s1)
            a2 = subnet_torch(sd, layout, i, 2, torch.cat([y2, c], 1))
            s2, t2 = a2[:, :L1], a2[:, L1:]
            s2 = clamp * (ATAN_CLAMP_GAIN * torch.atan(s2))
            y1 = (x1 - t2) * torch.exp(-s2)
            x = torch.cat((y1, y2), 1)
            x = x[:, _t(sd[key_perm_inv(i, layout.module_offset)]).long()]
        if layout.sigmoid_on_output:
            x = 1 / (1 + torch.exp(-x))  # InvertibleSigmoidFlipped, rev branch (ikflow/model.py:124-127)
        b = _t(sd["module_list.0.b"]) if "module_list.0.b" in sd else 0.0
        x = (x - b).mm(_t(sd["module_list.0.M_inv"]))
    return x


def run_inference_torch(
    sd: Dict, layout: FlowLayout, limits, latent: torch.Tensor, conditional: torch.Tensor, clamp_to_joint_limits: bool
) -> torch.Tensor:
    """ikflow_solver.py:98-102: flow inverse, [:, :ndof], optional clamp."""
    out = flow_inverse_torch(sd, layout, latent, conditional)
    sol = out[:, : layout.ndof]
    if clamp_to_joint_limits:
        lo = torch.tensor([l[0] for l in limits], dtype=torch.float32)
        hi = torch.tensor([l[1] for l in limits], dtype=torch.float32)
        sol = torch.max(torch.min(sol, hi), lo)
    return sol.clone()


def generate_ik_solutions_torch(sd, layout, limits, y: torch.Tensor, latent: torch.Tensor, clamp=True, n=None):
    """ikflow_solver.py:328-343 conditional assembly (single-pose broadcast or batch) + inference."""
    if y.numel() == 7:
        n = latent.shape[0] if n is None else n
        cond = torch.cat([y.reshape(1, 7).expand((n, 7)), torch.zeros((n, 1))], dim=1)
    else:
        n = y.shape[0]
        cond = torch.cat([y, torch.zeros((n, 1))], dim=1)
    if layout.dim_cond == 7:
        cond = cond[:, :7]
    return run_inference_torch(sd, layout, limits, latent, cond.contiguous(), clamp)


# ---------------------------------------------------------------------------------------------------
# float64 twin
# ---------------------------------------------------------------------------------------------------
def flow_inverse_f64(sd: Dict, layout: FlowLayout, latent: np.ndarray, conditional: np.ndarray) -> np.ndarray:
    L1, L2 = layout.split1, layout.split2
    clamp = np.float64(np.float32(layout.clamp))
    gain = np.float64(np.float32(ATAN_CLAMP_GAIN))
    slope = np.float64(np.float32(LEAKY_RELU_SLOPE))
    n_lin = layout.n_hidden + 1

    def subnet(block, which, u):
        h = u
        for layer in range(n_lin):
            w = np.asarray(sd[key_linear(block, which, layer, "weight", layout.module_offset)], dtype=np.float64)
            b = np.asarray(sd[key_linear(block, which, layer, "bias", layout.module_offset)], dtype=np.float64)
            h = h @ w.T + b
            if layer != n_lin - 1:
                h = np.where(h > 0, h, slope * h)
        return h

    x = np.asarray(latent, dtype=np.float64)
    c = np.asarray(conditional, dtype=np.float64)
    for i in reversed(range(layout.nb_nodes)):
        x1, x2 = x[:, :L1], x[:, L1:]
        a1 = subnet(i, 1, np.concatenate([x1, c], 1))
        s1, t1 = a1[:, :L2], a1[:, L2:]
        y2 = (x2 - t1) * np.exp(-(clamp * (gain * np.arctan(s1))))
        a2 = subnet(i, 2, np.concatenate([y2, c], 1))
        s2, t2 = a2[:, :L1], a2[:, L1:]
        y1 = (x1 - t2) * np.exp(-(clamp * (gain * np.arctan(s2))))
        x = np.concatenate([y1, y2], 1)[:, np.asarray(sd[key_perm_inv(i, layout.module_offset)], dtype=np.int64)]
    if layout.sigmoid_on_output:
        x = 1.0 / (1.0 + np.exp(-x))
    b = np.asarray(sd["module_list.0.b"], dtype=np.float64) if "module_list.0.b" in sd else 0.0
    return (x - b) @ np.asarray(sd["module_list.0.M_inv"], dtype=np.float64)


def run_inference_f64(sd, layout, limits, latent, conditional, clamp_to_joint_limits: bool) -> np.ndarray:
    out = flow_inverse_f64(sd, layout, latent, conditional)[:, : layout.ndof]
    if clamp_to_joint_limits:
        lo = np.array([np.float32(l[0]) for l in limits], dtype=np.float64)
        hi = np.array([np.float32(l[1]) for l in limits], dtype=np.float64)
        out = np.clip(out, lo, hi)
    return out
