"""Generates tests/golden/ref_vectors.npz: vectors produced by the REFERENCE's own code, run in the build container.

The reference package cannot be imported (its module-level imports need FrEIA and jrl, which are absent - SURVEY 8(c)),
but four pieces of /root/reference/ikflow/model.py on the hot path depend on nothing but torch:

  * ``subnet_constructor``                    ikflow/model.py:51-96    the coupling subnet (Linear / LeakyReLU stack, configs 1..4)
  * ``IkflowModelParameters`` + TINY          ikflow/model.py:17-48    the hyper-parameter bag and its defaults
  * ``IkFlowFixedLinearTransform.forward``    ikflow/model.py:191-233  y = x.mm(M) + b ;  rev: (x - b).mm(M_inv)   (the in-tree
                                                                       twin of FrEIA's FixedLinearTransform, graph node 0)
  * ``InvertibleSigmoidFlipped.forward``      ikflow/model.py:120-146  rev: 1 / (1 + exp(-x))  (sigmoid_on_output graph)

They are taken out of the reference file with ``ast`` at generation time and executed - functions as they are, the two
``forward`` methods as plain functions called with a namespace object in place of ``self`` that carries the tensors
(M, M_inv, b, logDetM, joint_limits).  No stand-in module is written and nothing is copied into the repo; the fixture
holds inputs and outputs only.  It pins, against the reference itself: layer order, LeakyReLU slope, Sequential key
names ("0.weight", "2.weight", ...), the nn.Linear initialisation order for every ``coeff_fn_config``, the default
hyper-parameters, the (x - b).mm(M_inv) algebra and the flipped sigmoid.

Run from the repo root (only where /root/reference exists):  python tests/golden/make_ref_vectors.py
"""
import ast
import json
import os
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference/ikflow/model.py"
HERE = os.path.dirname(os.path.abspath(__file__))
WIDTH, CH_IN, CH_OUT, ROWS = 32, 11, 8, 6


UTILS = "/root/reference/ikflow/utils.py"


def _load(names, methods=()):
    """Execute the named top-level functions / classes of the reference's model.py, and the ``forward`` methods of the
    classes listed in `methods` as plain functions ``<Class>_forward``."""
    from typing import Iterable, List, Tuple, Union  # the annotations of the extracted code

    ns = {"nn": nn, "torch": torch, "Iterable": Iterable, "Tuple": Tuple, "List": List, "Union": Union}
    utree = ast.parse(open(UTILS).read())
    keep = [n for n in utree.body if isinstance(n, ast.FunctionDef) and n.name == "assert_joint_angle_tensor_in_joint_limits"]
    exec(compile(ast.Module(body=keep, type_ignores=[]), UTILS, "exec"), ns)
    tree = ast.parse(open(REF).read())
    keep = [n for n in tree.body
            if (isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names)
            or (isinstance(n, ast.Assign) and any(isinstance(t, (ast.Name, ast.Attribute)) and ast.unparse(t).startswith("TINY_MODEL_PARAMS") for t in n.targets))]
    for cls in tree.body:
        if isinstance(cls, ast.ClassDef) and cls.name in methods:
            for fn in cls.body:
                if isinstance(fn, ast.FunctionDef) and fn.name == "forward":
                    fn.name = cls.name + "_forward"
                    keep.append(fn)
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF, "exec"), ns)
    return ns


def main():
    torch.set_num_threads(1)
    ns = _load({"subnet_constructor", "IkflowModelParameters"}, methods=("IkFlowFixedLinearTransform", "InvertibleSigmoidFlipped"))
    out = {}
    for n_layers in (1, 2, 3, 4):
        torch.manual_seed(100 + n_layers)
        net = ns["subnet_constructor"](WIDTH, n_layers, CH_IN, CH_OUT)
        x = torch.randn(ROWS, CH_IN, generator=torch.Generator().manual_seed(n_layers))
        with torch.inference_mode():
            y = net(x)
        sd = net.state_dict()
        out[f"c{n_layers}_keys"] = np.array(json.dumps(list(sd.keys())))
        for k, v in sd.items():
            out[f"c{n_layers}_{k}"] = v.numpy().copy()
        out[f"c{n_layers}_x"] = x.numpy().copy()
        out[f"c{n_layers}_y"] = y.numpy().copy()
        out[f"c{n_layers}_modules"] = np.array(json.dumps([type(m).__name__ for m in net]))
        out[f"c{n_layers}_slopes"] = np.array([m.negative_slope for m in net if isinstance(m, nn.LeakyReLU)], dtype=np.float64)
    out["hparam_defaults"] = np.array(json.dumps(ns["IkflowModelParameters"]().__dict__))
    out["hparam_tiny"] = np.array(json.dumps(ns["TINY_MODEL_PARAMS"].__dict__))
    # --- IkFlowFixedLinearTransform.forward, both directions, on the Panda scaling node (D = 9: 7 joints + 2 padding columns)
    # limits: /root/reference/tests/model_test.py:27-44; padding columns [-1, 1] (ikflow/config.py:31)
    limits = [(-2.8973, 2.8973), (-1.7628, 1.7628), (-2.8973, 2.8973), (-3.0718, -0.0698), (-2.8973, 2.8973),
              (-0.0175, 3.7525), (-2.8973, 2.8973)]
    D = 9
    scaling, offset = torch.eye(D), torch.zeros(D)
    for i in range(D):
        lo, hi = limits[i] if i < 7 else (-1.0, 1.0)
        scaling[i, i] = 1.0 / (hi - lo)
        offset[i] = -lo / (hi - lo)
    # the constructor's own statements (ikflow/model.py:180-189): M.t(), M.t().inverse(), b.unsqueeze(0), slogdet
    me = types.SimpleNamespace(M=scaling.t(), M_inv=scaling.t().inverse(), b=offset.unsqueeze(0), joint_limits=limits,
                               logDetM=torch.slogdet(scaling)[1])
    g = torch.Generator().manual_seed(7)
    lo_t = torch.tensor([l[0] for l in limits] + [-1.0, -1.0])
    hi_t = torch.tensor([l[1] for l in limits] + [1.0, 1.0])
    xq = lo_t + (hi_t - lo_t) * (0.02 + 0.96 * torch.rand(ROWS, D, generator=g))
    (fwd,), jf = ns["IkFlowFixedLinearTransform_forward"](me, (xq,), rev=False)
    x01 = 0.02 + 0.96 * torch.rand(ROWS, D, generator=g)
    (rev,), jr = ns["IkFlowFixedLinearTransform_forward"](me, (x01,), rev=True)
    out.update(flt_M=me.M.numpy().copy(), flt_M_inv=me.M_inv.numpy().copy(), flt_b=me.b.numpy().copy(),
               flt_fwd_in=xq.numpy().copy(), flt_fwd_out=fwd.numpy().copy(), flt_rev_in=x01.numpy().copy(),
               flt_rev_out=rev.numpy().copy(), flt_logdet=np.array([float(jf[0]), float(jr[0])]))
    # --- InvertibleSigmoidFlipped.forward, rev (the direction the inverse pass runs) and forward
    z = 3.0 * torch.randn(ROWS, D, generator=g)
    (sig,), sj = ns["InvertibleSigmoidFlipped_forward"](None, (z,), rev=True)
    (logit,), lj = ns["InvertibleSigmoidFlipped_forward"](None, (x01,), rev=False)
    out.update(sig_rev_in=z.numpy().copy(), sig_rev_out=sig.numpy().copy(), sig_rev_logdet=sj.numpy().copy(),
               sig_fwd_out=logit.numpy().copy(), sig_fwd_logdet=lj.numpy().copy())
    path = os.path.join(HERE, "ref_vectors.npz")
    np.savez(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
