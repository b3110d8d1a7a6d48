"""Generates tests/golden/ref_vectors.npz: vectors produced by the REFERENCE's own code, run in the build container.

The reference package cannot be imported (its module-level imports need FrEIA and jrl, which are absent - SURVEY 8(c)),
but four pieces of /root/reference/ikflow/model.py on the hot path depend on nothing but torch:

  * ``subnet_constructor``                    ikflow/model.py:51-96    the coupling subnet (Linear / LeakyReLU stack, configs 1..4)
  * ``IkflowModelParameters`` + TINY          ikflow/model.py:17-48    the hyper-parameter bag and its defaults
  * ``IkFlowFixedLinearTransform.forward``    ikflow/model.py:191-233  y = x.mm(M) + b ;  rev: (x - b).mm(M_inv)   (the in-tree
                                                                       twin of FrEIA's FixedLinearTransform, graph node 0)
  * ``InvertibleSigmoidFlipped.forward``      ikflow/model.py:120-146  rev: 1 / (1 + exp(-x))  (sigmoid_on_output graph)
and three more of the path's helpers that are pure torch / numpy:
  * ``draw_latent``                           ikflow/ikflow_solver.py:16-29        the latent sampler (row A0)
  * ``calculate_joint_limits_exceeded``       ikflow/evaluation_utils.py:100-112   strict-inequality limit flags (f-2)
  * ``_get_target_pose_batch``                ikflow/evaluation_utils.py:22-34     single-pose tiling rule

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
    # --- draw_latent (ikflow_solver.py:16-29), calculate_joint_limits_exceeded / _get_target_pose_batch (evaluation_utils.py)
    import typing

    def load_fns(path, names, extra):
        tree = ast.parse(open(path).read())
        keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
        env = {"torch": torch, "np": np, "Tuple": typing.Tuple, "List": typing.List, "Optional": typing.Optional}
        env.update(extra)
        exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), env)
        return env

    sol = load_fns("/root/reference/ikflow/ikflow_solver.py", {"draw_latent"}, {})
    torch.manual_seed(1234)
    out["latent_gaussian"] = sol["draw_latent"]("gaussian", 0.75, (5, 7), "cpu").numpy().copy()
    out["latent_uniform"] = sol["draw_latent"]("uniform", 2.0, (5, 7), "cpu").numpy().copy()
    # PT_NP_TYPE is jrl.config's annotation alias for "numpy array or torch tensor"; only the annotation needs the name
    ev = load_fns("/root/reference/ikflow/evaluation_utils.py", {"calculate_joint_limits_exceeded", "_get_target_pose_batch"},
                  {"PT_NP_TYPE": typing.Union[np.ndarray, torch.Tensor]})
    qg = torch.Generator().manual_seed(11)
    lo7 = torch.tensor([l[0] for l in limits])
    hi7 = torch.tensor([l[1] for l in limits])
    cfg = lo7 + (hi7 - lo7) * (1.2 * torch.rand(400, 7, generator=qg) - 0.1)  # ~1/3 of the rows leave the limits
    cfg[:7] = lo7.repeat(7, 1)                                                   # exactly on a limit: not exceeded (strict)
    cfg[7:14] = hi7.repeat(7, 1)
    cfg[14, 3] = float(np.nextafter(np.float32(hi7[3]), np.float32(1e9)))        # one float32 ulp beyond: exceeded
    cfg[15, 5] = float(np.nextafter(np.float32(lo7[5]), np.float32(-1e9)))
    out["limits_cfg"] = cfg.numpy().copy()
    out["limits_exceeded"] = ev["calculate_joint_limits_exceeded"](cfg, limits).numpy().copy()
    one = torch.arange(7, dtype=torch.float32)
    out["tpb_single"] = ev["_get_target_pose_batch"](one, 4).numpy().copy()
    batch = cfg[:5].clone()
    out["tpb_batch_is_identity"] = np.array(ev["_get_target_pose_batch"](batch, 5) is batch)
    path = os.path.join(HERE, "ref_vectors.npz")
    np.savez(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
