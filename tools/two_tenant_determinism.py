"""Is a call's result a function of its inputs when another process uses the same GPU?  (r06: the red driver test of round 5 was NOT gloo's
staging - tools/two_rank_gather_repro.py showed the same (poses, latent) giving different rows from one call to the next under two tenants.)

`tenants` processes share cuda:0; each loops `iters` calls of generate_ik_solutions over the SAME inputs, alternating two batch sizes like the
failing test, and compares every result bit for bit with its first one.  One JSON line per (case, tenant): calls that differed, the worst
difference, which rows (first / last / count), and - per differing call - whether the rows come in whole tiles.

    python tools/two_tenant_determinism.py --out gpurun_out/r06/two_tenant_determinism.jsonl
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.multiprocessing as mp


def _worker(tenant, case, iters, barrier, q):
    from helpers import latents, panda_model, reachable_poses, tiny_model
    from ikflow_amd.ikflow_solver import IKFlowSolver

    dev = "cuda:0"
    model, sizes, variants, sync = case["model"], case["sizes"], case.get("variants", []), case.get("sync", False)
    robot, hp, lay, sd = (tiny_model(seed=4) if model == "tiny" else panda_model())
    s = IKFlowSolver(hp, robot)
    s.load_state_dict_tensors(sd)
    eng = s.engine(dev)
    for v in variants:
        eng.set_gemm_variant(v)
    inputs = []
    for i, n in enumerate(sizes):
        _, poses = reachable_poses(robot, n, 5 + i)
        inputs.append((poses.to(dev), latents(n, lay.dim, 6 + i).to(dev)))
    first = [s.generate_ik_solutions(p, latent=l).clone() for p, l in inputs]
    torch.cuda.synchronize()
    plans = [eng.plan(n) for n in sizes]
    barrier.wait()
    bad_calls, worst, examples = 0, 0.0, []
    for it in range(iters):
        for k, (p, l) in enumerate(inputs):
            got = s.generate_ik_solutions(p, latent=l)
            if sync:
                torch.cuda.synchronize()
            d = (got - first[k]).abs()
            m = float(d.max())
            if m > 0:
                bad_calls += 1
                worst = max(worst, m)
                rows = (d > 0).any(dim=1).nonzero().flatten()
                if len(examples) < 8:
                    r = rows.tolist()
                    examples.append({"iter": it, "size": sizes[k], "max_abs": m, "rows_first_last_count": [r[0], r[-1], len(r)],
                                     "rows_mod16_all_in_one_tile_run": bool(r[-1] - r[0] + 1 == len(r)), "first_rows": r[:6]})
    torch.cuda.synchronize()
    repairs = eng.cluster_repairs
    q.put({"case": case["name"], "tenant": tenant, "calls": iters * len(sizes), "calls_that_differ": bad_calls, "worst_abs": worst, "plans": plans,
           "cluster_repairs": repairs, "examples": examples})


CASES = [
    {"name": "tiny_default_1tenant", "model": "tiny", "sizes": [751, 750], "tenants": 1},
    {"name": "tiny_default_2tenants", "model": "tiny", "sizes": [751, 750], "tenants": 2},
    {"name": "tiny_default_2tenants_sync_every_call", "model": "tiny", "sizes": [751, 750], "tenants": 2, "sync": True},
    {"name": "tiny_no_write_through_2tenants", "model": "tiny", "sizes": [751, 750], "tenants": 2, "variants": [130]},
    {"name": "tiny_head_off_2tenants", "model": "tiny", "sizes": [751, 750], "tenants": 2, "variants": [110]},
    {"name": "tiny_unfused_2tenants", "model": "tiny", "sizes": [751, 750], "tenants": 2, "variants": [0]},
    {"name": "tiny_one_size_2tenants", "model": "tiny", "sizes": [751], "tenants": 2},
    {"name": "tiny_16row_tiles_2tenants", "model": "tiny", "sizes": [100, 40, 200], "tenants": 2},
    {"name": "panda_perlayer600_2tenants", "model": "panda", "sizes": [600], "tenants": 2, "variants": [185, 180], "iters": 60},
    {"name": "panda_cluster512_2tenants", "model": "panda", "sizes": [512], "tenants": 2, "iters": 60},
    {"name": "panda_rowowner4096_2tenants", "model": "panda", "sizes": [4096], "tenants": 2, "iters": 30},
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--only", default="")
    ap.add_argument("--strict", action="store_true", help="exit 1 when any call of any case differed from its first result")
    ap.add_argument("--out", default="gpurun_out/r06/two_tenant_determinism.jsonl")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    ctx = mp.get_context("spawn")
    differing = 0
    with open(a.out, "w") as f:
        for case in CASES:
            if a.only and a.only != case["name"]:
                continue
            n_t = case["tenants"]
            q, barrier = ctx.Queue(), ctx.Barrier(n_t)
            procs = [ctx.Process(target=_worker, args=(t, case, case.get("iters", a.iters), barrier, q)) for t in range(n_t)]
            for p in procs:
                p.start()
            res = [q.get(timeout=900) for _ in range(n_t)]
            for p in procs:
                p.join(timeout=60)
            for r in sorted(res, key=lambda x: x["tenant"]):
                differing += r["calls_that_differ"]
                line = json.dumps(r)
                f.write(line + "\n")
                f.flush()
                print(line[:700], flush=True)


    if a.strict and differing:
        sys.exit(1)


if __name__ == "__main__":
    main()
