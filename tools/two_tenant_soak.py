"""Two processes share one GPU and both use the cluster form (which needs all of its workgroups resident): waits run out, repair launches
recompute, the form pauses and comes back.  Every result of every call must stay inside the tolerance against the same
rows through the row-owner launch (no inter-workgroup hand-over), whatever the interleaving - including tagged launches queued behind one
that gave up.  usage: python tools/two_tenant_soak.py [calls=400]  ->  one JSON line per tenant + a summary"""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def tenant(idx, calls, barrier, q):
    import torch

    from helpers import latents, panda_model, reachable_poses
    from ikflow_amd.ikflow_solver import IKFlowSolver

    dev = torch.device("cuda:0")
    robot, hp, lay, sd = panda_model(seed=idx)
    s = IKFlowSolver(hp, robot)
    s.load_state_dict_tensors(sd)
    eng = s.engine(dev)
    sets = []
    for k, n in enumerate((512, 200, 1024, 2048, 64, 700)):
        _, poses = reachable_poses(robot, n, 10 * idx + k)
        P, L = poses.to(dev), latents(n, lay.dim, 100 + 10 * idx + k).to(dev)
        eng.set_gemm_variant(182)
        ref = s.generate_ik_solutions(P, latent=L).clone()
        eng.set_gemm_variant(181)
        sets.append((P, L, ref))
    torch.cuda.synchronize()
    barrier.wait()
    worst, t0 = 0.0, time.perf_counter()
    for c in range(calls):
        P, L, ref = sets[c % len(sets)]
        out = s.generate_ik_solutions(P, latent=L)
        if c % 7 == 0:   # (mostly no synchronisation: launches queue up behind one another)
            worst = max(worst, float((out - ref).abs().max()))
    torch.cuda.synchronize()
    for P, L, ref in sets:
        worst = max(worst, float((s.generate_ik_solutions(P, latent=L) - ref).abs().max()))
    q.put({"tenant": idx, "calls": calls, "seconds": round(time.perf_counter() - t0, 2), "max_abs_diff_vs_row_owner_form": worst,
           "cluster_repairs": eng.cluster_repairs, "cluster_backoff_at_end": eng.cluster_backoff})


if __name__ == "__main__":
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(2), ctx.Queue()
    procs = [ctx.Process(target=tenant, args=(i, calls, barrier, q)) for i in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in sorted(res, key=lambda r: r["tenant"]):
        print(json.dumps(r))
    ok = all(r["max_abs_diff_vs_row_owner_form"] <= 1e-5 for r in res) and all(p.exitcode == 0 for p in procs)
    print(json.dumps({"ok": ok, "repairs_total": sum(r["cluster_repairs"] for r in res)}))
    sys.exit(0 if ok else 1)
