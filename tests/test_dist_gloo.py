"""World-size-2 (and 3) gloo tests of the multi-GPU sharding/gather logic (ikflow_amd/dist.py) on CPU.
The per-rank compute is a stand-in row-wise function - the engine itself only runs on the GPU - so what is covered is
exactly what differs between N=1 and N>1: block bounds, order preservation, ragged blocks, the single all-gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ikflow_amd.dist import gather_rows, shard_bounds, sharded_rows


def test_shard_bounds_cover_and_order():
    for n in (0, 1, 7, 8, 4096, 1000003):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(n, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [h - l for l, h in edges]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)  # same full inputs on every rank
        poses = torch.randn(n, 7, generator=g)
        latent = torch.randn(n, 7, generator=g)

        def compute(p, l):  # row-wise stand-in for solver.generate_ik_solutions
            assert p.shape[0] == l.shape[0] == shard_bounds(n, world, rank)[1] - shard_bounds(n, world, rank)[0]
            return torch.tanh(p) + 0.5 * l

        full = sharded_rows(compute, poses, latent)
        expect = torch.tanh(poses) + 0.5 * latent
        ok = bool(torch.equal(full, expect))
        lo, hi = shard_bounds(n, world, rank)
        ok = ok and bool(torch.equal(gather_rows(expect[lo:hi].contiguous(), n), expect))
        q.put((rank, ok, tuple(full.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 4096), (2, 4097), (3, 10), (2, 1)])
def test_sharded_rows_gloo(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok, f"rank {rank} gathered a wrong tensor"
        assert shape == (n, 7)


# ---- bench.py's own multi-rank plumbing (env handling, process group, ShardedStepper step / fence / gather, max over
# ranks, the one JSON line) under gloo on CPU tensors, launched exactly as the driver launches it ------------------------
@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("extra", [[], ["--million"], ["--global-batch", "4097"]])
def test_bench_dry_run_under_torch_distributed_run(extra, world):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--batch", "257", "--dist-dry-run"] + extra
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=root,
                       env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["value"] is None and out["n_gpus"] == world and out["gathered_ok"] is True
    assert out["scaling"] == ("strong" if extra else "weak")
    # strong modes: ceil(G / world) rows per rank; the line reports the batch that was ASKED for and the padding beside it
    per_rank = (1_000_000 + world - 1) // world if extra == ["--million"] else ((4097 + world - 1) // world if extra else 257)
    asked = 1_000_000 if extra == ["--million"] else (4097 if extra else world * 257)
    assert out["rows_per_rank"] == per_rank and out["global_batch"] == asked and out["padded_rows"] == world * per_rank - asked
    rc = out["rccl"]  # the self-proof every multi-rank line carries (here: gloo, CPU processes)
    assert rc["backend"] == "gloo" and rc["world_size"] == world and rc["gathered_shards_ok"] is True and not rc["is_rccl"]
    assert [r["rank"] for r in rc["ranks"]] == list(range(world)) and len({r["pid"] for r in rc["ranks"]}) == world
    assert 0 < rc["rank_elapsed_ms_min"] <= rc["rank_elapsed_ms_max"]


def test_bench_refuses_more_ranks_than_visible_gpus_with_a_message_naming_the_variable():
    """LOCAL_RANK beyond the visible GPUs (a launch with --nproc-per-node above HIP_VISIBLE_DEVICES) must fail with a message that says so,
    not with a HIP error from set_device; and the NUMA-affinity record never raises, whatever sysfs offers."""
    import bench

    rec = bench.gpu_numa_affinity("cpu", pin=False)
    assert rec["pinned"] is False and "note" in rec
    # all of the node's GPUs visible: cuda:LOCAL_RANK; the launcher narrowed this rank's view to its own GPU: cuda:0; anything else: a message
    assert bench.device_index_for(5, 5, 8, env={}) == 5
    assert bench.device_index_for(5, 5, 1, env={"HIP_VISIBLE_DEVICES": "5"}) == 0
    assert bench.device_index_for(3, 3, 1, env={"ROCR_VISIBLE_DEVICES": "3"}) == 0
    for n_vis, env in ((1, {}), (4, {"HIP_VISIBLE_DEVICES": "0,1,2,3"}), (0, {})):
        with pytest.raises(AssertionError, match="HIP_VISIBLE_DEVICES"):
            bench.device_index_for(6, 6, n_vis, env=env)
    assert bench.asked_global_batch("strong", 8, 513, 4097) == 4097 and bench.asked_global_batch("weak", 8, 4096, 0) == 32768
    assert bench.asked_global_batch("million", 8, 125000, 0) == 1_000_000


class _FakeSolver:
    """Row-wise stand-in with IKFlowSolver's two call signatures (the engine only runs on the GPU)."""

    network_width = 7

    def generate_ik_solutions(self, y, n=None, latent=None, **kw):
        return torch.tanh(y) + 0.5 * latent

    def generate_exact_ik_solutions(self, target_poses, **kw):
        self.exact_calls = getattr(self, "exact_calls", 0) + 1
        assert target_poses.shape[0] > 0, "the solver is never called on an empty shard"
        sol = torch.tanh(target_poses)
        valid = target_poses[:, 0] > 0
        return torch.where(valid[:, None], sol, torch.zeros_like(sol)), valid


def _worker_solver(rank, world, port, n, q):
    from ikflow_amd.dist import sharded_generate_exact_ik_solutions, sharded_generate_ik_solutions

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        poses = torch.randn(n, 7, generator=torch.Generator().manual_seed(0))
        s = _FakeSolver()
        # exact: solutions and valid flags travel in one collective; order = input order
        sol, valid = sharded_generate_exact_ik_solutions(s, poses)
        want_sol, want_valid = s.generate_exact_ik_solutions(poses)
        ok = bool(torch.equal(sol, want_sol)) and bool(torch.equal(valid, want_valid)) and valid.dtype == torch.bool
        # approx with latent=None: the FULL latent is drawn on every rank (same seed everywhere, as launchers set it) and
        # sliced - equal to the single-process call on that seed, and no two shards share a latent block
        torch.manual_seed(123)
        got = sharded_generate_ik_solutions(s, poses)
        torch.manual_seed(123)
        full_latent = 1.0 * torch.randn((n, 7))
        ok = ok and bool(torch.equal(got, torch.tanh(poses) + 0.5 * full_latent))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _worker_from_shard(rank, world, port, n, q):
    from ikflow_amd.dist import (ShardedStepper, draw_latent_shard, gather_blocks, sharded_generate_exact_ik_solutions_from_shard,
                                 sharded_generate_ik_solutions_from_shard)

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank holds ONLY its own block, of a size only it knows (rank r: n + 3 r rows; n = 0: rank 0 has nothing at all)
        rows = [n + 3 * r for r in range(world)]
        blocks = [torch.randn(rows[r], 7, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]   # (the expectation only)
        mine = blocks[rank]
        s = _FakeSolver()
        s.ndof = 7
        base = 77
        lat = [draw_latent_shard(rows[r], 7, "cpu", base, r) for r in range(world)]
        want = torch.cat([torch.tanh(blocks[r]) + 0.5 * lat[r] for r in range(world)], dim=0)
        got = sharded_generate_ik_solutions_from_shard(s, mine, base_seed=base)                     # sizes exchanged
        ok = bool(torch.equal(got, want))
        got = sharded_generate_ik_solutions_from_shard(s, mine, base_seed=base, counts=rows)       # sizes known: one collective
        ok = ok and bool(torch.equal(got, want))
        given = torch.full((rows[rank], 7), float(rank))
        got = sharded_generate_ik_solutions_from_shard(s, mine, given, counts=rows)
        ok = ok and bool(torch.equal(got, torch.cat([torch.tanh(blocks[r]) + 0.5 * torch.full((rows[r], 7), float(r)) for r in range(world)], dim=0)))
        local = sharded_generate_ik_solutions_from_shard(s, mine, base_seed=base, gather=False)
        ok = ok and bool(torch.equal(local, torch.tanh(mine) + 0.5 * lat[rank]))
        # no two ranks draw the same block, and a rank's block does not depend on the world size
        ok = ok and all(not torch.equal(lat[0][: min(rows)], lat[r][: min(rows)]) for r in range(1, world) if min(rows) > 0)
        # exact: also when a rank has no rows at all (r06: it still enters the collective, with empty solutions / flags)
        calls_before = getattr(s, "exact_calls", 0)
        sol, valid = sharded_generate_exact_ik_solutions_from_shard(s, mine)
        ws, wv = s.generate_exact_ik_solutions(torch.cat(blocks, dim=0))
        ok = ok and bool(torch.equal(sol, ws)) and bool(torch.equal(valid, wv)) and valid.dtype == torch.bool and sol.shape == (sum(rows), 7)
        if rows[rank] == 0:
            ok = ok and getattr(s, "exact_calls", 0) == calls_before + 1   # (only the expectation above called the solver: the empty shard did not)
        ok = ok and bool(torch.equal(gather_blocks(mine), torch.cat(blocks, dim=0)))
        # the stepper: equal shards, the gather double buffered
        B = 5
        shard = torch.randn(B, 7, generator=torch.Generator().manual_seed(rank))
        st = ShardedStepper(lambda: torch.tanh(shard), world, rank, B, 7, "cpu", True)
        ok = ok and st.last_gathered() is None   # (nothing gathered before the first step)
        for _ in range(3):
            sol = st.step()
        st.fence()
        full = st.last_gathered()
        ok = ok and bool(torch.equal(full[rank * B : (rank + 1) * B], sol))
        ok = ok and all(bool(torch.equal(full[r * B : (r + 1) * B], torch.tanh(torch.randn(B, 7, generator=torch.Generator().manual_seed(r))))) for r in range(world))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 1001), (3, 64), (3, 0)])
def test_shard_in_forms_gloo(world, n):
    """SURVEY 8(e) "Determinism: per-rank seeds = base_seed + rank, generated in place": the shard-in entry points take a rank's OWN block
    (sizes exchanged or given; a rank may have no rows), draw its latents in place, and gather in rank order - nothing of size O(n) is
    replicated but the result.  ShardedStepper (what bench.py steps through) gathers equal shards double-buffered."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_from_shard, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


@pytest.mark.parametrize("world,n", [(2, 1001), (3, 64)])
def test_sharded_solver_entry_points_gloo(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_solver, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def _worker_real_solver(rank, world, port, n, q, backend="gloo"):
    """Both ranks on cuda:0 of the one-GPU box, gloo in place of RCCL: the real IKFlowSolver behind the sharded entry points.
    With backend="nccl" (world size 1 on a one-GPU box) the same calls run through RCCL itself."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import latents, reachable_poses, tiny_model
    from ikflow_amd.dist import sharded_generate_exact_ik_solutions, sharded_generate_ik_solutions
    from ikflow_amd.ikflow_solver import IKFlowSolver

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        import gloo_staging

        gloo_staging.install()   # gloo never sees a device tensor (tests/gloo_staging.py)
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        assert dist.get_backend() == backend and dist.get_world_size() == world
        dev = "cuda:0"
        robot, hp, lay, sd = tiny_model(seed=4)
        s = IKFlowSolver(hp, robot)
        s.load_state_dict_tensors(sd)
        q_true, poses = reachable_poses(robot, n, 5)
        poses, lat = poses.to(dev), latents(n, lay.dim, 6).to(dev)
        single = s.generate_ik_solutions(poses, latent=lat)
        # a shard of n/2 rows may take another tile shape than the n-row call (other summation order): equal to fp32 rounding
        close = lambda a, b: float((a - b).abs().max()) <= 1e-5
        ok = [close(sharded_generate_ik_solutions(s, poses, latent=lat), single)]
        torch.manual_seed(77)
        torch.cuda.manual_seed(77)
        drawn = sharded_generate_ik_solutions(s, poses)
        torch.manual_seed(77)
        torch.cuda.manual_seed(77)
        ok.append(close(drawn, s.generate_ik_solutions(poses)))
        # exact IK: retry rounds are per pose, shards never interact; every valid row meets the thresholds, invalid rows are zero
        sol, valid = sharded_generate_exact_ik_solutions(s, poses, pos_error_threshold=5e-3, rot_error_threshold=0.1, repeat_counts=(1, 4))
        ok.append(sol.shape == (n, robot.ndof) and valid.dtype == torch.bool and valid.shape == (n,))
        ok.append(not bool(sol[~valid].any()))
        if bool(valid.any()):
            from ikflow_amd.evaluation_utils import pose_errors

            pe, re = pose_errors(robot.forward_kinematics(sol[valid]), poses[valid])
            ok.append(float(pe.max()) <= 5e-3 and float(re.max()) <= 0.1)
        # the shard-in form on the device: this rank's own block only, latents drawn in place from base_seed + rank
        from ikflow_amd.dist import draw_latent_shard, shard_bounds, sharded_generate_ik_solutions_from_shard

        lo, hi = shard_bounds(n, world, rank)
        got = sharded_generate_ik_solutions_from_shard(s, poses[lo:hi].contiguous(), base_seed=31)
        blocks = []
        for r in range(world):
            l, h = shard_bounds(n, world, r)
            blocks.append(s.generate_ik_solutions(poses[l:h].contiguous(), latent=draw_latent_shard(h - l, lay.dim, dev, 31, r)))
        ok.append(got.shape == (n, robot.ndof) and close(got, torch.cat(blocks, dim=0)))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_entry_points_with_the_real_solver_two_ranks_one_gpu():
    world, n = 2, 1501
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_real_solver, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(all(ok) for _, ok in res), res


@pytest.mark.gpu
def test_sharded_entry_points_through_rccl_world_size_one():
    """ikflow_amd/dist.py under the backend the 8-GPU node will use: init_process_group("nccl", device_id=...) = RCCL, world size 1 (what a
    one-GPU box allows): the padded all_gather_into_tensor of the approximate path (given latents and drawn latents), the packed
    solutions + valid-flag gather of the exact path, destroy_process_group.  n is odd so the padding branch is taken."""
    world, n = 1, 1501
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_real_solver, args=(0, world, _free_port(), n, q, "nccl"))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert all(res[1]), res


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [[], ["--global-batch", "4096"], ["--million"]])
def test_bench_under_torch_distributed_run_with_rccl_world_size_one(mode):
    """bench.py launched exactly as the driver launches it for N > 1 - `python -m torch.distributed.run ... bench.py --gpus N` - with the
    real engine and the real backend (nccl = RCCL), at the world size a one-GPU box allows: RCCL init with device_id, the all-gather on its
    side stream behind an event, record_stream, fence + barrier, the max-over-ranks all_reduce, the `rccl` proof object, destroy."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("IKF_BENCH_TEST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
           "--no-cells", "--no-split-extra", "--no-cpu-baseline", "--no-live-pmc"] + mode
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=root, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = lines[0]
    rc = out["rccl"]
    assert rc["backend"] == "nccl" and rc["is_rccl"] and rc["world_size"] == 1 and rc["gathered_shards_ok"] is True
    assert rc["distinct_devices"] == 1 and len(rc["ranks"]) == 1 and rc["ranks"][0]["rank"] == 0 and rc["rccl_version"]
    assert rc["rank_elapsed_ms_min"] <= rc["rank_elapsed_ms_max"] and "test_backend" not in out
    want_rows = 1_000_000 if mode == ["--million"] else 4096
    assert out["n_gpus"] == 1 and out["config"]["global_batch"] == want_rows and out["value"] > 0
    assert out["scaling"] == ("strong" if mode else "weak") and out["config"]["scaling_mode"] == ("million" if mode == ["--million"] else "strong" if mode else "weak")
    assert abs(out["value"] - want_rows * 4 / (out["ms_per_step"] * 4e-3)) <= 1e-6 * out["value"]
    if not mode:  # the default multi-rank run also measures both strong-scaling readings in the same launch
        sm = out["extra"]["scaling_modes"]
        assert sm["strong_global_batch_4096"]["global_batch"] == 4096 and sm["strong_global_batch_4096"]["gathered_shards_ok"]
        assert sm["million_poses_config5"]["global_batch"] == 1_000_000 and sm["million_poses_config5"]["gathered_shards_ok"]
        assert sm["million_poses_config5"]["value"] > 5e5
