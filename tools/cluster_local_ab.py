import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import MODEL_DESCRIPTIONS, hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import get_robot
MODEL = "panda__full__lp191_5.25m"
dev = torch.device("cuda", 0)
robot = get_robot(MODEL_DESCRIPTIONS[MODEL]["robot_name"]); hp = hparams_for(MODEL); layout = layout_from(hp, robot)
solver = IKFlowSolver(hp, robot); solver.load_state_dict_tensors(random_state_dict(layout, robot, seed=0)); eng = solver.engine(dev)
for rows in (129, 200, 256, 300, 512, 600, 768, 1024):
    q = torch.tensor(robot.sample_joint_angles(rows, 0.004, np.random.default_rng(0)), device=dev); p = robot.forward_kinematics(q)
    l = torch.randn(rows, layout.dim, generator=torch.Generator().manual_seed(1)).to(dev)
    outs = {}
    for rep in range(2):
        for v in (189, 190):
            eng.set_gemm_variant(v)
            for _ in range(5): o = solver.generate_ik_solutions(p, latent=l)
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            for _ in range(100): o = solver.generate_ik_solutions(p, latent=l)
            torch.cuda.synchronize(dev); ms = (time.perf_counter() - t0) * 10
            outs[v] = o
            print(rows, "spread" if v == 189 else "local ", f"{ms:.4f} ms", eng.plan(rows), "repairs", eng.cluster_repairs)
    print("   bitwise equal:", torch.equal(outs[189], outs[190]))
