import time, numpy as np, torch
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import MODEL_DESCRIPTIONS, hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import get_robot
MODEL = "panda__full__lp191_5.25m"
dev = torch.device("cuda", 0)
robot = get_robot("panda"); hp = hparams_for(MODEL); layout = layout_from(hp, robot)
solver = IKFlowSolver(hp, robot); solver.load_state_dict_tensors(random_state_dict(layout, robot, seed=0))
eng = solver.engine(dev)
B = 4096
q = torch.tensor(robot.sample_joint_angles(B, 0.004, np.random.default_rng(0)), device=dev)
p = robot.forward_kinematics(q)
l = torch.randn(B, layout.dim, generator=torch.Generator().manual_seed(1)).to(dev)
def loop(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): solver.generate_ik_solutions(p, latent=l)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(10): solver.generate_ik_solutions(p, latent=l)
print("loop 50:", loop(50))
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for a, b in evs:
    a.record(); solver.generate_ik_solutions(p, latent=l); b.record()
torch.cuda.synchronize()
print("torch event pairs:", sum(a.elapsed_time(b) for a, b in evs) / len(evs), "first->last per launch", evs[0][0].elapsed_time(evs[-1][1]) / len(evs))
eng.profile_begin()
for _ in range(20): solver.generate_ik_solutions(p, latent=l)
n, ms = eng.profile_end()
print("engine profile:", n, ms / n)
eng.profile_begin()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): solver.generate_ik_solutions(p, latent=l)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
n, ms = eng.profile_end()
print("engine profile:", n, ms / n, "wall per step in the same loop", dt)
print("loop 50:", loop(50))
