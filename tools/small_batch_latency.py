import sys, time, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import panda_model, reachable_poses, latents
from ikflow_amd.ikflow_solver import IKFlowSolver
robot, hp, lay, sd = panda_model()
s = IKFlowSolver(hp, robot); s.load_state_dict_tensors(sd)
for n in (1, 16, 32, 64, 128, 256):
    _, poses = reachable_poses(robot, n, 1); poses = poses.to("cuda:0"); lat = latents(n, lay.dim, 2).to("cuda:0")
    for _ in range(20): s.generate_ik_solutions(poses, n=(1 if n==1 else None), latent=lat)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): s.generate_ik_solutions(poses, n=(1 if n==1 else None), latent=lat)
    torch.cuda.synchronize(); print(n, "%.3f ms" % ((time.perf_counter() - t0) * 5))
