"""Clearance of the approximate Panda capsule model (ikflow_amd.robots.PANDA_APPROX_CAPSULES) at named postures, colliding
fraction of uniformly random configurations, and the capsule pairs that contribute.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ikflow_amd.robots import Panda
r = Panda().use_approximate_collision_model()
folded, pairs = r._collision_model
print(len(folded), "capsules", len(pairs), "pairs")
home = torch.tensor([[0, -np.pi/4, 0, -3*np.pi/4, 0, np.pi/2, np.pi/4],
                     [0, 0, 0, -1.5708, 0, 1.8675, 0],
                     [0.5, 0.3, -0.4, -2.0, 0.2, 2.2, 1.0],
                     [0, 1.7, 0, -0.1, 0, 3.7, 0],
                     [0, -1.76, 0, -3.07, 0, 0.0, 0]], dtype=torch.float32, device="cuda")
d = r.self_collision_distances(home)
print("clearance at named configs", d.cpu().numpy())
q = torch.tensor(r.sample_joint_angles(20000, 0.0, np.random.default_rng(0)), device="cuda")
d = r.self_collision_distances(q)
print("random configs colliding fraction", float((d < 0).float().mean()), "min", float(d.min()))
# which pairs collide most
from ikflow_amd.engine import kinematics_engine_for
eng = kinematics_engine_for(r, "cuda")
for (a,b) in pairs:
    eng.set_collision_model(folded, [(a,b)])
    dd = eng.self_collision(q)[0]
    f = float((dd<0).float().mean())
    if f > 0.01: print("pair", a, b, "colliding fraction %.3f" % f, "home clearance %.3f" % float(eng.self_collision(home[:1])[0]))
