"""Generates the committed golden fixtures under tests/golden/.

The reference package cannot be imported in the build container (FrEIA 0.2 and jrl@2ba7c39 are absent),
so these vectors are produced by the in-repo oracle (oracle/flow_oracle.py, oracle/kinematics_oracle.py) on seeded
synthetic weights; they freeze the oracle's outputs so that any later drift (torch version, refactor) is caught, and
give the GPU parity tests committed input/output pairs.  The vectors the reference's OWN tests hold (Panda FK(0),
joint limits, pi geodesic, L2 known answer) and the numpy-MT19937 permutation tables are literals in
tests/test_oracle_golden.py.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import latents, panda_model, reachable_poses, tiny_model  # noqa: E402
from oracle import flow_oracle as fo  # noqa: E402
from oracle import kinematics_oracle as ko  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# README.md:54-56 / examples/example.py:69-73 target poses
README_POSES = torch.tensor(
    [[0.25, 0, 0.5, 1, 0, 0, 0], [0.35, 0, 0.5, 1, 0, 0, 0], [0.45, 0, 0.5, 1, 0, 0, 0], [0.55, 0, 0.5, 1, 0, 0, 0], [0.65, 0, 0.5, 1, 0, 0, 0]],
    dtype=torch.float32,
)


def main():
    torch.set_num_threads(1)
    # --- flow, TINY model (ikflow/model.py:45-48), weights = random_state_dict(seed 0) ---
    robot, hp, lay, sd = tiny_model(seed=0)
    n = 16
    poses = README_POSES[torch.arange(n) % 3]
    lat = torch.randn(n, lay.dim, generator=torch.Generator().manual_seed(0))
    out = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat, clamp=True)
    out_nc = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat, clamp=False)
    single = fo.generate_ik_solutions_torch(sd, lay, robot, README_POSES[0], lat, n=n)
    np.savez(os.path.join(HERE, "tiny_flow.npz"), poses=poses.numpy(), latent=lat.numpy(), q_clamped=out.numpy(),
             q_unclamped=out_nc.numpy(), q_single_pose0=single.numpy(), weights_seed=np.int64(0))
    # --- flow, full Panda architecture (BASELINE config 1: 3 README poses, batch 16) ---
    robot, hp, lay, sd = panda_model(seed=0)
    lat = torch.randn(n, lay.dim, generator=torch.Generator().manual_seed(0))
    poses = README_POSES[torch.arange(n) % 3]
    out = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat, clamp=True)
    out_nc = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat, clamp=False)
    np.savez(os.path.join(HERE, "panda_flow.npz"), poses=poses.numpy(), latent=lat.numpy(), q_clamped=out.numpy(),
             q_unclamped=out_nc.numpy(), weights_seed=np.int64(0))
    # --- kinematics: FK / pose error / LM step (fp64 twin) on 64 seeded Panda configurations ---
    q, poses = reachable_poses(robot, 64, seed=5)
    q0 = ko.clamp_to_joint_limits(robot, q + 0.1 * torch.randn(64, 7, generator=torch.Generator().manual_seed(6)))
    pe, re = ko.calculate_pose_error(robot, q0, poses)
    lm64 = ko.lm_step(robot, poses.double(), q0.double())
    np.savez(os.path.join(HERE, "panda_kinematics.npz"), q=q.numpy(), fk=ko.forward_kinematics(robot, q.double()).numpy(),
             q0=q0.numpy(), target=poses.numpy(), pos_err=pe.numpy(), rot_err=re.numpy(), lm_step_f64=lm64.numpy(),
             jac_f64=ko.jacobian(robot, q.double()).numpy())
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
