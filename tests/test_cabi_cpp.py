"""The C-ABI used from plain C++ (examples/cabi_demo.cpp): no Python, no torch on the calling side.  Python only writes the
input file and checks the output file against the oracle."""
import ctypes
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

from helpers import O, latents, reachable_poses, tiny_model
from ikflow_amd import _lib
from ikflow_amd.engine import _make_desc
from oracle import flow_oracle as fo
from oracle import kinematics_oracle as ko

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_ikfbin(path, layout, robot, sd, poses, latent):
    desc = _make_desc(layout, robot)
    with open(path, "wb") as f:
        f.write(bytes(desc))
        items = [(k, np.ascontiguousarray(v)) for k, v in sd.items() if v.dtype.kind in "fiu"]
        f.write(struct.pack("<i", len(items)))
        for k, v in items:
            v = v.astype(np.float32) if v.dtype.kind == "f" else v.astype(np.int64)
            name = k.encode()
            shape = list(v.shape) + [0] * (4 - v.ndim)
            f.write(struct.pack("<i", len(name)) + name + struct.pack("<ii4q", 0 if v.dtype.kind == "f" else 1, v.ndim, *shape))
            f.write(v.tobytes())
        f.write(struct.pack("<q", poses.shape[0]))
        f.write(np.ascontiguousarray(poses, dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(latent, dtype=np.float32).tobytes())


@pytest.mark.gpu
def test_cpp_client_of_the_cabi(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "cabi_demo")
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", os.path.join(ROOT, "examples", "cabi_demo.cpp"),
           "-I" + os.path.join(ROOT, "include"), "-L" + lib_dir, "-likflow_amd", "-Wl,-rpath," + lib_dir, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    robot, hp, lay, sd = tiny_model(seed=4)
    n = 200
    _, poses = reachable_poses(robot, n, 9)
    lat = latents(n, lay.dim, 10)
    write_ikfbin(str(tmp_path / "m.ikfbin"), lay, robot, sd, poses.numpy(), lat.numpy())
    r = subprocess.run([exe, str(tmp_path / "m.ikfbin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(str(tmp_path / "out.bin"), dtype=np.float32)
    q = out[: n * 7].reshape(n, 7)
    fk = out[n * 7 : n * 14].reshape(n, 7)
    pe, re = out[n * 14 : n * 15], out[n * 15 : n * 16]
    ref = fo.generate_ik_solutions_torch(sd, lay, robot, poses, lat)
    assert np.abs(q - ref.numpy()).max() <= 1e-5
    fk_ref = ko.forward_kinematics(robot, torch.from_numpy(q))
    assert np.abs(fk[:, :3] - fk_ref[:, :3].numpy()).max() <= 2e-6
    pe_ref, re_ref = ko.calculate_pose_error(robot, torch.from_numpy(q), poses)
    assert np.abs(pe - pe_ref.numpy()).max() <= 2e-6 and np.abs(re - re_ref.numpy()).max() <= 3e-5
    # exact IK from C++: ikf_refine_exact on the approximate solutions (the program itself checks that ikf_generate_exact with a C
    # latent callback gives the identical result); oracle = one _generate_exact_ik_solutions round on the SAME seeds, LM in fp64
    q2 = out[n * 16 : n * 23].reshape(n, 7)
    valid = out[n * 23 : n * 24] > 0.5
    assert "exact IK" in r.stdout
    ref_sol, ref_valid = ko.exact_round(robot, torch.from_numpy(q.copy()), poses, 1, 0.05, 0.3, lm_dtype=torch.float64)
    assert (torch.from_numpy(valid) == ref_valid).float().mean().item() >= 0.98
    both = torch.from_numpy(valid) & ref_valid
    assert 0 < int(both.sum()) and np.abs(q2[both.numpy()] - ref_sol[both].numpy()).max() <= 5e-6
    assert not q2[~valid].any()


def test_ikfbin_descriptor_layout_matches_header():
    """The ctypes mirror of ikf_model_desc must have the C struct's size (the C++ client freads it verbatim)."""
    assert ctypes.sizeof(_lib.ikf_model_desc) == 664 and ctypes.sizeof(_lib.ikf_tensor) == 56
