"""Builds libikflow_amd.so (HIP, gfx950 only) in-tree with hipcc.  `python -m ikflow_amd.build [--force] [--probes]`.

Two flavours of the same sources: the product library (lib/libikflow_amd.so) and the probes library (lib/libikflow_amd_probes.so,
-DIKF_PROBES) that additionally carries the priced-and-rejected forms of rounds 2 - 3 (in-launch entry phase, one-launch chain for
<= 128 rows, tile configurations 5 / 7 / 11) for the tests and tools that keep their measurements reproducible."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from typing import List

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libikflow_amd.so")
PROBES_LIB_PATH = os.path.join(LIB_DIR, "libikflow_amd_probes.so")
SOURCES = ["flow_kernels.hip", "flow_fused.hip", "flow_rowowner.hip", "flow_split.hip", "kin_kernels.hip", "ikf_api.hip"]
HEADERS = [os.path.join(CSRC, "ikf_internal.h"), os.path.join(CSRC, "kin_math.h"), os.path.join(CSRC, "flow_split_dma.inc"), os.path.join(CSRC, "flow_fused_probes.inc"), os.path.join(_HERE, "..", "include", "ikflow_amd.h"),
           os.path.join(_HERE, "..", "include", "ikflow_amd_debug.h")]
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("IKF_HIPCC_FLAGS", "").split()  # probes only (e.g. -DIKF_NO_RANGE_FLAG); the shipped library is built without


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libikflow_amd.so cannot be built (no prebuilt library in-tree either)")


def lib_path(flavour: str = "") -> str:
    return PROBES_LIB_PATH if flavour == "probes" else LIB_PATH


def is_stale(flavour: str = "") -> bool:
    path = lib_path(flavour)
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, flavour: str = "") -> str:
    """Compile every HIP translation unit for gfx950 and link the shared library of that flavour ("" or "probes"). Returns its path."""
    out_path = lib_path(flavour)
    if not force and not is_stale(flavour):
        return out_path
    hipcc = _hipcc()
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "probes") if flavour == "probes" else LIB_DIR
    os.makedirs(obj_dir, exist_ok=True)
    flavour_flags = ["-DIKF_PROBES"] if flavour == "probes" else []
    objs: List[str] = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        extra = os.environ.get("IKF_HIPCC_FLAGS_" + src.split(".")[0].upper(), "").split()  # probes: flags for ONE unit, e.g. IKF_HIPCC_FLAGS_FLOW_FUSED=-DIKF_TRACE
        cmd = [hipcc] + FLAGS + flavour_flags + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + out)
        if verbose and out.strip():
            print(out)
    # --no-undefined: a kernel template whose host stub was dropped must fail the build, not the first launch
    link = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,--no-undefined", "-o", out_path] + objs
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + " ".join(link) + "\n" + r.stdout)
    return out_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, flavour="probes" if "--probes" in sys.argv else ""))
