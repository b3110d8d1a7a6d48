"""ctypes binding of libikflow_amd.so (the C-ABI declared in include/ikflow_amd.h).

There is no fallback: if the in-tree shared library is missing or does not load, importing the binding raises.
"""
from __future__ import annotations

import ctypes as C
import os

from ikflow_amd import build as _build

IKF_ABI_VERSION = 3
IKF_MAX_DOF = 8
IKF_MAX_DIM = 16
IKF_MAX_ROUNDS = 8
IKF_MAX_CAPSULES = 24

IKF_OK = 0
IKF_ERR_NULL_POINTER = 1
IKF_ERR_BAD_SHAPE = 2
IKF_ERR_NOT_LOADED = 3
IKF_ERR_MISSING_TENSOR = 4
IKF_ERR_HIP = 5
IKF_ERR_NO_DEVICE = 6
IKF_ERR_BAD_ARGUMENT = 7


class ikf_joint(C.Structure):
    _fields_ = [("kind", C.c_int32), ("axis", C.c_float * 3), ("pre", C.c_float * 12)]


class ikf_model_desc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("nb_nodes", C.c_int32),
        ("dim", C.c_int32),
        ("dim_cond", C.c_int32),
        ("width", C.c_int32),
        ("n_hidden", C.c_int32),
        ("clamp", C.c_float),
        ("leaky_slope", C.c_float),
        ("ndof", C.c_int32),
        ("joint_lo", C.c_float * IKF_MAX_DOF),
        ("joint_hi", C.c_float * IKF_MAX_DOF),
        ("chain", ikf_joint * IKF_MAX_DOF),
        ("tool", C.c_float * 12),
        ("sigmoid_on_output", C.c_int32),
    ]


class ikf_capsule(C.Structure):
    _fields_ = [("frame", C.c_int32), ("p0", C.c_float * 3), ("p1", C.c_float * 3), ("radius", C.c_float)]


class ikf_tensor(C.Structure):
    _fields_ = [
        ("name", C.c_char_p),
        ("h_data", C.c_void_p),
        ("dtype", C.c_int32),
        ("ndim", C.c_int32),
        ("shape", C.c_int64 * 4),
    ]


LATENT_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int)
SEED_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int)

# name -> (restype, argtypes); every symbol include/ikflow_amd.h declares
SIGNATURES = {
    "ikf_create": (C.c_int, [C.POINTER(ikf_model_desc), C.c_int, C.POINTER(C.c_void_p)]),
    "ikf_destroy": (None, [C.c_void_p]),
    "ikf_last_error": (C.c_char_p, []),
    "ikf_abi_version": (C.c_int, []),
    "ikf_load_weights": (C.c_int, [C.c_void_p, C.POINTER(ikf_tensor), C.c_int]),
    "ikf_weights_loaded": (C.c_int, [C.c_void_p]),
    "ikf_reserve": (C.c_int, [C.c_void_p, C.c_int64]),
    "ikf_reserve_exact": (C.c_int, [C.c_void_p, C.c_int64, C.c_int]),
    "ikf_set_exact_upfront_rows": (C.c_int, [C.c_void_p, C.c_int64]),
    "ikf_generate_approx": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_void_p],
    ),
    "ikf_forward_kinematics": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ikf_pose_error": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ikf_lm_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ikf_jacobian": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ikf_clamp_to_joint_limits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ikf_joint_limits_exceeded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ikf_set_collision_model": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "ikf_self_collision": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ikf_pose_distance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ikf_limits_exceeded": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ikf_generate_exact": (
        C.c_int,
        [
            C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_float, C.c_float,
            LATENT_FN, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p,
        ],
    ),
    "ikf_generate_exact_seeded": (
        C.c_int,
        [
            C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_float, C.c_float,
            SEED_FN, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p,
        ],
    ),
    "ikf_refine_exact": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "ikf_time_gemm": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_float), C.c_void_p]),
    "ikf_profile_begin": (C.c_int, [C.c_void_p]),
    "ikf_profile_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_void_p]),
    "ikf_profile_event_overhead_ms": (C.c_double, [C.c_void_p]),
    "ikf_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "ikf_get_precision": (C.c_int, [C.c_void_p]),
    "ikf_set_lm_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "ikf_get_lm_precision": (C.c_int, [C.c_void_p]),
    "ikf_set_split_guard": (C.c_int, [C.c_void_p, C.c_int]),
    "ikf_split_fallback_count": (C.c_int64, [C.c_void_p]),
    "ikf_split_overflow_pending": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ikf_split_kernel_name": (C.c_char_p, []),
    "ikf_dominant_kernel_name": (C.c_char_p, []),
    "ikf_dominant_kernel_for": (C.c_char_p, [C.c_void_p, C.c_int64]),
    "ikf_cluster_repairs": (C.c_int64, [C.c_void_p]),
    "ikf_cluster_backoff": (C.c_int64, [C.c_void_p]),
    "ikf_load_time_ms": (C.c_double, [C.c_void_p]),
    "ikf_frag_image_time_ms": (C.c_double, [C.c_void_p]),
    "ikf_probes_build": (C.c_int, []),
    "ikf_plan_describe": (C.c_int, [C.c_void_p, C.c_int64, C.c_char_p, C.c_int]),
    "ikf_cluster_local": (C.c_int, [C.c_void_p]),
    "ikf_plan_describe_for": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "ikf_set_gemm_variant": (C.c_int, [C.c_void_p, C.c_int]),
}

LIB_PATH = _build.LIB_PATH
_libs = {}


def load(flavour: str = "") -> C.CDLL:
    """dlopen the in-tree library and bind every declared symbol. Raises if anything is missing.
    flavour "probes": lib/libikflow_amd_probes.so (-DIKF_PROBES: the product plus the priced-and-rejected forms of rounds 2 - 3; tests and
    tools only - both flavours may be loaded in one process, each handle belongs to the library that created it)."""
    if flavour in _libs:
        return _libs[flavour]
    LIB_PATH = _build.lib_path(flavour)
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP engine has not been built (run `python -m ikflow_amd.build{' --probes' if flavour else ''}`). "
            "ikflow_amd has no CPU path."
        )
    # torch FIRST: it ships its own copy of the HIP runtime (torch/lib/libamdhip64.so); if this library were loaded before
    # torch, the loader would bind it to /opt/rocm's copy and the process would hold two HIP runtimes, the second of which sees
    # no device ("ikf_create: no HIP device visible" although torch.cuda.is_available()).  With torch's copy already mapped, this
    # library's libamdhip64.so.7 dependency resolves to it.  (A C / C++ client without torch links /opt/rocm's copy alone.)
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.ikf_abi_version()
    if got != IKF_ABI_VERSION:
        raise ImportError(f"libikflow_amd.so ABI {got} != binding ABI {IKF_ABI_VERSION}; rebuild the library")
    if bool(lib.ikf_probes_build()) != (flavour == "probes"):
        raise ImportError(f"{LIB_PATH} is not the {'probes' if flavour else 'product'} flavour; rebuild it")
    _libs[flavour] = lib
    return lib


def last_error(lib: C.CDLL = None) -> str:
    return (lib or load()).ikf_last_error().decode("utf-8", "replace")
