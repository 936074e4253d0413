"""ctypes loader for libdiral_env.so (the C-ABI of include/diral_env.h).

There is NO CPU fallback: if the HIP library is missing or fails to load, every
product entry point raises.  (The CPU restatement under oracle/ is test
infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes
import os

from .config import DiralCfg

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DIRAL_LIB") or os.path.join(HERE, "libdiral_env.so")   # DIRAL_LIB: tuning variants

# every symbol include/diral_env.h declares
SYMBOLS = [
    "diral_cfg_defaults", "diral_env_state_space", "diral_env_validate", "diral_env_strerror",
    "diral_env_abi_version", "diral_env_create", "diral_env_destroy", "diral_env_hbm_bytes",
    "diral_env_reset", "diral_env_step", "diral_env_observe", "diral_env_update_velocity",
    "diral_env_sample", "diral_env_info_age", "diral_env_export_state", "diral_env_import_state",
    "diral_env_metrics", "diral_env_check", "diral_env_last_hip_error", "diral_sps_step", "diral_sps_init",
    "diral_env_set_trace", "diral_env_set_option", "diral_env_last_kernel",
    "diral_sps_window_from_chobs", "diral_sps_step_chobs", "diral_driver_shape",
    "diral_env_export_entries", "diral_env_import_entries",
    "diral_env_set_clock", "diral_clock_add", "diral_sps_step_chobs_clocked", "diral_env_step_policy",
    "diral_env_set_capture_rotation", "diral_env_align_phase",
    "diral_env_export_prev_obs", "diral_env_import_prev_obs", "diral_env_prefill",
]

_lib = None


class DiralLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64; it must be in the process BEFORE
    # libdiral_env.so is dlopen'ed so both bind to the SAME HIP runtime (same
    # SONAME -> the loader reuses it).  Loading ours first would pull in
    # /opt/rocm's copy and torch's device pointers would belong to another runtime.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise DiralLibraryError(
            "%s is missing - build it with `python -m diral_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # e.g. libamdhip64 not present
        raise DiralLibraryError("cannot load %s: %s" % (LIB_PATH, exc)) from exc
    P, I, I64, U64, D = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_double
    CFG = ctypes.POINTER(DiralCfg)
    sig = {
        "diral_cfg_defaults": (None, [CFG]),
        "diral_env_state_space": (I, [CFG]),
        "diral_env_validate": (I, [CFG]),
        "diral_env_strerror": (ctypes.c_char_p, [I]),
        "diral_env_abi_version": (I, []),
        "diral_env_create": (I, [CFG, I, I, ctypes.POINTER(P)]),
        "diral_env_destroy": (I, [P]),
        "diral_env_hbm_bytes": (I64, [P]),
        "diral_env_reset": (I, [P, P, P, P, U64, P]),
        "diral_env_step": (I, [P, I, P, I64, P, P, P, P, I, D, D, P]),
        "diral_env_observe": (I, [P, P, P, P, P, I, D, D, P]),
        "diral_env_update_velocity": (I, [P, P, U64, P]),
        "diral_env_sample": (I, [P, P, U64, P]),
        "diral_env_info_age": (I, [P, I64, P, P]),
        "diral_env_export_state": (I, [P] + [P] * 8 + [P]),
        "diral_env_import_state": (I, [P] + [P] * 7 + [P]),
        "diral_env_export_entries": (I, [P, P, P]),
        "diral_env_import_entries": (I, [P, P, P]),
        "diral_env_metrics": (I, [P, P, I, P]),
        "diral_env_check": (I, [P, P]),
        "diral_env_last_hip_error": (ctypes.c_char_p, [P]),
        "diral_sps_step": (I, [I, I, P, P, P, D, D, D, P, P, P, U64, P, P]),
        "diral_sps_init": (I, [I, I, P, P, U64, P]),
        "diral_env_set_trace": (I, [P, P, I, I, P]),
        "diral_env_set_option": (I, [P, I, I64]),
        "diral_env_last_kernel": (I, [P]),
        "diral_sps_window_from_chobs": (I, [I, I, P, I, P, P, P]),
        "diral_sps_step_chobs": (I, [I, I, P, I, P, P, P, D, D, D, P, P, P, U64, P, P]),
        "diral_driver_shape": (I, [I, I, I, P, I, P, P, P, P, P, I, I, D, P, P, P, P, P, P]),
        "diral_env_set_clock": (I, [P, P]),
        "diral_clock_add": (I, [P, I64, P]),
        "diral_sps_step_chobs_clocked": (I, [I, I, P, I, P, P, P, D, D, D, U64, P, P, P]),
        "diral_env_step_policy": (I, [P, I, P, I64, P, P, P, P, I, P, P]),
        "diral_env_set_capture_rotation": (I, [P, I, P]),
        "diral_env_align_phase": (I, [P, I, P]),
        "diral_env_export_prev_obs": (I, [P, P, P]),
        "diral_env_import_prev_obs": (I, [P, P, P]),
        "diral_env_prefill": (I, [P, P, I, U64, P, I, P, P, P, D, D, P]),
    }
    for name in SYMBOLS:
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise DiralLibraryError("libdiral_env.so lacks symbol %s" % name) from exc
        fn.restype, fn.argtypes = sig[name]
    from .config import ABI_VERSION
    got = int(lib.diral_env_abi_version())
    if got != ABI_VERSION:
        raise DiralLibraryError("%s speaks ABI %d, this package binds ABI %d (include/diral_env.h): rebuild it with "
                                "`python -m diral_amd.build`" % (LIB_PATH, got, ABI_VERSION))
    _lib = lib
    return lib


def strerror(status: int) -> str:
    return load().diral_env_strerror(status).decode()
