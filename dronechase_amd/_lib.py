"""ctypes binding of dronechase_amd/libthreatengage.so (the C ABI of include/threatengage.h).

There is no CPU fallback: if the HIP library is missing or fails to load, importing callers get a
RuntimeError that says how to build it."""
from __future__ import annotations

import ctypes as C
import os

from . import config as K

_LIB = None
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libthreatengage.so")

# every symbol include/threatengage.h declares (checked by tests/test_abi.py)
EXPORTS = (
    "te_config_default", "te_create", "te_destroy", "te_reset", "te_observe", "te_step", "te_random_actions",
    "te_state_words", "te_get_state", "te_set_state", "te_algorithmic_bytes_per_env_step", "te_profile_begin",
    "te_profile_end", "te_debug_stamps", "te_abi_version", "te_last_error", "te_step_stacked", "te_observe_stacked", "te_observe_ally", "te_set_ally_actions", "te_wingman_info", "te_calculate_rounds", "te_observe_wingman", "te_set_wingman_actions", "te_quad_preset", "te_step_students", "te_set_persistent_obs",
)


class TEError(RuntimeError):
    pass


def load() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m dronechase_amd.build` (needs hipcc, gfx950). "
            "dronechase_amd has no CPU or PyTorch fallback.")
    # ONE HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same soname as /opt/rocm's).  Loaded after torch,
    # libthreatengage.so binds to the copy torch already mapped and shares its device context; loaded BEFORE torch, it would pull
    # in /opt/rocm's copy, torch would then bring its own, and the first runtime reports "no ROCm-capable device" at te_create.
    import torch  # noqa: F401  (device memory / streams plumbing of every caller anyway)
    L = C.CDLL(LIB_PATH)
    vp, i32, u64 = C.c_void_p, C.c_int32, C.c_uint64
    L.te_last_error.restype = C.c_char_p
    L.te_abi_version.restype = C.c_int
    L.te_config_default.argtypes = [C.POINTER(K.Config), i32]
    L.te_quad_preset.argtypes = [C.POINTER(K.Config), i32]
    L.te_create.argtypes = [C.POINTER(K.Config), i32, C.POINTER(vp)]
    L.te_destroy.argtypes = [vp]
    L.te_destroy.restype = None
    L.te_reset.argtypes = [vp, vp, vp]
    L.te_observe.argtypes = [vp, vp, vp, vp, vp]
    L.te_step.argtypes = [vp] + [vp] * 10 + [vp]
    L.te_step_stacked.argtypes = [vp] + [vp] * 12 + [vp]
    L.te_observe_stacked.argtypes = [vp] + [vp] * 4 + [vp]
    L.te_step_students.argtypes = [vp] + [vp] * 8 + [vp]
    L.te_observe_ally.argtypes = [vp] + [vp] * 4 + [vp]
    L.te_set_ally_actions.argtypes = [vp, vp, vp]
    L.te_wingman_info.argtypes = [vp, vp, vp]
    L.te_calculate_rounds.argtypes = [C.c_int32, C.c_int32]
    L.te_observe_wingman.argtypes = [vp, C.c_int32] + [vp] * 4 + [vp]
    L.te_set_wingman_actions.argtypes = [vp, C.c_int32, vp, vp]
    L.te_random_actions.argtypes = [vp, vp, u64, u64, vp]
    L.te_state_words.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.te_get_state.argtypes = [vp, vp, C.c_size_t, vp]
    L.te_set_state.argtypes = [vp, vp, C.c_size_t, vp]
    L.te_algorithmic_bytes_per_env_step.argtypes = [C.POINTER(K.Config), C.POINTER(C.c_size_t)]
    L.te_debug_stamps.argtypes = [vp, C.POINTER(C.c_uint64), i32]
    L.te_profile_begin.argtypes = [vp, i32]
    L.te_set_persistent_obs.argtypes = [vp, i32]
    L.te_profile_end.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(i32)]
    if L.te_abi_version() != K.TE_ABI_VERSION:
        raise RuntimeError("libthreatengage.so ABI version differs from dronechase_amd.config")
    _LIB = L
    return L


def check(rc: int, what: str = "") -> None:
    if rc:
        raise TEError(f"{what}: {load().te_last_error().decode()}")


def default_config(task, **overrides) -> K.Config:
    """te_config_default(task) with keyword overrides (`quad__mass=...` reaches into te_quad_params; `quad_preset=1` swaps the
    whole quadrotor table with te_quad_preset before the other overrides apply)."""
    t = K.TASKS[task] if isinstance(task, str) else int(task)
    cfg = K.Config()
    rc = load().te_config_default(C.byref(cfg), t)
    if rc:
        raise ValueError(f"unknown task {task!r}")
    if "quad_preset" in overrides:
        if load().te_quad_preset(C.byref(cfg), int(overrides.pop("quad_preset"))):
            raise ValueError("unknown quad_preset")
    return K.apply_overrides(cfg, **overrides)


def algorithmic_bytes_per_env_step(cfg: K.Config) -> int:
    out = C.c_size_t()
    check(load().te_algorithmic_bytes_per_env_step(C.byref(cfg), C.byref(out)), "te_algorithmic_bytes_per_env_step")
    return int(out.value)
