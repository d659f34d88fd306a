"""ctypes binding of libspacegym_hip.so (include/spacegym.h).  There is no CPU path: if the HIP library is
missing or no GPU is visible, creating an engine raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPACEGYM_LIB points at another build of the SAME library (e.g. the stamped diagnostic build); never at a CPU path
LIB_PATH = os.environ.get("SPACEGYM_LIB") or os.path.join(_HERE, "lib", "libspacegym_hip.so")
_lib = None


class SgConfig(C.Structure):
    _fields_ = [("env_id", C.c_char * 64), ("num_envs", C.c_int64), ("seed", C.c_uint64),
                ("env_index_base", C.c_uint32), ("max_episode_steps", C.c_int32), ("auto_reset", C.c_int32),
                ("steering", C.c_int32)]


class SgParams(C.Structure):
    """sg_params (include/spacegym.h): the reference's constructor kwargs; NaN / -1 keeps the id's registered value"""
    _fields_ = [("struct_size", C.c_uint32), ("n_planets", C.c_int32), ("randomize", C.c_int32), ("reserved", C.c_int32),
                ("goal_vel_reward_scale", C.c_double), ("safety_reward_scale", C.c_double), ("goal_sparse_reward", C.c_double),
                ("survival_reward_scale", C.c_double), ("danger_zone", C.c_double),
                ("ref_orbit_a", C.c_double), ("ref_orbit_eccentricity", C.c_double), ("ref_orbit_angle", C.c_double),
                ("numerator_C", C.c_double), ("rad_penalty_C", C.c_double), ("act_penalty_C", C.c_double), ("step_size", C.c_double),
                ("ship_moi", C.c_double), ("max_engine_force", C.c_double)]


class SgTerminalList(C.Structure):
    _fields_ = [("count", C.c_void_p), ("step_env", C.c_void_p), ("obs", C.c_void_p), ("capacity", C.c_uint32)]


class SgCounters(C.Structure):
    _fields_ = [("env_steps", C.c_uint64), ("episodes_finished", C.c_uint64), ("truncations", C.c_uint64), ("goal_hits", C.c_uint64)]


class NativeError(RuntimeError):
    pass


# every symbol include/spacegym.h declares: (restype, argtypes)
_fp, _u8p, _i32p, _vp = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.c_void_p
SYMBOLS = {
    "sg_create": (C.c_int, [C.POINTER(SgConfig), C.c_int, C.POINTER(_vp)]),
    "sg_create_ex": (C.c_int, [C.POINTER(SgConfig), C.POINTER(SgParams), C.c_int, C.POINTER(_vp)]),
    "sg_params_init": (None, [C.POINTER(SgParams)]),
    "sg_get_params": (C.c_int, [_vp, C.POINTER(SgParams)]),
    "sg_create_sharded": (C.c_int, [C.POINTER(SgConfig), C.c_int, C.POINTER(C.c_int), C.POINTER(_vp)]),
    "sg_create_sharded_ex": (C.c_int, [C.POINTER(SgConfig), C.POINTER(SgParams), C.c_int, C.POINTER(C.c_int), C.POINTER(_vp)]),
    "sg_destroy": (C.c_int, [_vp]),
    "sg_last_error": (C.c_char_p, [_vp]),
    "sg_num_envs": (C.c_int64, [_vp]),
    "sg_obs_dim": (C.c_int32, [_vp]),
    "sg_num_planets": (C.c_int32, [_vp]),
    "sg_discrete_actions": (C.c_int32, [_vp]),
    "sg_seed": (C.c_int, [_vp, C.c_uint64]),
    "sg_set_auto_reset": (C.c_int, [_vp, C.c_int32]),
    "sg_reset": (C.c_int, [_vp, _vp]),
    "sg_reset_device": (C.c_int, [_vp, _vp, _vp]),
    "sg_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sg_step_device": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sg_step_begin": (C.c_int, [_vp, _vp, C.c_int32]),
    "sg_step_end": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "sg_rollout_device": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sg_set_unfused_rollout": (C.c_int, [_vp, C.c_int32]),
    "sg_rollout_device_terminal": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, C.POINTER(SgTerminalList), _vp]),
    "sg_check_status": (C.c_int, [_vp]),
    "sg_set_counters": (C.c_int, [_vp, C.c_int32]),
    "sg_get_counters": (C.c_int, [_vp, C.POINTER(SgCounters), C.c_int32]),
    "sg_state_bytes": (C.c_size_t, [_vp]),
    "sg_save_state": (C.c_int, [_vp, _vp, C.c_size_t]),
    "sg_load_state": (C.c_int, [_vp, _vp, C.c_size_t]),
    "sg_get_state": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "sg_set_state": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "sg_vector_field": (C.c_int, [_vp, _vp, _vp, _vp]),
    "sg_host_alloc": (_vp, [C.c_size_t]),
    "sg_host_free": (None, [_vp]),
    "sg_set_profiling": (C.c_int, [_vp, C.c_int32]),
    "sg_get_profile": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "sg_stream": (_vp, [_vp]),
    "sg_rollout_kernel": (C.c_char_p, [_vp, C.c_int32]),
    "sg_random_actions_device": (C.c_int, [_vp, C.c_int32, C.c_uint64, C.c_uint64, _vp, _vp]),
    "sg_version": (C.c_char_p, []),
}


def load():
    """Load the HIP library once.  torch (if installed) is imported first so that both use the same HIP runtime
    (torch bundles libamdhip64 under the same SONAME) and device pointers can be shared."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(f"{LIB_PATH} not found: build it with `python -m space_gym_amd.build` "
                          "(hipcc, gfx950). The engine is HIP-only; there is no CPU fallback.")
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        except AttributeError:
            if not os.environ.get("SPACEGYM_LIB"):
                raise
            continue  # an older diagnostic build of the same library (A/B measurements): newer entry points are absent
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(lib, handle, rc, what):
    if rc != 0:
        msg = lib.sg_last_error(handle)
        raise NativeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
