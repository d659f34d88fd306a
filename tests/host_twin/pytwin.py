"""ctypes binding of the TEST-ONLY host twin (fp32 device math compiled for the CPU)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build as _build  # noqa: E402


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


class Twin:
    def __init__(self, env_id, steering_acceleration=False, defines=(), tag=""):
        self.lib = C.CDLL(_build.build(defines=defines, tag=tag))
        self.steering_acceleration = bool(steering_acceleration)
        self.env_id = env_id.encode()
        self.obs_dim = self.lib.twin_obs_dim(self.env_id)
        assert self.obs_dim > 0
        self.is_goal = env_id.startswith("Goal")
        self.discrete = "Discrete" in env_id
        self.n_planets = (self.obs_dim - 9) // 2 if self.is_goal else 0

    def step(self, state, action, planets=None, goal=None, diag=False):
        state = np.ascontiguousarray(state, np.float32); m = len(state)
        action = np.ascontiguousarray(action, np.int32 if self.discrete else np.float32)
        planets = np.ascontiguousarray(planets, np.float32) if planets is not None else None
        goal = np.ascontiguousarray(goal, np.float32) if goal is not None else None
        out = dict(state1=np.empty((m, 6), np.float32), obs=np.empty((m, self.obs_dim), np.float32),
                   reward=np.empty(m, np.float32), done=np.empty(m, np.uint8), goal_hit=np.empty(m, np.uint8),
                   t=np.empty(m, np.float32), n_rk=np.empty(m, np.int32), event=np.empty(m, np.int32))
        f, u8, i32 = C.c_float, C.c_uint8, C.c_int32
        self.lib.twin_set_steering_acceleration(int(self.steering_acceleration))
        if diag:  # how each env-step was integrated (0 probe step kept, 1 probe step terminal, 2 / 3 scipy's sequence, 4 probe off)
            out["path"], out["probe_err"], out["probe_abs"] = np.empty(m, np.int32), np.empty(m, np.float32), np.empty((m, 2), np.float32)
            self.lib.twin_set_diag(_p(out["path"], i32), _p(out["probe_err"], f), _p(out["probe_abs"], f))
        rc = self.lib.twin_step(self.env_id, C.c_int64(m), _p(state, f), _p(planets, f), _p(goal, f), action.ctypes.data_as(C.c_void_p),
                                _p(out["state1"], f), _p(out["obs"], f), _p(out["reward"], f), _p(out["done"], u8),
                                _p(out["goal_hit"], u8), _p(out["t"], f), _p(out["n_rk"], i32), _p(out["event"], i32))
        self.lib.twin_set_diag(None, None, None)
        assert rc == 0
        return out

    def reset(self, m, seed=0, env0=0, episode=0, n_hits=0):
        n = max(self.n_planets, 1)
        out = dict(state=np.empty((m, 6), np.float32), planets=np.zeros((m, n, 2), np.float32),
                   goals=np.zeros((m, n_hits + 1, 2), np.float32), tiles=np.zeros((m, n_hits + 1), np.uint32),
                   free_counts=np.zeros((m, n_hits + 1), np.uint64), col_shift=np.zeros((m, 4), np.float32),
                   orbit=np.zeros((m, 2), np.float32))
        f = C.c_float
        rc = self.lib.twin_reset(self.env_id, C.c_uint64(seed), C.c_int64(m), C.c_uint32(env0), C.c_uint32(episode),
                                 C.c_int(n_hits), _p(out["state"], f), _p(out["planets"], f), _p(out["goals"], f),
                                 _p(out["tiles"], C.c_uint32), _p(out["free_counts"], C.c_uint64),
                                 _p(out["col_shift"], f), _p(out["orbit"], f))
        assert rc == 0
        return out

    def philox(self, key, ctr):
        c = (C.c_uint32 * 4)(*ctr); o = (C.c_uint32 * 4)()
        self.lib.twin_philox(C.c_uint32(key[0]), C.c_uint32(key[1]), c, o)
        return list(o)


def sincos(which, r):
    """which: 0 sincos_small, 1 sincos_small2 (x of the sine, y of the cosine of the pair (r, -r)), 2 sincos_acc"""
    lib = C.CDLL(_build.build())
    r = np.ascontiguousarray(r, np.float32)
    s, c = np.empty_like(r), np.empty_like(r)
    lib.twin_sincos(C.c_int(which), C.c_int64(r.size), _p(r, C.c_float), _p(s, C.c_float), _p(c, C.c_float))
    return s, c
