"""ctypes binding of oracle/spacegym_oracle.c (fp64 restatement of the reference step path).

TEST INFRASTRUCTURE ONLY -- see oracle/spacegym_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_PLANETS = 4


def lib_path():
    return os.path.join(_HERE, "_build", "libspacegym_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("spacegym_oracle.c", "spacegym_oracle.h", "Makefile")]
    out = lib_path()
    if force or not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return out


class Params(C.Structure):
    _fields_ = [
        ("family", C.c_int32), ("n_planets", C.c_int32),
        ("step_size", C.c_double), ("world_size", C.c_double), ("max_abs_vel_angle", C.c_double),
        ("planet_mass", C.c_double * MAX_PLANETS), ("planet_radius", C.c_double * MAX_PLANETS),
        ("max_engine_force", C.c_double), ("ship_mass", C.c_double),
        ("goal_radius", C.c_double), ("danger_zone", C.c_double), ("survival_reward_scale", C.c_double),
        ("goal_vel_reward_scale", C.c_double), ("safety_reward_scale", C.c_double),
        ("goal_sparse_reward", C.c_double), ("distance_fctr", C.c_double),
        ("ref_orbit_a", C.c_double), ("ref_orbit_eccentricity", C.c_double), ("ref_orbit_angle", C.c_double),
        ("numerator_C", C.c_double), ("rad_penalty_C", C.c_double), ("act_penalty_C", C.c_double),
        ("tiling_rows", C.c_int32), ("tiling_cols", C.c_int32), ("tiling_a", C.c_double),
        ("max_episode_steps", C.c_int32), ("randomize_orbit", C.c_int32), ("discrete_actions", C.c_int32),
        ("steering_acceleration", C.c_int32), ("moi", C.c_double), ("max_thruster_force", C.c_double),
    ]


class Diag(C.Structure):
    _fields_ = [("n_rk_steps", C.c_int32), ("nfev", C.c_int32), ("event_index", C.c_int32), ("t_event", C.c_double)]


class EnvState(C.Structure):
    _fields_ = [
        ("state", C.c_double * 6), ("planets_xy", C.c_double * (2 * MAX_PLANETS)), ("goal_xy", C.c_double * 2),
        ("orbit", C.c_double * 3), ("elapsed", C.c_int32), ("episode", C.c_uint32), ("goal_draws", C.c_uint32),
        ("ship_tile", C.c_int32), ("goal_tile", C.c_int32), ("n_free", C.c_int32), ("free_tiles", C.c_int32 * 64),
    ]


ENV_STATE_DTYPE = np.dtype([
    ("state", "f8", 6), ("planets_xy", "f8", (MAX_PLANETS, 2)), ("goal_xy", "f8", 2), ("orbit", "f8", 3),
    ("elapsed", "i4"), ("episode", "u4"), ("goal_draws", "u4"), ("ship_tile", "i4"), ("goal_tile", "i4"),
    ("n_free", "i4"), ("free_tiles", "i4", 64)], align=True)
assert ENV_STATE_DTYPE.itemsize == C.sizeof(EnvState), (ENV_STATE_DTYPE.itemsize, C.sizeof(EnvState))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


class Oracle:
    """fp64 CPU restatement for one registered env id."""

    # constructor keywords of the reference (goal.py:18-31, kepler.py:189-203) -> field of sgo_params
    KWARGS = dict(max_engine_force="max_engine_force", ship_moi="moi", danger_zone="danger_zone",
                  survival_reward_scale="survival_reward_scale", goal_vel_reward_scale="goal_vel_reward_scale",
                  safety_reward_scale="safety_reward_scale", goal_sparse_reward="goal_sparse_reward",
                  ref_orbit_a="ref_orbit_a", ref_orbit_eccentricity="ref_orbit_eccentricity", ref_orbit_angle="ref_orbit_angle",
                  numerator_C="numerator_C", rad_penalty_C="rad_penalty_C", act_penalty_C="act_penalty_C", step_size="step_size")

    def __init__(self, env_id, threads=1, steering_acceleration=False, **kwargs):
        """kwargs: keyword arguments of the reference's constructor on top of the id's registered ones (gym.make(id, **kwargs))"""
        self.lib = C.CDLL(build())
        self.env_id = env_id
        self.threads = int(threads)
        self.params = Params()
        self.lib.sgo_params_for_id.argtypes = [C.c_char_p, C.POINTER(Params)]
        base_id = env_id
        if "n_planets" in kwargs:  # GoalEnv(n_planets=k): the tiling, radii and masses of the k-planet ids (goal.py:83-87,43)
            n = int(kwargs.pop("n_planets"))
            base_id = f"GoalDiscrete{n}-v0" if "Discrete" in env_id else f"GoalContinuous{n}P-v0"
        if self.lib.sgo_params_for_id(base_id.encode(), C.byref(self.params)) != 0:
            raise ValueError(f"unknown env id {base_id!r}")
        if base_id != env_id:  # what the id itself sets beyond the planet count (max_engine_force = 1 of the GoalDiscrete ids)
            own = Params()
            self.lib.sgo_params_for_id(env_id.encode(), C.byref(own))
            self.params.max_engine_force = own.max_engine_force
        if "ship_steering" in kwargs:
            steering_acceleration = int(kwargs.pop("ship_steering")) == 0
        if "randomize" in kwargs:
            self.params.randomize_orbit = int(bool(kwargs.pop("randomize")))
        for k, v in kwargs.items():
            setattr(self.params, self.KWARGS[k], float(v))
        self.params.steering_acceleration = int(bool(steering_acceleration))  # Steering.acceleration (ship_steering=0)
        self.lib.sgo_obs_dim.argtypes = [C.POINTER(Params)]
        self.obs_dim = self.lib.sgo_obs_dim(C.byref(self.params))
        self.n_planets = self.params.n_planets
        self.is_goal = self.params.family == 0
        self.discrete = bool(self.params.discrete_actions)
        dp, fp, u8 = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint8)
        self.lib.sgo_env_step_batch.argtypes = [C.POINTER(Params), C.c_int64, dp, dp, dp, C.c_void_p, dp, dp, u8, u8,
                                                C.POINTER(Diag), C.c_int]
        self.lib.sgo_env_step_batch.restype = None
        es = C.POINTER(EnvState)
        self.lib.sgo_vec_reset.argtypes = [C.POINTER(Params), C.c_uint64, C.c_int64, C.c_uint32, es, dp, C.c_int]
        self.lib.sgo_vec_reset.restype = None
        self.lib.sgo_vec_step.argtypes = [C.POINTER(Params), C.c_uint64, C.c_int64, C.c_uint32, es, C.c_void_p, dp, dp, u8, u8,
                                          dp, C.c_int]
        self.lib.sgo_vec_step.restype = None
        self.lib.sgo_env_resample_goal.argtypes = [C.POINTER(Params), C.c_uint64, C.c_uint32, es]
        self.lib.sgo_env_resample_goal.restype = None
        self.lib.sgo_vector_field.argtypes = [C.POINTER(Params), dp, dp, C.c_void_p, dp]
        self.lib.sgo_vector_field.restype = None
        self.lib.sgo_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        self.lib.sgo_philox4x32_10.restype = None

    def _actions(self, a, m):
        """float32 [m, 2] raw actions, or int32 [m] indices for the discrete ids"""
        if self.discrete:
            a = np.ascontiguousarray(a, np.int32)
            assert a.shape == (m,)
        else:
            a = np.ascontiguousarray(a, np.float32)
            assert a.shape == (m, 2)
        return a

    # ---- single transitions on injected inputs (spaceship_env.py:68-78), no reset
    def step(self, state, action, planets=None, goal=None, orbit=None, with_diag=False):
        """state [M,6] f64, action [M,2] raw f32, planets [M,N,2], goal [M,2] (Goal) / orbit [M,3] (Kepler, optional).
        Returns dict(state1, obs, reward, done, goal_hit[, diag])."""
        state = np.ascontiguousarray(state, np.float64).copy()
        m = state.shape[0]
        action = self._actions(action, m)
        if self.is_goal:
            planets = np.ascontiguousarray(planets, np.float64).reshape(m, self.n_planets * 2)
            aux = np.ascontiguousarray(goal, np.float64).reshape(m, 2)
        else:
            planets = None
            aux = None if orbit is None else np.ascontiguousarray(orbit, np.float64).reshape(m, 3)
        obs = np.empty((m, self.obs_dim)); reward = np.empty(m)
        done = np.empty(m, np.uint8); hit = np.empty(m, np.uint8)
        diag = (Diag * m)() if with_diag else None
        self.lib.sgo_env_step_batch(C.byref(self.params), m, _p(planets, C.c_double), _p(aux, C.c_double),
                                    _p(state, C.c_double), action.ctypes.data_as(C.c_void_p), _p(obs, C.c_double),
                                    _p(reward, C.c_double), _p(done, C.c_uint8), _p(hit, C.c_uint8), diag, self.threads)
        out = dict(state1=state, obs=obs, reward=reward, done=done, goal_hit=hit)
        if with_diag:
            out["diag"] = np.array([(d.n_rk_steps, d.nfev, d.event_index, d.t_event) for d in diag],
                                   dtype=[("n_rk_steps", "i4"), ("nfev", "i4"), ("event_index", "i4"), ("t_event", "f8")])
        return out

    def vector_field(self, state, action, planets=None):
        """SpaceshipEnv.vector_field (spaceship_env.py:96-100) for m states: [m, 6]"""
        state = np.ascontiguousarray(state, np.float64); m = len(state)
        action = self._actions(action, m)
        planets = np.ascontiguousarray(planets, np.float64).reshape(m, -1) if planets is not None else None
        out = np.empty((m, 6))
        stride = action.strides[0]
        for i in range(m):
            self.lib.sgo_vector_field(C.byref(self.params), _p(planets[i], C.c_double) if planets is not None else None,
                                      _p(state[i], C.c_double), C.c_void_p(action.ctypes.data + i * stride), _p(out[i], C.c_double))
        return out

    # ---- vector env with TimeLimit + auto-reset (engine semantics)
    def vec_reset(self, num_envs, seed=0, env_id0=0):
        envs = np.zeros(num_envs, ENV_STATE_DTYPE)
        obs = np.empty((num_envs, self.obs_dim))
        self.lib.sgo_vec_reset(C.byref(self.params), seed, num_envs, env_id0, _p(envs, EnvState), _p(obs, C.c_double),
                               self.threads)
        return envs, obs

    def vec_step(self, envs, actions, seed=0, env_id0=0, want_terminal_obs=False):
        b = len(envs)
        actions = self._actions(actions, b)
        obs = np.empty((b, self.obs_dim)); reward = np.empty(b)
        done = np.empty(b, np.uint8); trunc = np.empty(b, np.uint8)
        tobs = np.full((b, self.obs_dim), np.nan) if want_terminal_obs else None
        self.lib.sgo_vec_step(C.byref(self.params), seed, b, env_id0, _p(envs, EnvState), actions.ctypes.data_as(C.c_void_p),
                              _p(obs, C.c_double), _p(reward, C.c_double), _p(done, C.c_uint8), _p(trunc, C.c_uint8),
                              _p(tobs, C.c_double), self.threads)
        return (obs, reward, done, trunc, tobs) if want_terminal_obs else (obs, reward, done, trunc)

    def resample_goal(self, envs, i, seed=0, env_id0=0):
        ptr = C.cast(envs.ctypes.data + i * ENV_STATE_DTYPE.itemsize, C.POINTER(EnvState))
        self.lib.sgo_env_resample_goal(C.byref(self.params), seed, env_id0 + i, ptr)

    def philox(self, key, ctr):
        k = (C.c_uint32 * 2)(*key); c = (C.c_uint32 * 4)(*ctr); o = (C.c_uint32 * 4)()
        self.lib.sgo_philox4x32_10(k, c, o)
        return list(o)
