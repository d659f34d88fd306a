"""gym.vector.VectorEnv-shaped front end of the HIP engine.

Mirrors the reference's env surface for the step path -- `reset() -> obs`, `step(a) -> (obs, reward, done, info)`
(old-gym 4-tuple, gym_space/envs/spaceship_env.py:59-78), `seed(s)` (:92-94), observation/action spaces
(:102-111,206-208; kepler.py:158-170) -- batched over `num_envs` instances, with what gym.wrappers.TimeLimit
(max_episode_steps=500, gym_space/__init__.py:29) and a VectorEnv add: per-env step counters, truncation and
auto-reset, all inside the step kernel.

Two I/O modes:
  * NumPy (`reset`, `step` = `step_async` + `step_wait`): host arrays in/out (H2D + kernel + D2H per step).  step_async
    enqueues all of it (sg_step_begin) and returns; step_wait waits (sg_step_end) and hands out the step's results, which
    sit in one of two page-locked blocks that alternate -- with copy=False the arrays of step t stay valid while step t + 1
    is in flight;
  * torch (`reset_torch`, `step_torch`, `rollout_torch`): device tensors in/out through sg_step_device on torch's
    current stream, zero-copy -- the high-throughput path.
"""
import ctypes as C

import numpy as np

from . import _native
from .registration import (ENV_CLASSES, ENV_SPECS, constructor_kwargs, is_discrete, obs_dim, single_action_space,
                           single_observation_space)
from .spaces import MultiDiscrete, batch_box


class StepInfo(dict):
    """Batched info: {"TimeLimit.truncated": bool[B], "terminal_observation": float32[B, D] (rows of finished envs)}.
    A per-env list of dicts (old gym VectorEnv) would cost more than the step itself at B = 65536."""


class SpaceGymVectorEnv:
    metadata = {"render.modes": []}

    def __init__(self, env_id, num_envs, device=0, seed=0, env_index_base=0, max_episode_steps=None, auto_reset=True,
                 validate_actions=True, terminal_observation=True, copy=True, steering=None, env_kwargs=None, from_class=False,
                 _handle=None):
        """steering: "velocity" (ship_steering=1, what every registered id uses) or "acceleration" (ship_steering=0, the
        constructor default of the reference classes: omega is a state, the thruster a torque); None: what env_kwargs say.
        env_kwargs: keyword arguments of the reference's constructor (GoalEnv.__init__ goal.py:18-31, KeplerEnv.__init__
        kepler.py:189-203) on top of the ones the id is registered with -- what gym.make(env_id, **env_kwargs) does; make_vec
        passes its unknown keywords here.  from_class: the id only names the family and action space, every keyword comes from
        the class defaults and env_kwargs (make_vec_from_class).
        validate_actions: step() checks on the host that the actions are in range, as the reference's step asserts
        (spaceship_env.py:71; discrete ids: ValueError, :201-202); off, out-of-range actions are clamped on the device (the
        device-tensor calls never validate: that would need a device-to-host synchronisation).
        copy=False: reset()/step() return views of the engine's pinned output buffers, overwritten by the next call
        (no per-step allocation or copy); copy=True returns fresh arrays like gym's vector envs."""
        if env_id not in ENV_SPECS:
            raise ValueError(f"unknown env id {env_id!r}; served ids: {sorted(ENV_SPECS)}")
        self._lib = _native.load()
        self.env_id, self.num_envs, self.device = env_id, int(num_envs), int(device)
        self.spec = dict(ENV_SPECS[env_id])
        # the constructor's keyword arguments as the reference would see them, and the native parameter block they fill
        self.env_kwargs = constructor_kwargs(env_id, env_kwargs, from_class=from_class)
        if steering is None:
            if self.env_kwargs["ship_steering"] not in (0, 1):  # Steering.angle: no thruster does anything (dynamic_model.py:138-141,160-163)
                raise ValueError("ship_steering must be 0 (Steering.acceleration) or 1 (Steering.velocity)")
            steering = "velocity" if self.env_kwargs["ship_steering"] == 1 else "acceleration"
        self.env_kwargs["ship_steering"] = {"velocity": 1, "acceleration": 0}[steering]
        params = self._native_params(self.env_kwargs)
        if self.spec["family"] == "goal":
            self.spec["n_planets"] = int(self.env_kwargs["n_planets"])
        self.obs_dim = obs_dim(env_id, self.spec["n_planets"])
        self.n_planets = self.spec["n_planets"]
        self.single_observation_space = single_observation_space(env_id, self.n_planets)
        self.single_action_space = single_action_space(env_id)
        self.observation_space = batch_box(self.single_observation_space, self.num_envs)
        self.discrete = is_discrete(env_id)
        self.action_space = (MultiDiscrete([self.single_action_space.n] * self.num_envs) if self.discrete
                             else batch_box(self.single_action_space, self.num_envs))
        self.validate_actions = validate_actions
        self.want_terminal_obs = terminal_observation
        cfg = _native.SgConfig(env_id=env_id.encode(), num_envs=self.num_envs, seed=int(seed),
                               env_index_base=int(env_index_base), max_episode_steps=int(max_episode_steps or 0),
                               auto_reset=int(bool(auto_reset)), steering={"velocity": 0, "acceleration": 1}[steering])
        self._cfg, self._params = cfg, params
        if _handle is None:
            h = C.c_void_p()
            rc = self._lib.sg_create_ex(C.byref(cfg), C.byref(params), self.device, C.byref(h))
            _native.check(self._lib, None, rc, "sg_create_ex")
        else:  # a handle made by sg_create_sharded_ex (MultiDeviceVectorEnv): `num_envs`, `device`, `env_index_base` describe it
            h = _handle
        self._h = h
        assert self._lib.sg_obs_dim(h) == self.obs_dim
        B, D = self.num_envs, self.obs_dim
        self.copy = bool(copy)
        self._pinned = []
        self._obs = self._host_array((B, D), np.float32)  # reset()'s observations (a step's outputs sit in the handle's blocks)
        self._act = self._host_array((B,) if self.discrete else (B, 2), np.int32 if self.discrete else np.float32)
        self._pending = False
        self._blocks = {}
        self._torch_bufs = None

    def _native_params(self, kw):
        """sg_params (include/spacegym.h) from the constructor's keyword arguments"""
        p = _native.SgParams()
        self._lib.sg_params_init(C.byref(p))
        names = (("goal_vel_reward_scale", "safety_reward_scale", "goal_sparse_reward", "survival_reward_scale", "danger_zone")
                 if self.spec["family"] == "goal" else
                 ("ref_orbit_a", "ref_orbit_eccentricity", "ref_orbit_angle", "numerator_C", "rad_penalty_C", "act_penalty_C", "step_size"))
        for k in names + ("ship_moi", "max_engine_force"):
            setattr(p, k, float(kw[k]))
        if self.spec["family"] == "goal":
            p.n_planets = int(kw["n_planets"])
        else:
            p.randomize = int(bool(kw["randomize"]))
        return p

    def native_params(self):
        """the parameters the handle was built with (sg_get_params), as a dict of the reference's keyword names"""
        p = _native.SgParams()
        self._ck(self._lib.sg_get_params(self._h, C.byref(p)), "sg_get_params")
        out = {k: getattr(p, k) for k, _ in p._fields_ if k not in ("struct_size", "reserved")}
        return {k: v for k, v in out.items() if not (isinstance(v, float) and np.isnan(v)) and v != -1}

    def _host_array(self, shape, dtype):
        """NumPy array over page-locked memory (sg_host_alloc); ordinary memory if pinning fails."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = self._lib.sg_host_alloc(n)
        if not ptr:
            return np.empty(shape, dtype)
        self._pinned.append(ptr)
        return np.frombuffer((C.c_char * n).from_address(ptr), dtype=dtype).reshape(shape)

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "_h", None):
            self._lib.sg_destroy(self._h)
            self._h = None
            self._obs = self._act = None
            self._blocks = {}
            for ptr in self._pinned:
                self._lib.sg_host_free(ptr)
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        _native.check(self._lib, self._h, rc, what)

    def seed(self, seed=None):
        """SpaceshipEnv.seed (spaceship_env.py:92-94): returns [seed]; applies from the next reset()."""
        seed = int(np.random.SeedSequence().entropy % (1 << 63)) if seed is None else int(seed)
        self._ck(self._lib.sg_seed(self._h, seed), "sg_seed")
        return [seed]

    def set_counters(self, on=True):
        """per-batch event counters (sg_set_counters): env-steps, finished episodes, truncations, goals reached; off by default"""
        self._ck(self._lib.sg_set_counters(self._h, int(bool(on))), "sg_set_counters")

    def counters(self, reset=False):
        k = _native.SgCounters()
        self._ck(self._lib.sg_get_counters(self._h, C.byref(k), int(bool(reset))), "sg_get_counters")
        return {f: int(getattr(k, f)) for f, _ in k._fields_}

    def set_auto_reset(self, on):
        self._ck(self._lib.sg_set_auto_reset(self._h, int(bool(on))), "sg_set_auto_reset")

    # ------------------------------------------------------------------ NumPy path
    @staticmethod
    def _ptr(a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else None

    def reset(self):
        self._ck(self._lib.sg_reset(self._h, self._ptr(self._obs)), "sg_reset")
        return self._obs.copy() if self.copy else self._obs

    def _check_actions(self, actions):
        if self.discrete:  # int index per env, spaceship_env.py:189-202
            actions = np.ascontiguousarray(actions, dtype=np.int32)
            if actions.shape != (self.num_envs,):
                raise ValueError(f"actions must have shape ({self.num_envs},), got {actions.shape}")
            if self.validate_actions and not (np.all(actions >= 0) and np.all(actions <= 5)):
                raise ValueError("discrete action out of range")  # the reference raises ValueError, spaceship_env.py:201-202
            return actions
        actions = np.ascontiguousarray(actions, dtype=np.float32)  # raw_action.astype(np.float32), spaceship_env.py:69-70
        if actions.shape != (self.num_envs, 2):
            raise ValueError(f"actions must have shape ({self.num_envs}, 2), got {actions.shape}")
        if self.validate_actions:  # assert self.action_space.contains(raw_action), spaceship_env.py:71
            assert np.all(actions >= -1.0) and np.all(actions <= 1.0), actions
        return actions

    def step_async(self, actions):
        """gym.vector's step_async: the step kernel is enqueued on the engine's stream -- it reads the actions from the pinned
        action buffer and stores its outputs into one of the handle's two page-locked result blocks itself -- and the call
        returns; step_wait() collects."""
        if self._pending:
            raise RuntimeError("step_async() while a step is in flight (step_wait() first)")
        np.copyto(self._act, self._check_actions(actions))  # into the pinned action buffer (unchanged until step_wait)
        self._ck(self._lib.sg_step_begin(self._h, self._ptr(self._act), int(self.want_terminal_obs)), "sg_step_begin")
        self._pending = True

    def _block_views(self, ptrs):
        """NumPy views of one of the handle's two result blocks (made once per block)"""
        key = ptrs[0]
        if key not in self._blocks:
            B, D = self.num_envs, self.obs_dim

            def view(ptr, shape, dtype):
                n = int(np.prod(shape)) * np.dtype(dtype).itemsize
                return np.frombuffer((C.c_char * n).from_address(ptr), dtype=dtype).reshape(shape)
            self._blocks[key] = (view(ptrs[0], (B, D), np.float32), view(ptrs[1], (B,), np.float32), view(ptrs[2], (B,), np.uint8),
                                 view(ptrs[3], (B,), np.uint8), view(ptrs[4], (B, D), np.float32) if ptrs[4] else None)
        return self._blocks[key]

    def step_wait(self):
        if not self._pending:
            raise RuntimeError("step_wait() without step_async()")
        p = [C.c_void_p() for _ in range(5)]
        rc = self._lib.sg_step_end(self._h, *[C.byref(x) for x in p])
        self._pending = False  # (the native side has given the step up as well if the wait failed)
        self._ck(rc, "sg_step_end")
        obs, rew, done, trunc, tobs = self._block_views([x.value for x in p])
        info = StepInfo({"TimeLimit.truncated": trunc.view(np.bool_) if not self.copy else trunc.astype(bool)})
        if tobs is not None:
            info["terminal_observation"] = tobs.copy() if self.copy else tobs
        if self.copy:
            return obs.copy(), rew.copy(), done.astype(bool), info
        return obs, rew, done.view(np.bool_), info

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    # ------------------------------------------------------------------ state access (golden-vector injection, checkpoints)
    def get_state(self):
        B, N = self.num_envs, self.n_planets
        ship = np.empty((B, 6), np.float32)
        planets = np.empty((B, N, 2), np.float32) if N else None
        goal = np.empty((B, 2), np.float32)
        elapsed = np.empty(B, np.int32)
        self._ck(self._lib.sg_get_state(self._h, self._ptr(ship), self._ptr(planets), self._ptr(goal), self._ptr(elapsed)),
                 "sg_get_state")
        return dict(ship=ship, planets=planets, goal=goal, elapsed=elapsed)

    def set_state(self, ship=None, planets=None, goal=None, elapsed=None):
        B, N = self.num_envs, self.n_planets

        def prep(a, shape, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dt)
            if a.shape != shape:
                raise ValueError(f"expected shape {shape}, got {a.shape}")
            return a
        ship, goal = prep(ship, (B, 6), np.float32), prep(goal, (B, 2), np.float32)
        planets, elapsed = prep(planets, (B, N, 2), np.float32), prep(elapsed, (B,), np.int32)
        self._ck(self._lib.sg_set_state(self._h, self._ptr(ship), self._ptr(planets), self._ptr(goal), self._ptr(elapsed)),
                 "sg_set_state")

    def vector_field(self, actions, ship=None):
        """SpaceshipEnv.vector_field (spaceship_env.py:96-100) for every env: float32 [B, 6] = (vx, vy, omega, ax, ay, alpha),
        at the current state or at the given `ship` states [B, 6] (planets as they are now)."""
        a = self._check_actions(actions)
        ship = None if ship is None else np.ascontiguousarray(ship, np.float32)
        out = np.empty((self.num_envs, 6), np.float32)
        self._ck(self._lib.sg_vector_field(self._h, self._ptr(a), self._ptr(ship), self._ptr(out)), "sg_vector_field")
        return out

    # ------------------------------------------------------------------ torch path (device tensors, current stream)
    def _torch(self):
        import torch
        if self._torch_bufs is None:
            dev = torch.device("cuda", self.device)
            B, D = self.num_envs, self.obs_dim
            self._torch_bufs = dict(
                obs=torch.empty((B, D), dtype=torch.float32, device=dev), reward=torch.empty(B, dtype=torch.float32, device=dev),
                done=torch.empty(B, dtype=torch.uint8, device=dev), trunc=torch.empty(B, dtype=torch.uint8, device=dev))
        return torch, self._torch_bufs

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset_torch(self, out=None):
        torch, bufs = self._torch()
        obs = bufs["obs"] if out is None else out
        if out is not None:
            self._check_tensor("out", out, torch.float32, (self.num_envs, self.obs_dim))
        self._ck(self._lib.sg_reset_device(self._h, C.c_void_p(obs.data_ptr()), self._stream()), "sg_reset_device")
        return obs

    def step_torch(self, actions, out=None, terminal_obs=None):
        """actions: float32 CUDA tensor [B, 2] (discrete ids: int32 [B]), or any device array exporting DLPack
        (`__dlpack__`: CuPy, JAX, ...; taken zero-copy).  Returns (obs, reward, done, truncated) device tensors, which are
        reused by the next call unless `out` (a dict with the same keys) is given; `torch.utils.dlpack.to_dlpack` /
        `__dlpack__` hands them on to other frameworks without a copy."""
        torch, bufs = self._torch()
        o = bufs if out is None else out
        if not isinstance(actions, torch.Tensor) and hasattr(actions, "__dlpack__"):
            actions = torch.from_dlpack(actions)
        B, D = self.num_envs, self.obs_dim
        self._check_tensor("actions", actions, torch.int32 if self.discrete else torch.float32, (B,) if self.discrete else (B, 2))
        if out is not None:
            self._check_tensor("out['obs']", o["obs"], torch.float32, (B, D))
            self._check_tensor("out['reward']", o["reward"], torch.float32, (B,))
            self._check_tensor("out['done']", o["done"], torch.uint8, (B,))
            self._check_tensor("out['trunc']", o["trunc"], torch.uint8, (B,))
        if terminal_obs is not None:
            self._check_tensor("terminal_obs", terminal_obs, torch.float32, (B, D))
        rc = self._lib.sg_step_device(self._h, C.c_void_p(actions.data_ptr()), C.c_void_p(o["obs"].data_ptr()),
                                      C.c_void_p(o["reward"].data_ptr()), C.c_void_p(o["done"].data_ptr()),
                                      C.c_void_p(o["trunc"].data_ptr()),
                                      C.c_void_p(terminal_obs.data_ptr()) if terminal_obs is not None else None, self._stream())
        self._ck(rc, "sg_step_device")
        return o["obs"], o["reward"], o["done"], o["trunc"]

    def _check_tensor(self, name, t, dtype, shape):
        """a raw pointer goes to the kernel: refuse anything whose memory is not what the kernel will write / read"""
        import torch
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.device.index == self.device):
            raise ValueError(f"{name}: expected a CUDA tensor on device {self.device}")
        if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
            raise ValueError(f"{name}: expected contiguous {dtype} of shape {tuple(shape)}, got {t.dtype} {tuple(t.shape)}"
                             f"{'' if t.is_contiguous() else ' (not contiguous)'}")

    def rollout_torch(self, actions, obs, reward, done, trunc, terminal=None):
        """actions [K, B, 2] (discrete ids: int32 [K, B]) -> obs [K, B, D], reward/done/trunc [K, B]: K steps, one launch.
        terminal: optional dict(count=uint32/int32 [1], step_env=int32 [cap, 2], obs=float32 [cap, D]) of device tensors
        that receives one record per finished env-step: its (step, env) and the LAST observation of the episode that ended
        there (sg_rollout_device_terminal); `terminal_records` turns it into sorted host arrays."""
        import torch
        K, B, D = int(actions.shape[0]), self.num_envs, self.obs_dim
        self._check_tensor("actions", actions, torch.int32 if self.discrete else torch.float32, (K, B) if self.discrete else (K, B, 2))
        self._check_tensor("obs", obs, torch.float32, (K, B, D))
        self._check_tensor("reward", reward, torch.float32, (K, B))
        self._check_tensor("done", done, torch.uint8, (K, B))
        self._check_tensor("trunc", trunc, torch.uint8, (K, B))
        args = (self._h, K, C.c_void_p(actions.data_ptr()), C.c_void_p(obs.data_ptr()), C.c_void_p(reward.data_ptr()),
                C.c_void_p(done.data_ptr()), C.c_void_p(trunc.data_ptr()))
        if terminal is None:
            self._ck(self._lib.sg_rollout_device(*args, self._stream()), "sg_rollout_device")
        else:
            cap = int(terminal["step_env"].shape[0])
            if terminal["count"].dtype not in (torch.int32, torch.uint32) or terminal["count"].numel() != 1:
                raise ValueError("terminal['count']: expected one 32-bit integer")
            self._check_tensor("terminal['step_env']", terminal["step_env"], torch.int32, (cap, 2))
            self._check_tensor("terminal['obs']", terminal["obs"], torch.float32, (cap, D))
            tl = _native.SgTerminalList(terminal["count"].data_ptr(), terminal["step_env"].data_ptr(), terminal["obs"].data_ptr(), cap)
            self._ck(self._lib.sg_rollout_device_terminal(*args, C.byref(tl), self._stream()), "sg_rollout_device_terminal")
        return obs, reward, done, trunc

    def prepare_rollout(self, actions, obs, reward, done, trunc):
        """Validates the buffers once and returns a zero-argument callable that enqueues the rollout on torch's current
        stream: for loops that re-use the same buffers (the per-call checks of rollout_torch cost more host time than a
        short rollout takes on the GPU)."""
        import torch
        K, B, D = int(actions.shape[0]), self.num_envs, self.obs_dim
        self._check_tensor("actions", actions, torch.int32 if self.discrete else torch.float32, (K, B) if self.discrete else (K, B, 2))
        self._check_tensor("obs", obs, torch.float32, (K, B, D))
        self._check_tensor("reward", reward, torch.float32, (K, B))
        self._check_tensor("done", done, torch.uint8, (K, B))
        self._check_tensor("trunc", trunc, torch.uint8, (K, B))
        keep = (actions, obs, reward, done, trunc)  # the callable keeps the tensors alive
        args = (self._h, K) + tuple(C.c_void_p(t.data_ptr()) for t in keep)
        fn, ck, dev = self._lib.sg_rollout_device, self._ck, self.device
        cur = torch.cuda.current_stream

        def call():
            ck(fn(*args, C.c_void_p(cur(dev).cuda_stream)), "sg_rollout_device")
        call.keep = keep
        return call

    def terminal_list_torch(self, capacity):
        """device buffers for rollout_torch(..., terminal=...)"""
        import torch
        dev = torch.device("cuda", self.device)
        return dict(count=torch.zeros(1, dtype=torch.int32, device=dev), step_env=torch.empty((capacity, 2), dtype=torch.int32, device=dev),
                    obs=torch.empty((capacity, self.obs_dim), dtype=torch.float32, device=dev))

    @staticmethod
    def terminal_records(terminal):
        """(step int32 [n], env int32 [n], obs float32 [n, D]) sorted by (step, env); raises if the list overflowed"""
        n = int(terminal["count"].item())
        cap = int(terminal["step_env"].shape[0])
        if n > cap:
            raise OverflowError(f"terminal list overflow: {n} records, capacity {cap}")
        se = terminal["step_env"][:n].cpu().numpy()
        ob = terminal["obs"][:n].cpu().numpy()
        order = np.lexsort((se[:, 1], se[:, 0]))
        return se[order, 0], se[order, 1], ob[order]

    def check_status(self):
        """waits for the enqueued work; raises if a rollout kernel's bounded wave hand-off wait ran out (sg_check_status)"""
        self._ck(self._lib.sg_check_status(self._h), "sg_check_status")

    # ------------------------------------------------------------------ complete snapshot
    def save_state(self):
        """opaque uint8 blob with every per-env column (incl. the tiling state and episode counters) and the RNG key"""
        n = int(self._lib.sg_state_bytes(self._h))
        blob = np.empty(n, np.uint8)
        self._ck(self._lib.sg_save_state(self._h, self._ptr(blob), n), "sg_save_state")
        return blob

    def load_state(self, blob):
        blob = np.ascontiguousarray(blob, np.uint8)
        self._ck(self._lib.sg_load_state(self._h, self._ptr(blob), blob.size), "sg_load_state")

    SNAPSHOT_HEADER_BYTES = 48

    def snapshot_columns(self, blob):
        """introspection of a save_state() blob: the engine's per-env columns as NumPy views --
        q0 (x, y, theta, vx), q1 (vy, omega, goal_x, goal_y | orbit angle, eccentricity), ctr (elapsed, episode),
        aux (goal draws, ship_tile | goal_tile << 8 | case_b << 16 | flip << 17, free-tile multiset lo, hi),
        Goal: pl0 / pl1 (two planets each), cshift (tiling column shifts); KeplerRandomOrbits: orbd (cos, sin of the angle)"""
        B, off, out = self.num_envs, self.SNAPSHOT_HEADER_BYTES, {}
        cols = [("q0", np.float32, 4), ("q1", np.float32, 4), ("ctr", np.uint32, 2), ("aux", np.uint32, 4)]
        if self.spec["family"] == "goal":
            cols += [("pl0", np.float32, 4)] + ([("pl1", np.float32, 4)] if self.n_planets > 2 else []) + [("cshift", np.float32, 4)]
        elif self.env_id == "KeplerRandomOrbits-v0":
            cols += [("orbd", np.float64, 2)]
        for name, dt, w in cols:
            nb = B * w * np.dtype(dt).itemsize
            out[name] = np.frombuffer(blob, dtype=dt, count=B * w, offset=off).reshape(B, w)
            off += nb
        assert off == blob.size, (off, blob.size)
        return out

    def random_actions_torch(self, n_steps, seed=0, first_step=0, out=None):
        """the uniformly random policy generated on the device: [n_steps, B, 2] float32 in (-1, 1) (discrete ids: int32
        [n_steps, B] in 0..5); entry (t, i) depends only on (seed, global env index, first_step + t)."""
        import torch
        shape = (int(n_steps), self.num_envs) if self.discrete else (int(n_steps), self.num_envs, 2)
        if out is None:
            out = torch.empty(shape, dtype=torch.int32 if self.discrete else torch.float32, device=f"cuda:{self.device}")
        assert tuple(out.shape) == shape and out.is_contiguous()
        rc = self._lib.sg_random_actions_device(self._h, int(n_steps), C.c_uint64(seed), C.c_uint64(first_step),
                                                C.c_void_p(out.data_ptr()), self._stream())
        self._ck(rc, "sg_random_actions_device")
        return out

    def set_unfused_rollout(self, on):
        """rollout_torch as K launches of the step kernel instead of the fused K-step kernel (A/B, equivalence test)."""
        self._ck(self._lib.sg_set_unfused_rollout(self._h, int(on)), "sg_set_unfused_rollout")

    # ------------------------------------------------------------------ measurement aid
    def rollout_kernel(self, n_steps):
        """name of the kernel rollout_torch launches for n_steps steps (as rocprofv3 prints it)"""
        return self._lib.sg_rollout_kernel(self._h, int(n_steps)).decode()

    def set_profiling(self, on):
        self._ck(self._lib.sg_set_profiling(self._h, int(bool(on))), "sg_set_profiling")

    def get_profile(self):
        """(launches, total_ms, min_ms, max_ms) of the step-kernel launches since the last call; synchronise first."""
        n, tot, mn, mx = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
        self._ck(self._lib.sg_get_profile(self._h, C.byref(n), C.byref(tot), C.byref(mn), C.byref(mx)), "sg_get_profile")
        return n.value, tot.value, mn.value, mx.value


_ENGINE_KWARGS = ("device", "seed", "env_index_base", "max_episode_steps", "auto_reset", "validate_actions", "terminal_observation",
                  "copy", "steering", "env_kwargs", "from_class")


def make_vec(env_id, num_envs=1, **kwargs):
    """Batched counterpart of gym.make(env_id, **kwargs) for the ids in gym_space/__init__.py: keywords of the reference's
    constructors (GoalEnv.__init__ goal.py:18-31: goal_vel_reward_scale, safety_reward_scale, goal_sparse_reward, danger_zone,
    survival_reward_scale, n_planets, ship_steering, ship_moi, max_engine_force; KeplerEnv.__init__ kepler.py:189-203: randomize,
    ref_orbit_a, ref_orbit_eccentricity, ref_orbit_angle, numerator_C, rad_penalty_C, act_penalty_C, step_size, ship_steering,
    ship_moi, max_engine_force) override what the id is registered with; the engine's own keywords (device, seed, ...) are
    SpaceGymVectorEnv's.  devices=[...] cuts the batch into one block per listed GPU, driven by this one process
    (MultiDeviceVectorEnv)."""
    devices = kwargs.pop("devices", None)
    if devices is not None:  # one VectorEnv over several GPUs, driven by this process (space_gym_amd/multi_device.py)
        from .multi_device import MultiDeviceVectorEnv
        return MultiDeviceVectorEnv(env_id, num_envs, devices, **kwargs)
    engine = {k: kwargs.pop(k) for k in list(kwargs) if k in _ENGINE_KWARGS}
    if kwargs:
        engine["env_kwargs"] = {**(engine.get("env_kwargs") or {}), **kwargs}
    return SpaceGymVectorEnv(env_id, num_envs, **engine)


def make_vec_from_class(class_name, num_envs=1, **kwargs):
    """Batched counterpart of constructing one of the reference's classes directly -- GoalContinuousEnv(**kwargs),
    GoalDiscreteEnv, KeplerContinuousEnv, KeplerDiscreteEnv (goal.py:286-291, kepler.py:270-275): the constructor's own
    defaults apply (ship_steering=0, i.e. Steering.acceleration; KeplerEnv: step_size=0.1) and GoalEnv's three reward scales
    are required.  No TimeLimit unless max_episode_steps is given (the classes have none; it is gym.make that adds it)."""
    if class_name not in ENV_CLASSES:
        raise ValueError(f"unknown class {class_name!r}; served: {sorted(ENV_CLASSES)}")
    kwargs.setdefault("max_episode_steps", 2 ** 31 - 1)
    return make_vec(ENV_CLASSES[class_name], num_envs, from_class=True, **kwargs)
