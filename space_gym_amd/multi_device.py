"""Single-process multi-GPU front end: one VectorEnv over several GPUs of one node, driven by one Python process.

Envs never interact (gym_space/dynamic_model.py:145-165 sums only an env's own planets), so the batch is cut into contiguous
blocks, one native handle per device (sg_create_sharded_ex: the RNG is keyed by the global env index, so the blocks together
are the same envs as one handle of the whole batch).  A step enqueues, per device and on that device's own stream: the copy of
the block's actions from the root device, the step kernel, and the copy of the block's (obs, reward, done, truncated) into the
root device's [num_envs, ...] arrays (device-to-device copies: xGMI peer transfers between GPUs; every peer has its own link to
the root, nothing is concatenated).  The block on the root device itself writes straight into its slice of the root arrays.
`rollout_torch(actions[K])` does the same with ONE transfer each way per K steps (the K-step rollout kernel on every device).
The one-process-per-GPU front end with torch.distributed / RCCL is space_gym_amd/sharded.py.
"""
import ctypes as C

import numpy as np

from . import _native
from .sharded import shard_bounds
from .vector_env import SpaceGymVectorEnv, _ENGINE_KWARGS


class MultiDeviceVectorEnv:
    def __init__(self, env_id, num_envs, devices, seed=0, env_index_base=0, copy=True, **kwargs):
        """devices: GPU indices, one block of envs each (the first is the root: where actions are taken from and results
        land); an index may repeat (several blocks on one GPU: how the tests run it on a one-GPU box).  kwargs: make_vec's."""
        import torch
        self._torch = torch
        self.devices = [int(d) for d in devices]
        if not self.devices:
            raise ValueError("devices must name at least one GPU")
        self.env_id, self.num_envs, self.copy = env_id, int(num_envs), bool(copy)
        n = len(self.devices)
        self.bounds = [shard_bounds(self.num_envs, n, k) for k in range(n)]
        engine = {k: kwargs.pop(k) for k in list(kwargs) if k in _ENGINE_KWARGS}
        if kwargs:
            engine["env_kwargs"] = {**(engine.get("env_kwargs") or {}), **kwargs}
        engine.pop("device", None)
        engine.setdefault("terminal_observation", False)
        # a template block gives the native config and parameter block (and validates the keywords); sg_create_sharded_ex
        # then makes the real handles: contiguous blocks, env_index_base of block k = env_index_base + its first env
        lib = _native.load()
        probe = SpaceGymVectorEnv.__new__(SpaceGymVectorEnv)
        probe._lib = lib
        tmpl = SpaceGymVectorEnv(env_id, self.bounds[0][1], device=self.devices[0], seed=seed, env_index_base=env_index_base, **engine)
        cfg, params = tmpl._cfg, tmpl._params
        tmpl.close()
        cfg.num_envs = self.num_envs
        handles = (C.c_void_p * n)()
        devs = (C.c_int * n)(*self.devices)
        rc = lib.sg_create_sharded_ex(C.byref(cfg), C.byref(params), n, devs, handles)
        _native.check(lib, None, rc, "sg_create_sharded_ex")
        self.shards = [SpaceGymVectorEnv(env_id, hi - lo, device=d, seed=seed, env_index_base=env_index_base + lo,
                                         _handle=C.c_void_p(handles[k]), **engine)
                       for k, (d, (lo, hi)) in enumerate(zip(self.devices, self.bounds))]
        s0 = self.shards[0]
        self.obs_dim, self.discrete, self.n_planets, self.spec = s0.obs_dim, s0.discrete, s0.n_planets, s0.spec
        self.single_observation_space, self.single_action_space = s0.single_observation_space, s0.single_action_space
        self.root = torch.device("cuda", self.devices[0])
        self.streams = [torch.cuda.Stream(device=d) for d in self.devices]
        self._sets, self._cur, self._roll = [self._alloc(()) for _ in range(2)], 0, {}
        a_tail = () if self.discrete else (2,)
        self._a_dtype = torch.int32 if self.discrete else torch.float32
        self._a_tail = a_tail
        # per-block staging on the block's own device (blocks on the root device need none: they use the root arrays' slices)
        self._stage = [None if d == self.devices[0] else self._block_buffers((), hi - lo, d) for d, (lo, hi) in zip(self.devices, self.bounds)]
        self._a_stage = [None if d == self.devices[0] else torch.empty((hi - lo,) + a_tail, dtype=self._a_dtype, device=torch.device("cuda", d))
                         for d, (lo, hi) in zip(self.devices, self.bounds)]

    # ------------------------------------------------------------------ buffers
    def _alloc(self, lead):
        torch, B, D = self._torch, self.num_envs, self.obs_dim
        return [torch.empty(lead + (B, D), dtype=torch.float32, device=self.root), torch.empty(lead + (B,), dtype=torch.float32, device=self.root),
                torch.empty(lead + (B,), dtype=torch.uint8, device=self.root), torch.empty(lead + (B,), dtype=torch.uint8, device=self.root)]

    def _block_buffers(self, lead, n, d):
        torch, D, dev = self._torch, self.obs_dim, self._torch.device("cuda", d)
        return [torch.empty(lead + (n, D), dtype=torch.float32, device=dev), torch.empty(lead + (n,), dtype=torch.float32, device=dev),
                torch.empty(lead + (n,), dtype=torch.uint8, device=dev), torch.empty(lead + (n,), dtype=torch.uint8, device=dev)]

    def _hand_out(self, fields):
        return tuple(x.clone() for x in fields) if self.copy else tuple(fields)

    def _fan_out(self, work):
        """run work(k, shard, stream) for every block on the block's device and stream, ordered after the root's current
        stream; the root's current stream then waits for all of them"""
        torch = self._torch
        cur = torch.cuda.current_stream(self.root)
        ready = torch.cuda.Event()
        ready.record(cur)
        events = []
        for k, (sh, st) in enumerate(zip(self.shards, self.streams)):
            with torch.cuda.device(sh.device), torch.cuda.stream(st):
                st.wait_event(ready)
                work(k, sh)
                ev = torch.cuda.Event()
                ev.record(st)
                events.append(ev)
        for ev in events:
            cur.wait_event(ev)

    # ------------------------------------------------------------------ device-tensor path
    def reset_torch(self):
        """first observations of all envs: float32 [num_envs, obs_dim] on the root device"""
        self._cur ^= 1
        obs = self._sets[self._cur][0]

        def work(k, sh):
            lo, hi = self.bounds[k]
            if self._stage[k] is None:
                sh.reset_torch(out=obs[lo:hi])
            else:
                sh.reset_torch(out=self._stage[k][0])
                obs[lo:hi].copy_(self._stage[k][0], non_blocking=True)
        self._fan_out(work)
        return obs.clone() if self.copy else obs

    def step_torch(self, actions):
        """actions: float32 [num_envs, 2] (discrete ids: int32 [num_envs]) on the root device -> (obs, reward, done, truncated)
        for all envs on the root device.  copy=False: the front end's own arrays, two sets that alternate (what a call returns
        stays valid during the next call)."""
        torch = self._torch
        if tuple(actions.shape) != (self.num_envs,) + self._a_tail or actions.dtype != self._a_dtype or actions.device != self.root:
            raise ValueError(f"actions: expected {self._a_dtype} {(self.num_envs,) + self._a_tail} on {self.root}")
        self._cur ^= 1
        fields = self._sets[self._cur]

        def work(k, sh):
            lo, hi = self.bounds[k]
            if self._stage[k] is None:
                out = dict(zip(("obs", "reward", "done", "trunc"), [f[lo:hi] for f in fields]))
                sh.step_torch(actions[lo:hi], out=out)
            else:
                self._a_stage[k].copy_(actions[lo:hi], non_blocking=True)
                st = self._stage[k]
                sh.step_torch(self._a_stage[k], out=dict(zip(("obs", "reward", "done", "trunc"), st)))
                for f, x in zip(fields, st):
                    f[lo:hi].copy_(x, non_blocking=True)
        self._fan_out(work)
        return self._hand_out(fields)

    def rollout_torch(self, actions):
        """K steps, one transfer each way per device: actions [K, num_envs, 2] (discrete ids int32 [K, num_envs]) on the root
        device -> (obs [K, num_envs, obs_dim], reward, done, truncated [K, num_envs]) on the root device"""
        torch = self._torch
        K = int(actions.shape[0])
        if tuple(actions.shape) != (K, self.num_envs) + self._a_tail or actions.dtype != self._a_dtype or actions.device != self.root:
            raise ValueError(f"actions: expected {self._a_dtype} {(K, self.num_envs) + self._a_tail} on {self.root}")
        if K not in self._roll:
            self._roll[K] = dict(sets=[self._alloc((K,)) for _ in range(2)], cur=0,
                                 stage=[self._block_buffers((K,), hi - lo, d) for d, (lo, hi) in zip(self.devices, self.bounds)],
                                 a=[torch.empty((K, hi - lo) + self._a_tail, dtype=self._a_dtype, device=torch.device("cuda", d))
                                    for d, (lo, hi) in zip(self.devices, self.bounds)])
        r = self._roll[K]
        r["cur"] ^= 1
        fields = r["sets"][r["cur"]]

        def work(k, sh):  # (a block's [K, n, ...] results are contiguous only in its own buffers: every block is staged)
            lo, hi = self.bounds[k]
            r["a"][k].copy_(actions[:, lo:hi], non_blocking=True)
            st = r["stage"][k]
            sh.rollout_torch(r["a"][k], *st)
            for f, x in zip(fields, st):
                f[:, lo:hi].copy_(x, non_blocking=True)
        self._fan_out(work)
        return self._hand_out(fields)

    # ------------------------------------------------------------------ NumPy convenience, lifecycle
    def reset(self):
        return self.reset_torch().cpu().numpy()

    def step(self, actions):
        """NumPy in / out like SpaceGymVectorEnv.step: (obs, reward, done, info)"""
        a = self._torch.as_tensor(np.ascontiguousarray(actions, dtype=np.int32 if self.discrete else np.float32)).to(self.root)
        obs, rew, done, trunc = self.step_torch(a)
        return obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy().astype(bool), {"TimeLimit.truncated": trunc.cpu().numpy().astype(bool)}

    def seed(self, seed=None):
        out = None
        for sh in self.shards:
            out = sh.seed(seed if seed is not None else (out[0] if out else None))
        return out

    def get_state(self):
        """the blocks' states, concatenated in env order (host arrays)"""
        parts = [sh.get_state() for sh in self.shards]
        return {k: (None if parts[0][k] is None else np.concatenate([p[k] for p in parts])) for k in parts[0]}

    def check_status(self):
        for sh in self.shards:
            sh.check_status()

    def close(self):
        for sh in getattr(self, "shards", []):
            sh.close()
        self.shards = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
