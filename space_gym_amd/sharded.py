"""Env-sharded multi-GPU front end: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Envs never interact (gym_space/dynamic_model.py:145-165 sums only an env's own planets), so the batch is cut into
contiguous blocks -- rank k owns global envs [k*B/W, (k+1)*B/W) -- and `step_local` involves no communication at all.
The RNG is keyed by the GLOBAL env index (sg_config.env_index_base), so results do not depend on the number of ranks.

Only the single-process VectorEnv view needs an exchange: `step(actions)` scatters rank 0's actions ([B, 2] float32, or [B]
int32 indices for the discrete ids) and gathers every rank's (obs, [terminal obs], reward, done, truncated) into rank 0's
result arrays -- allocated once, [B, ...] each; the peers' blocks are received straight into their slices (one batch of
point-to-point transfers per step: on the fully connected xGMI mesh every peer uses its own link to the root; no ring, no
all-reduce, no concatenation on the root).  With copy=True (the default, like SpaceGymVectorEnv and gym's vector envs)
reset() / step() / rollout() hand out fresh copies of those arrays; with copy=False the arrays themselves, from two sets
that alternate: what a call returns stays valid during the next call and is overwritten by the one after.
`rollout(actions[K])` is K steps with ONE scatter and ONE gather (the K-step rollout kernel on every rank).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(num_envs, world_size, rank):
    """Contiguous block of global env indices owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(int(num_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedVectorEnv:
    """`local_env` is the per-rank engine: by default a SpaceGymVectorEnv on this rank's GPU.  Tests inject a stand-in with
    the same interface -- reset_tensors(), step_tensors(actions) -> (obs, reward, done, truncated[, terminal obs]), close(),
    attributes obs_dim and discrete -- to exercise the sharding and the collectives on CPU/gloo.

    terminal_observation=True adds the last observation of every finished episode to what step() returns (rows of envs
    that did not finish are NaN): the `info["terminal_observation"]` of a gym VectorEnv."""

    def __init__(self, env_id, num_envs, seed=0, group=None, device=None, local_env=None, terminal_observation=False, copy=True,
                 **kwargs):
        """kwargs: make_vec's (engine keywords and the reference's constructor kwargs).  copy: see the module docstring."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised (one process per GPU)")
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.num_envs = int(num_envs)
        self.lo, self.hi = shard_bounds(num_envs, self.world, self.rank)
        self.n_local = self.hi - self.lo
        self.counts = [shard_bounds(num_envs, self.world, r) for r in range(self.world)]
        self.with_terminal = bool(terminal_observation)
        if local_env is None:
            from .vector_env import make_vec
            dev_index = torch.cuda.current_device() if device is None else int(device)
            eng = make_vec(env_id, self.n_local, device=dev_index, seed=seed, env_index_base=self.lo, **kwargs)
            local_env = _TorchEngineAdapter(eng, self.with_terminal)
            self.device = torch.device("cuda", dev_index)
        else:
            self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.local = local_env
        self.obs_dim = local_env.obs_dim
        self.discrete = bool(getattr(local_env, "discrete", False))
        # rank 0's result arrays for all envs, allocated once (two sets that alternate: see the module docstring); each rank's
        # block is received straight into its slice
        self.copy = bool(copy)
        self._sets, self._cur, self._roll = None, 0, {}
        if self.rank == 0:
            self._sets = [self._alloc(()) for _ in range(2)]
        # the env's action spec: one int32 index per env for the discrete ids (spaceship_env.py:183-202), else float32 [2]
        self._act_dtype = torch.int32 if self.discrete else torch.float32
        self._act_shape = (self.n_local,) if self.discrete else (self.n_local, 2)
        self._act_local = torch.empty(self._act_shape, dtype=self._act_dtype, device=self.device)

    # ---- no communication: each rank drives its own shard (learner on the same GPU)
    def reset_local(self):
        return self.local.reset_tensors()

    def step_local(self, actions_local):
        return self.local.step_tensors(actions_local)

    def _alloc(self, lead, with_terminal=None):
        B, D, dev = self.num_envs, self.obs_dim, self.device
        f = [torch.empty(lead + (B, D), dtype=torch.float32, device=dev), torch.empty(lead + (B,), dtype=torch.float32, device=dev),
             torch.empty(lead + (B,), dtype=torch.uint8, device=dev), torch.empty(lead + (B,), dtype=torch.uint8, device=dev)]
        if self.with_terminal if with_terminal is None else with_terminal:
            f.append(torch.empty(lead + (B, D), dtype=torch.float32, device=dev))
        return f

    def _hand_out(self, fields):
        return tuple(x.clone() for x in fields) if self.copy else tuple(fields)

    # ---- single-process view on rank 0
    def _gather(self, obs, reward, done, trunc, tobs=None):
        """every rank's block of every field into rank 0's [num_envs, ...] arrays: ONE batch of point-to-point transfers
        (a single grouped launch with the nccl backend), received in place -- nothing is concatenated or re-packed"""
        local = [obs, reward, done.to(torch.uint8), trunc.to(torch.uint8)] + ([tobs] if self.with_terminal else [])
        local = [x.contiguous() for x in local]
        fields = None
        if self.rank == 0:
            self._cur ^= 1
            fields = self._sets[self._cur]
            lo, hi = self.counts[0]
            for dst, src in zip(fields, local):
                dst[lo:hi].copy_(src)
            ops = [dist.P2POp(dist.irecv, f[lo:hi], r, self.group) for r, (lo, hi) in enumerate(self.counts) if r for f in fields]
        else:
            ops = [dist.P2POp(dist.isend, x, 0, self.group) for x in local]
        if ops:
            for q in dist.batch_isend_irecv(ops):
                q.wait()
        return self._hand_out(fields) if self.rank == 0 else None

    def reset(self):
        """Rank 0 gets the first observations of all envs ([num_envs, obs_dim]; copy=False: valid until the call after next), other ranks None."""
        obs = self.local.reset_tensors()
        z = torch.zeros(self.n_local, device=self.device)
        out = self._gather(obs, z, z.to(torch.uint8), z.to(torch.uint8), torch.full_like(obs, float("nan")) if self.with_terminal else None)
        return out[0] if out is not None else None

    def _scatter(self, actions, lead=()):
        """rank 0's actions of all envs ([..., num_envs, 2] float32; discrete ids [..., num_envs] int32) -> every rank's block"""
        tail = () if self.discrete else (2,)
        local = self._act_local if not lead else torch.empty(lead + (self.n_local,) + tail, dtype=self._act_dtype, device=self.device)
        if self.world == 1:
            local.copy_(torch.as_tensor(actions, dtype=self._act_dtype, device=self.device))
            return local
        if self.rank == 0:
            a = torch.as_tensor(actions, dtype=self._act_dtype, device=self.device)
            if tuple(a.shape) != lead + (self.num_envs,) + tail:
                raise ValueError(f"actions of shape {tuple(a.shape)} for {self.num_envs} envs (discrete={self.discrete}, leading {lead})")
            chunks = [a[..., lo:hi, :].contiguous() if tail else a[..., lo:hi].contiguous() for lo, hi in self.counts]
        if _equal_sizes(self.counts):
            dist.scatter(local, chunks if self.rank == 0 else None, src=0, group=self.group)
        elif self.rank == 0:
            local.copy_(chunks[0])
            for r in range(1, self.world):
                dist.send(chunks[r], dst=r, group=self.group)
        else:
            dist.recv(local, src=0, group=self.group)
        return local

    def rollout(self, actions=None, n_steps=None):
        """K consecutive steps with one scatter and one gather: rank 0 passes the actions of all envs for all steps
        ([K, num_envs, 2] float32; discrete ids [K, num_envs] int32), other ranks None and n_steps = K.  Every rank runs the
        K-step rollout kernel on its block (sg_rollout_device: env state in registers across the steps); rank 0 gets
        (obs [K, num_envs, obs_dim], reward, done, truncated [K, num_envs]) -- every rank's [k, block] slices received in place,
        one grouped batch of point-to-point transfers per call -- other ranks None.  (Terminal observations are not part of it:
        use step(), or sg_rollout_device_terminal on the ranks.)"""
        K = int(n_steps if actions is None else len(actions))
        a_local = self._scatter(actions, lead=(K,))
        local = [x.contiguous() for x in self.local.rollout_tensors(a_local)]
        local[2], local[3] = local[2].to(torch.uint8), local[3].to(torch.uint8)
        if self.rank == 0:
            if K not in self._roll:
                self._roll[K] = ([self._alloc((K,), False) for _ in range(2)], 0)
            sets, cur = self._roll[K]
            cur ^= 1
            self._roll[K] = (sets, cur)
            fields = sets[cur]
            lo, hi = self.counts[0]
            for dst, src in zip(fields, local):
                dst[:, lo:hi].copy_(src)
            ops = [dist.P2POp(dist.irecv, f[k, lo:hi], r, self.group) for r, (lo, hi) in enumerate(self.counts) if r
                   for f in fields for k in range(K)]
        else:
            ops = [dist.P2POp(dist.isend, x[k], 0, self.group) for x in local for k in range(K)]
        for j in range(0, len(ops), 1024):  # (one grouped launch per 1024 transfers)
            for q in dist.batch_isend_irecv(ops[j:j + 1024]):
                q.wait()
        return self._hand_out(fields) if self.rank == 0 else None

    def step(self, actions=None):
        """Rank 0 passes the actions of all envs (float32 [num_envs, 2]; discrete ids: int32 [num_envs]); other ranks pass
        None.  Rank 0 gets (obs, reward, done, truncated[, terminal obs]) for all envs, other ranks None.  With copy=False these
        are the front end's own arrays: valid during the next step() / reset(), overwritten by the one after."""
        return self._gather(*self.local.step_tensors(self._scatter(actions)))

    def close(self):
        self.local.close()


def _equal_sizes(counts):
    return len({hi - lo for lo, hi in counts}) == 1


class _TorchEngineAdapter:
    """SpaceGymVectorEnv's device-tensor path behind the local-engine interface of ShardedVectorEnv"""

    def __init__(self, eng, with_terminal=False):
        self.eng, self.obs_dim, self.discrete = eng, eng.obs_dim, eng.discrete
        self._tobs = None
        if with_terminal:
            self._tobs = torch.empty((eng.num_envs, eng.obs_dim), dtype=torch.float32, device=torch.device("cuda", eng.device))

    def reset_tensors(self):
        return self.eng.reset_torch()

    def step_tensors(self, actions):
        if self._tobs is None:
            return self.eng.step_torch(actions.contiguous())
        self._tobs.fill_(float("nan"))  # the kernel writes the rows of finished envs only
        return self.eng.step_torch(actions.contiguous(), terminal_obs=self._tobs) + (self._tobs,)

    def rollout_tensors(self, actions):
        K = int(actions.shape[0])
        if getattr(self, "_rk", None) != K:
            n, D, dev = self.eng.num_envs, self.eng.obs_dim, actions.device
            self._rbuf = (torch.empty((K, n, D), dtype=torch.float32, device=dev), torch.empty((K, n), dtype=torch.float32, device=dev),
                          torch.empty((K, n), dtype=torch.uint8, device=dev), torch.empty((K, n), dtype=torch.uint8, device=dev))
            self._rk = K
        return self.eng.rollout_torch(actions.contiguous(), *self._rbuf)

    def close(self):
        self.eng.close()
