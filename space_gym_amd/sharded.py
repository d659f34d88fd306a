"""Env-sharded multi-GPU front end: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Envs never interact (gym_space/dynamic_model.py:145-165 sums only an env's own planets), so the batch is cut into
contiguous blocks -- rank k owns global envs [k*B/W, (k+1)*B/W) -- and `step_local` involves no communication at all.
The RNG is keyed by the GLOBAL env index (sg_config.env_index_base), so results do not depend on the number of ranks.

Only the single-process VectorEnv view needs a collective: `step(actions)` scatters rank 0's [B, 2] actions and gathers one
packed buffer per rank -- obs | reward | done | truncated -- back to rank 0 (a rooted gather: on the fully connected xGMI
mesh every peer uses its own link to the root once; no ring, no all-reduce).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(num_envs, world_size, rank):
    """Contiguous block of global env indices owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(int(num_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class PackedResult:
    """obs f32 [n, D] | reward f32 [n] | done u8 [n] | truncated u8 [n] in ONE flat uint8 buffer (one message per rank)."""

    def __init__(self, n, obs_dim, device):
        self.n, self.d = n, obs_dim
        self.o_obs, self.o_rew = 0, 4 * n * obs_dim
        self.o_done = self.o_rew + 4 * n
        self.o_trunc = self.o_done + n
        self.nbytes = self.o_trunc + n
        self.buf = torch.empty(self.nbytes, dtype=torch.uint8, device=device)

    @staticmethod
    def views(buf, n, d):
        o_rew = 4 * n * d
        o_done = o_rew + 4 * n
        return (buf[:o_rew].view(torch.float32).view(n, d), buf[o_rew:o_done].view(torch.float32),
                buf[o_done:o_done + n], buf[o_done + n:o_done + 2 * n])

    def fill(self, obs, reward, done, trunc):
        o, r, dn, tr = self.views(self.buf, self.n, self.d)
        o.copy_(obs); r.copy_(reward); dn.copy_(done.to(torch.uint8)); tr.copy_(trunc.to(torch.uint8))
        return self.buf


class ShardedVectorEnv:
    """`local_env` is the per-rank engine: by default a SpaceGymVectorEnv on this rank's GPU.  Tests inject a stand-in with
    the same three methods (reset_tensors, step_tensors, close) to exercise the sharding and the collectives on CPU/gloo."""

    def __init__(self, env_id, num_envs, seed=0, group=None, device=None, local_env=None, **kwargs):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised (one process per GPU)")
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.num_envs = int(num_envs)
        self.lo, self.hi = shard_bounds(num_envs, self.world, self.rank)
        self.n_local = self.hi - self.lo
        self.counts = [shard_bounds(num_envs, self.world, r) for r in range(self.world)]
        if local_env is None:
            from .vector_env import SpaceGymVectorEnv
            dev_index = torch.cuda.current_device() if device is None else int(device)
            eng = SpaceGymVectorEnv(env_id, self.n_local, device=dev_index, seed=seed, env_index_base=self.lo, **kwargs)
            local_env = _TorchEngineAdapter(eng)
            self.device = torch.device("cuda", dev_index)
        else:
            self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.local = local_env
        self.obs_dim = local_env.obs_dim
        self._packed = PackedResult(self.n_local, self.obs_dim, self.device)
        self._gather_bufs = None
        if self.rank == 0:
            self._gather_bufs = [torch.empty(PackedResult(hi - lo, self.obs_dim, "cpu").nbytes, dtype=torch.uint8,
                                             device=self.device) for lo, hi in self.counts]
        self._act_local = torch.empty((self.n_local, 2), dtype=torch.float32, device=self.device)

    # ---- no communication: each rank drives its own shard (learner on the same GPU)
    def reset_local(self):
        return self.local.reset_tensors()

    def step_local(self, actions_local):
        return self.local.step_tensors(actions_local)

    # ---- single-process view on rank 0
    def _gather(self, obs, reward, done, trunc):
        buf = self._packed.fill(obs, reward, done, trunc)
        if self.world == 1:
            parts = [buf]
        elif _equal_sizes(self.counts):
            dist.gather(buf, self._gather_bufs if self.rank == 0 else None, dst=0, group=self.group)
            parts = self._gather_bufs
        else:  # ragged shards: point-to-point into the root
            if self.rank == 0:
                self._gather_bufs[0].copy_(buf)
                reqs = [dist.irecv(self._gather_bufs[r], src=r, group=self.group) for r in range(1, self.world)]
                for q in reqs:
                    q.wait()
            else:
                dist.send(buf, dst=0, group=self.group)
            parts = self._gather_bufs
        if self.rank != 0:
            return None
        outs = [PackedResult.views(p, hi - lo, self.obs_dim) for p, (lo, hi) in zip(parts, self.counts)]
        return tuple(torch.cat([o[k] for o in outs]) for k in range(4))

    def reset(self):
        obs = self.local.reset_tensors()
        z = torch.zeros(self.n_local, device=self.device)
        out = self._gather(obs, z, z.to(torch.uint8), z.to(torch.uint8))
        return out[0] if out is not None else None

    def step(self, actions=None):
        """Rank 0 passes float32 [num_envs, 2]; other ranks pass None.  Rank 0 gets (obs, reward, done, truncated) for all
        envs, other ranks None."""
        if self.world > 1:
            if self.rank == 0:
                a = torch.as_tensor(actions, dtype=torch.float32, device=self.device)
                chunks = [a[lo:hi].contiguous() for lo, hi in self.counts]
            if _equal_sizes(self.counts):
                dist.scatter(self._act_local, chunks if self.rank == 0 else None, src=0, group=self.group)
            elif self.rank == 0:
                self._act_local.copy_(chunks[0])
                for r in range(1, self.world):
                    dist.send(chunks[r], dst=r, group=self.group)
            else:
                dist.recv(self._act_local, src=0, group=self.group)
        else:
            self._act_local.copy_(torch.as_tensor(actions, dtype=torch.float32, device=self.device))
        return self._gather(*self.local.step_tensors(self._act_local))

    def close(self):
        self.local.close()


def _equal_sizes(counts):
    return len({hi - lo for lo, hi in counts}) == 1


class _TorchEngineAdapter:
    def __init__(self, eng):
        self.eng, self.obs_dim = eng, eng.obs_dim

    def reset_tensors(self):
        return self.eng.reset_torch()

    def step_tensors(self, actions):
        return self.eng.step_torch(actions.contiguous())

    def close(self):
        self.eng.close()
