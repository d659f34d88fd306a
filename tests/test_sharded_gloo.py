"""N > 1 path on CPU: world_size-2 (and 3, ragged; 8) gloo process groups exercise the env sharding, the action scatter and the
gather of (obs, reward, done, truncated[, terminal obs]) into rank 0's preallocated arrays of space_gym_amd/sharded.py.  The per-rank engine is GPU-only, so the
ranks drive the CPU oracle as a stand-in local engine (tests may use the oracle); the check is that the sharded run equals
a single-process run of the full batch, env for env -- which also pins that the RNG is keyed by the GLOBAL env index."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

NUM_ENVS, SEED, STEPS = 510, 17, 25


class OracleLocalEngine:
    """Stand-in for the GPU engine with the adapter interface ShardedVectorEnv expects."""

    def __init__(self, env_id, n, seed, env_index_base, max_episode_steps, with_terminal=False):
        from oracle import Oracle
        self.o = Oracle(env_id)
        self.o.params.max_episode_steps = max_episode_steps
        self.n, self.seed, self.base, self.obs_dim = n, seed, env_index_base, self.o.obs_dim
        self.discrete, self.with_terminal = self.o.discrete, with_terminal
        self.envs = None

    def reset_tensors(self):
        self.envs, obs = self.o.vec_reset(self.n, seed=self.seed, env_id0=self.base)
        return torch.from_numpy(obs.astype(np.float32))

    def step_tensors(self, actions):
        out = self.o.vec_step(self.envs, actions.numpy(), seed=self.seed, env_id0=self.base, want_terminal_obs=self.with_terminal)
        res = (torch.from_numpy(out[0].astype(np.float32)), torch.from_numpy(out[1].astype(np.float32)),
               torch.from_numpy(out[2]), torch.from_numpy(out[3]))
        return res + ((torch.from_numpy(out[4].astype(np.float32)),) if self.with_terminal else ())

    def rollout_tensors(self, actions):
        steps = [self.step_tensors(a)[:4] for a in actions]
        return tuple(torch.stack([st[k] for st in steps]) for k in range(4))

    def close(self):
        pass


def _actions(t, discrete=False):
    rng = np.random.default_rng(1000 + t)
    return rng.integers(0, 6, NUM_ENVS).astype(np.int32) if discrete else rng.uniform(-1, 1, size=(NUM_ENVS, 2)).astype(np.float32)


KEYS = ("obs", "rew", "done", "trunc", "tobs")


def _worker(rank, world, port, out_path, env_id, with_terminal):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from space_gym_amd.sharded import ShardedVectorEnv, shard_bounds
    lo, hi = shard_bounds(NUM_ENVS, world, rank)
    eng = OracleLocalEngine(env_id, hi - lo, SEED, lo, max_episode_steps=12, with_terminal=with_terminal)
    env = ShardedVectorEnv(env_id, NUM_ENVS, seed=SEED, local_env=eng, device="cpu", terminal_observation=with_terminal, copy=False)
    assert (env.lo, env.hi) == (lo, hi) and env.discrete == ("Discrete" in env_id)
    # copy=False: rank 0's results are the front end's own arrays, two sets that alternate -- what a call returns stays valid
    # during the next call (checked below) and is overwritten by the one after; no per-step allocation or concatenation
    def keep(res):
        return None if res is None else res.clone() if torch.is_tensor(res) else tuple(x.clone() for x in res)
    prev = env.reset()
    trace = [keep(prev)]
    seen, prev_copy = set(), trace[0]
    for t in range(STEPS):
        res = env.step(_actions(t, env.discrete) if rank == 0 else None)
        if rank == 0:
            seen.add(tuple(x.data_ptr() for x in res))
            now = prev if torch.is_tensor(prev) else prev[0]  # the previous call's observations: still intact
            assert torch.equal(now, prev_copy if torch.is_tensor(prev_copy) else prev_copy[0], ) or torch.isnan(now).any()
            prev, prev_copy = res, keep(res)
        trace.append(keep(res))
    if rank == 0:
        assert len(seen) == 2  # two result sets, alternating
    if rank == 0:
        assert all(len(step) == (5 if with_terminal else 4) for step in trace[1:])
        np.savez(out_path, reset_obs=trace[0].numpy(), **{f"{k}{t}": v.numpy() for t, step in enumerate(trace[1:])
                                                      for k, v in zip(KEYS, step)})
    else:
        assert all(x is None for x in trace)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,env_id,with_terminal", [(2, "GoalContinuous3P-v0", False), (3, "GoalContinuous3P-v0", True),
                                                        (2, "GoalDiscrete3-v0", True), (3, "KeplerDiscrete-v0", False),
                                                        (8, "GoalContinuous4P-v0", True)])  # config 5's id and rank count
def test_sharded_equals_single_process(world, env_id, with_terminal, tmp_path):
    """continuous and discrete action specs, with and without terminal observations, equal and ragged shards"""
    out = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(world, _free_port(), out, env_id, with_terminal), nprocs=world, join=True)
    got = np.load(out)
    ref = OracleLocalEngine(env_id, NUM_ENVS, SEED, 0, max_episode_steps=12, with_terminal=with_terminal)
    assert np.array_equal(got["reset_obs"], ref.reset_tensors().numpy())
    n_done = 0
    for t in range(STEPS):
        res = [x.numpy() for x in ref.step_tensors(torch.from_numpy(_actions(t, ref.discrete)))]
        obs, rew, done, trunc = res[:4]
        assert np.array_equal(got[f"obs{t}"], obs) and np.array_equal(got[f"rew{t}"], rew)
        assert np.array_equal(got[f"done{t}"], done) and np.array_equal(got[f"trunc{t}"], trunc)
        if with_terminal:  # rows of finished envs carry their last observation, the others NaN
            assert np.array_equal(got[f"tobs{t}"], res[4], equal_nan=True)
            assert np.isfinite(res[4][done.astype(bool)]).all() and np.isnan(res[4][~done.astype(bool)]).all()
        n_done += int(done.sum())
    assert n_done >= NUM_ENVS  # truncation at 12 steps: every env restarted at least once, on every rank


def test_shard_bounds_partition():
    from space_gym_amd.sharded import shard_bounds
    for n, w in [(65536, 8), (524288, 8), (510, 3), (7, 8), (1, 1)]:
        b = [shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1


def _rollout_worker(rank, world, port, out_path, env_id):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from space_gym_amd.sharded import ShardedVectorEnv, shard_bounds
    lo, hi = shard_bounds(NUM_ENVS, world, rank)
    eng = OracleLocalEngine(env_id, hi - lo, SEED, lo, max_episode_steps=12)
    env = ShardedVectorEnv(env_id, NUM_ENVS, seed=SEED, local_env=eng, device="cpu")  # copy=True: fresh arrays every call
    obs0 = env.reset()
    K = 10
    out = {}
    for c in range(2):
        acts = np.stack([_actions(c * K + t, env.discrete) for t in range(K)])
        res = env.rollout(acts if rank == 0 else None, n_steps=K)
        if rank == 0:
            assert tuple(res[0].shape) == (K, NUM_ENVS, env.obs_dim) and tuple(res[2].shape) == (K, NUM_ENVS)
            out.update({f"{k}{c}": v.numpy() for k, v in zip(KEYS, res)})
        else:
            assert res is None
    if rank == 0:
        assert out["obs0"].ctypes.data != out["obs1"].ctypes.data  # copies: the first chunk's arrays were not overwritten
        np.savez(out_path, reset_obs=obs0.numpy(), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,env_id", [(2, "GoalContinuous3P-v0"), (3, "GoalDiscrete3-v0")])
def test_sharded_rollout_one_gather_per_k_steps(world, env_id, tmp_path):
    """rollout(): K steps per call, one scatter and one gather; equal to a single-process run, equal and ragged shards"""
    out = str(tmp_path / "rank0.npz")
    mp.spawn(_rollout_worker, args=(world, _free_port(), out, env_id), nprocs=world, join=True)
    got = np.load(out)
    ref = OracleLocalEngine(env_id, NUM_ENVS, SEED, 0, max_episode_steps=12)
    assert np.array_equal(got["reset_obs"], ref.reset_tensors().numpy())
    for t in range(20):
        obs, rew, done, trunc = [x.numpy() for x in ref.step_tensors(torch.from_numpy(_actions(t, ref.discrete)))]
        c, k = divmod(t, 10)
        assert np.array_equal(got[f"obs{c}"][k], obs) and np.array_equal(got[f"rew{c}"][k], rew)
        assert np.array_equal(got[f"done{c}"][k], done) and np.array_equal(got[f"trunc{c}"][k], trunc)
