"""GPU tests of the one-launch-per-step kernels (goal_step_kernel / kepler_step_kernel) where a wave walks SEVERAL 64-env
subtiles: the carried inputs of the next subtile, the stores held back to the next pass, the replay records and restart lists
pooled per workgroup, the wave-local flush when a list fills up, and a ragged last subtile.  (The other step-kernel tests run
batches where every wave has one subtile.)  SPACEGYM_STEP_WGS_PER_CU=1 gives 256 workgroups of four waves, so 200 003 envs
are 3 126 subtiles = 13 per workgroup."""
import numpy as np
import pytest

from cases import adversarial_event_cases
from oracle import Oracle
from test_gpu_parity import check_against, make
from test_gpu_round2 import _rollout_buffers

pytestmark = pytest.mark.gpu

RAGGED = 200003  # = 64 * 3125 + 3: the last subtile has three envs


@pytest.mark.parametrize("wgs_per_cu", ["1", "2"])
@pytest.mark.parametrize("env_id,kw", [
    ("GoalContinuous3P-v0", {}),
    ("GoalContinuous2P-v0", {"auto_reset": False}),
    ("GoalContinuous4P-v0", {}),
    ("GoalDiscrete3-v0", {}),
    ("GoalContinuous3P-v0", {"steering": "acceleration"}),
    ("KeplerRandomOrbits-v0", {}),
    ("KeplerEllipseHard-v0", {"auto_reset": False}),
    ("KeplerDiscrete-v0", {}),
    ("KeplerCircleOrbit-v0", {"steering": "acceleration"}),
])
def test_step_kernel_over_many_subtiles_equals_fused_rollout(env_id, kw, wgs_per_cu, monkeypatch):
    """K launches of the step kernel (each wave walking several subtiles) against ONE launch of the wave-pair rollout kernel:
    every output of every step and the final state, bit for bit, on a ragged batch"""
    import torch
    monkeypatch.setenv("SPACEGYM_STEP_KERNEL", "single")
    monkeypatch.setenv("SPACEGYM_STEP_WGS_PER_CU", wgs_per_cu)
    n, K = RAGGED, 20
    env = make(env_id, n, seed=5, max_episode_steps=9, **kw)
    env.set_unfused_rollout(1)
    name = env.rollout_kernel(K)
    assert "pair" not in name and "step_kernel" in name, name
    gen = torch.Generator(device="cuda").manual_seed(5)
    if env.discrete:
        a = torch.randint(0, 6, (K, n), device="cuda", generator=gen, dtype=torch.int32)
    else:
        a = torch.rand((K, n, 2), device="cuda", generator=gen) * 2 - 1
    outs = []
    for mode in (0, 1):
        env.set_unfused_rollout(mode)
        env.seed(5); env.reset_torch()
        bufs = _rollout_buffers(env, K)
        env.rollout_torch(a, *bufs)
        torch.cuda.synchronize()
        env.check_status()
        outs.append([b.cpu() for b in bufs] + [env.get_state()])
        del bufs
    for name, x, y in zip(("obs", "reward", "done", "truncated"), outs[0][:4], outs[1][:4]):
        assert torch.equal(x, y), name
    for k in ("ship", "planets", "goal", "elapsed"):
        if outs[0][4][k] is not None:
            assert np.array_equal(outs[0][4][k], outs[1][4][k]), k
    done, trunc = outs[0][2], outs[0][3]
    assert int(done.sum()) > n // 2 and int(trunc.sum()) > 0
    assert bool(done[:, -3:].any())  # the three envs of the ragged subtile finished episodes too
    env.close()


@pytest.mark.parametrize("env_id", ["GoalContinuous3P-v0", "GoalContinuous4P-v0", "KeplerEllipseHard-v0"])
def test_step_kernel_terminal_observations_over_many_subtiles(env_id, monkeypatch):
    """step by step with terminal observations: the one-wave kernel walking several subtiles against the wave-pair step kernel
    (one subtile per wave pair), bit for bit"""
    import torch
    monkeypatch.setenv("SPACEGYM_STEP_WGS_PER_CU", "1")
    n, K = RAGGED, 14
    gen = torch.Generator(device="cuda").manual_seed(8)
    a = torch.rand((K, n, 2), device="cuda", generator=gen) * 2 - 1
    recs = []
    for kernel in ("single", "pair"):
        monkeypatch.setenv("SPACEGYM_STEP_KERNEL", kernel)
        env = make(env_id, n, seed=12, max_episode_steps=6)
        env.reset_torch()
        rec = []
        for t in range(K):
            tobs = torch.zeros((n, env.obs_dim), device="cuda")
            obs, rew, done, trunc = env.step_torch(a[t], terminal_obs=tobs)
            d = done.bool()
            rec.append((obs.clone(), rew.clone(), done.clone(), trunc.clone(), torch.where(d[:, None], tobs, torch.zeros_like(tobs))))
        torch.cuda.synchronize()
        env.check_status()
        recs.append((rec, env.get_state()))
        env.close()
    (r1, s1), (r2, s2) = recs
    assert sum(int(x[2].sum()) for x in r1) > n
    for t, (x, y) in enumerate(zip(r1, r2)):
        for f, (u, v) in enumerate(zip(x, y)):
            assert torch.equal(u, v), f"step {t} field {f}"
    for k in ("ship", "goal", "elapsed", "planets"):
        if s1[k] is not None:
            assert np.array_equal(s1[k], s2[k]), k


@pytest.mark.parametrize("auto_reset", [False, True])
@pytest.mark.parametrize("env_id", ["GoalContinuous3P-v0", "KeplerEllipseHard-v0"])
def test_step_kernel_lists_overflow_into_wave_local_flush(env_id, auto_reset, monkeypatch):
    """nearly every lane of every subtile ends its episode in the same step (the adversarial terminal inputs, tiled): a wave's
    replay-record and restart regions fill up before the workgroup's pooled flush, so the wave-local flush runs; outputs equal
    the wave-pair step kernel's bit for bit and the oracle's within the tolerances"""
    import torch
    monkeypatch.setenv("SPACEGYM_STEP_WGS_PER_CU", "1")
    o = Oracle(env_id, threads=16)
    s0, a, P, g = adversarial_event_cases(o, n=60000, seed=6)
    reps = 3
    s0, a = np.tile(s0, (reps, 1)), np.tile(a, (reps, 1))
    if P is not None:
        P, g = np.tile(P, (reps, 1, 1)), np.tile(g, (reps, 1))
    m = len(s0)
    assert m > 131072  # more than two subtiles per wave at 256 workgroups
    outs = []
    for kernel in ("single", "pair"):
        monkeypatch.setenv("SPACEGYM_STEP_KERNEL", kernel)
        env = make(env_id, m, seed=1, auto_reset=auto_reset)
        env.reset()
        env.set_state(ship=s0, planets=P, goal=g, elapsed=np.zeros(m, np.int32))
        obs, rew, done, info = env.step(a)
        st = env.get_state()
        outs.append((obs.copy(), rew.copy(), done.copy(), np.where(done[:, None], info["terminal_observation"], 0.0), st))
        env.close()
    x, y = outs
    assert x[2].mean() > 0.4
    for f in range(4):
        assert np.array_equal(x[f], y[f]), f
    for k in ("ship", "goal", "elapsed", "planets"):
        if x[4][k] is not None:
            assert np.array_equal(x[4][k], y[4][k]), k
    if not auto_reset:
        k = m // reps
        ref = o.step(s0[:k].astype(np.float64), a[:k], None if P is None else P[:k].astype(np.float64),
                     None if g is None else g[:k].astype(np.float64))
        assert np.array_equal(x[2][:k], ref["done"].astype(bool))
        check_against(x[0][:k], x[1][:k], x[2][:k], x[4]["ship"][:k], ref["state1"], ref["obs"], ref["reward"], ref["done"])


@pytest.mark.parametrize("n", [1, 63, 65, 257, 1000])
@pytest.mark.parametrize("env_id", ["GoalContinuous3P-v0", "KeplerRandomOrbits-v0", "GoalDiscrete3-v0"])
def test_step_kernel_on_tiny_and_ragged_batches(env_id, n, monkeypatch):
    """fewer envs than a wave, a subtile with one env, waves of a workgroup without a subtile: the one-wave step kernel (lanes
    and waves past the batch load env 0 by LDS-DMA and store nothing) against the wave-pair step kernel, bit for bit"""
    import torch
    K = 40
    gen = torch.Generator(device="cuda").manual_seed(n)
    recs = []
    for kernel in ("single", "pair"):
        monkeypatch.setenv("SPACEGYM_STEP_KERNEL", kernel)
        env = make(env_id, n, seed=3, max_episode_steps=7)
        gen.manual_seed(n)
        if env.discrete:
            a = torch.randint(0, 6, (K, n), device="cuda", generator=gen, dtype=torch.int32)
        else:
            a = torch.rand((K, n, 2), device="cuda", generator=gen) * 2 - 1
        env.reset_torch()
        rec = []
        for t in range(K):
            tobs = torch.zeros((n, env.obs_dim), device="cuda")
            obs, rew, done, trunc = env.step_torch(a[t], terminal_obs=tobs)
            rec.append((obs.clone(), rew.clone(), done.clone(), trunc.clone(), torch.where(done.bool()[:, None], tobs, torch.zeros_like(tobs))))
        torch.cuda.synchronize()
        env.check_status()
        recs.append((rec, env.get_state()))
        env.close()
    (r1, s1), (r2, s2) = recs
    assert sum(int(x[2].sum()) for x in r1) >= n  # every env finished an episode (truncation at 7 steps at the latest)
    for t, (x, y) in enumerate(zip(r1, r2)):
        for f, (u, v) in enumerate(zip(x, y)):
            assert torch.equal(u, v), f"step {t} field {f}"
    for k in ("ship", "goal", "elapsed", "planets"):
        if s1[k] is not None:
            assert np.array_equal(s1[k], s2[k]), k
