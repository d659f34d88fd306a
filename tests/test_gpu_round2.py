"""GPU tests added in round 2: the rollout path checked DIRECTLY against the oracle, adversarial events and reset statistics
on the card, terminal observations of rollouts, complete snapshots, stream ordering, the large-batch plan branch."""
import numpy as np
import pytest

from cases import adversarial_event_cases, check_reset_statistics
from conftest import FAMILIES, load_golden
from oracle import Oracle
from test_gpu_parity import TOL_OBS, TOL_REWARD_REL, TOL_STATE, check_against, make

pytestmark = pytest.mark.gpu


def _rollout_buffers(env, K):
    import torch
    n, D = env.num_envs, env.obs_dim
    return (torch.empty((K, n, D), device="cuda"), torch.empty((K, n), device="cuda"),
            torch.empty((K, n), dtype=torch.uint8, device="cuda"), torch.empty((K, n), dtype=torch.uint8, device="cuda"))


@pytest.mark.parametrize("env_id,n,K", [("GoalContinuous3P-v0", 65536, 64), ("GoalContinuous4P-v0", 16384, 48),
                                         ("KeplerCircleOrbit-v0", 65536, 64), ("KeplerRandomOrbits-v0", 65536, 48),
                                         ("GoalDiscrete3-v0", 16384, 48), ("KeplerDiscrete-v0", 16384, 48)])
def test_pair_rollout_matches_oracle_directly(env_id, n, K):
    """sg_rollout_device (ONE launch of the wave-pair kernel: pilot / finisher hand-off, probe steps, replayed terminal
    env-steps, episode queue) against the oracle, step by step, at the headline batch -- not through the step kernel: both
    families (BASELINE configs 3 and 4), per-env orbits, and the discrete-action ids.  The oracle needs the exact pre-step
    state of every step; a second handle with the same seed is stepped one launch at a time with sg_get_state in between
    (its outputs must be bit-identical to the rollout's, which is asserted, so its states are the rollout's)."""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(5)
    if "Discrete" in env_id:
        a = torch.randint(0, 6, (K, n), device="cuda", generator=gen, dtype=torch.int32)
    else:
        a = torch.rand((K, n, 2), device="cuda", generator=gen) * 2 - 1
    goal = env_id.startswith("Goal")
    # (KeplerDiscrete-v0 is registered without a TimeLimit, keyboard_agent.py:10-74: episodes only end by events)
    kw = dict(max_episode_steps=200) if env_id == "KeplerDiscrete-v0" else {}
    env = make(env_id, n, seed=13, **kw)
    assert env.rollout_kernel(K).startswith("goal_pair_rollout_kernel" if goal else "kepler_pair_rollout_kernel")
    env.reset_torch()
    pre = 40  # let episodes age: restarts, goal hits and a filled episode queue are all in play during the checked steps
    obs, rew, done, trunc = _rollout_buffers(env, K)
    env.rollout_torch(a[:pre], obs[:pre], rew[:pre], done[:pre], trunc[:pre])
    term = env.terminal_list_torch(capacity=n * K // 8)
    env.rollout_torch(a, obs, rew, done, trunc, terminal=term)
    env.check_status()
    obs, rew, done, trunc = (x.cpu().numpy() for x in (obs, rew, done, trunc))
    t_step, t_env, t_obs = env.terminal_records(term)
    last_obs = obs.copy()
    last_obs[t_step, t_env] = t_obs
    assert len(t_step) == int(done.sum())  # one record per finished env-step
    env.close()

    ref_env = make(env_id, n, seed=13, **kw)
    o = Oracle(env_id, threads=16)
    ref_env.reset()
    a_h = a.cpu().numpy()
    for t in range(pre):
        ref_env.step(a_h[t])
    worst = total = 0
    for t in range(K):
        st = ref_env.get_state()
        ob, rw, dn, info = ref_env.step(a_h[t])
        # the two handles are the same env bit for bit
        assert np.array_equal(ob, obs[t]) and np.array_equal(rw, rew[t]) and np.array_equal(dn, done[t].astype(bool))
        if goal:
            ref = o.step(st["ship"].astype(np.float64), a_h[t], st["planets"].astype(np.float64), st["goal"].astype(np.float64))
        elif env_id == "KeplerRandomOrbits-v0":  # per-env reference orbit (angle, eccentricity, a): kepler.py:257-259
            orbit = np.concatenate([st["goal"].astype(np.float64), np.full((n, 1), 1.2)], 1)
            ref = o.step(st["ship"].astype(np.float64), a_h[t], orbit=orbit)
        else:
            ref = o.step(st["ship"].astype(np.float64), a_h[t])
        ref_done = ref["done"].astype(bool) | trunc[t].astype(bool)
        same = done[t].astype(bool) == ref_done
        worst, total = max(worst, int((~same).sum())), total + int((~same).sum())
        assert (~same).sum() <= max(2, n // 20000)  # inputs within fp32 rounding of an event boundary
        assert np.abs(last_obs[t][same] - ref["obs"][same]).max() <= TOL_OBS
        rel = np.abs(rew[t][same] - ref["reward"][same]) / np.maximum(1.0, np.abs(ref["reward"][same]))
        assert rel.max() <= TOL_REWARD_REL, (t, rel.max())
        # the 7 leading observation slots are the state (x, y, cos, sin, vx, vy, omega)
        s1 = ref["state1"][same]
        assert np.abs(last_obs[t][same][:, [0, 1, 4, 5, 6]] - s1[:, [0, 1, 3, 4, 5]]).max() <= TOL_STATE
    print(f"done-flag disagreements with the oracle: {total} in {n * K} env-steps (worst step {worst})")
    assert total <= 2  # measured: 0 in 4 194 304 env-steps (an input within fp32 rounding of an event boundary could flip one)
    assert done.sum() > (n // 2 if goal else n // 8)  # restarts, and with them the replay of terminal steps, were exercised throughout
    ref_env.close()


@pytest.mark.parametrize("kernel", ["pair", "single", "steps"])
def test_event_counters_match_outputs_and_oracle(kernel, monkeypatch):
    """sg_set_counters / sg_get_counters: env-steps, finished episodes and truncations equal the sums of the flags the steps
    wrote; goals reached equal the oracle's goal-hit flags on the same transitions (terminal steps included) -- for the
    wave-pair rollout kernel (hits of replayed terminal steps are counted by their replay), the one-wave rollout kernel and
    one launch per step."""
    import torch
    env_id, n, K, pre = "GoalContinuous3P-v0", 4096, 50, 30
    if kernel != "steps":
        monkeypatch.setenv("SPACEGYM_ROLLOUT_KERNEL", kernel)
    a = torch.rand((pre + K, n, 2), device="cuda", generator=torch.Generator(device="cuda").manual_seed(6)) * 2 - 1
    env = make(env_id, n, seed=3, max_episode_steps=45)
    env.set_unfused_rollout(kernel == "steps")
    env.reset_torch()
    obs, rew, done, trunc = _rollout_buffers(env, pre + K)
    env.rollout_torch(a[:pre], obs[:pre], rew[:pre], done[:pre], trunc[:pre])
    assert env.counters() == dict(env_steps=0, episodes_finished=0, truncations=0, goal_hits=0)  # off: nothing is counted
    env.set_counters(True)
    env.rollout_torch(a[pre:], obs[pre:], rew[pre:], done[pre:], trunc[pre:])
    k = env.counters()
    env.close()
    assert k["env_steps"] == n * K
    assert k["episodes_finished"] == int(done[pre:].sum()) and k["truncations"] == int(trunc[pre:].sum()) and k["truncations"] > 0
    # the oracle's goal hits on the same transitions: a twin handle stepped one launch at a time gives the pre-step states
    ref_env = make(env_id, n, seed=3, max_episode_steps=45)
    ref_env.set_counters(True)
    o = Oracle(env_id, threads=16)
    ref_env.reset()
    a_h = a.cpu().numpy()
    hits = 0
    for t in range(pre + K):
        st = ref_env.get_state()
        if t == pre:
            ref_env.counters(reset=True)
        ref_env.step(a_h[t])
        if t >= pre:
            hits += int(o.step(st["ship"].astype(np.float64), a_h[t], st["planets"].astype(np.float64), st["goal"].astype(np.float64))["goal_hit"].sum())
    k2 = ref_env.counters()
    ref_env.close()
    assert k2 == k  # NumPy path, one launch per step: the same counts
    assert hits > 50 and abs(k["goal_hits"] - hits) <= 2  # (a position within fp32 rounding of the goal radius could flip one)


@pytest.mark.parametrize("env_id", ["GoalContinuous2P-v0", "GoalContinuous3P-v0", "KeplerEllipseHard-v0"])
def test_event_roots_on_grazing_and_corner_cases_gpu(env_id):
    """the adversarial terminal steps of tests/test_host_twin.py on the card (v_rsq / v_rcp / v_log / v_exp instead of libm):
    injected through sg_set_state, one sg_step, against the oracle"""
    o = Oracle(env_id, threads=16)
    outs = []
    for seed in (3, 4, 5):
        s0, a, P, g = adversarial_event_cases(o, n=60000, seed=seed)
        m = len(s0)
        env = make(env_id, m, seed=1, auto_reset=False)
        env.reset()
        env.set_state(ship=s0, planets=P, goal=g, elapsed=np.zeros(m, np.int32))
        obs, rew, done, info = env.step(a)
        s1 = env.get_state()["ship"]
        env.close()
        ref = o.step(s0.astype(np.float64), a, None if P is None else P.astype(np.float64), None if g is None else g.astype(np.float64))
        assert (ref["done"] == 1).mean() > 0.4
        assert np.array_equal(done, ref["done"].astype(bool))  # no event decision differs
        check_against(obs, rew, done, s1, ref["state1"], ref["obs"], ref["reward"], ref["done"])
        outs.append(m)
    assert sum(outs) > 150000


@pytest.mark.parametrize("env_id", ["GoalContinuous3P-v0", "GoalContinuous4P-v0", "KeplerEllipseHard-v0"])
def test_tangential_grazes_decide_like_scipy_gpu(env_id):
    """tests/test_host_twin.py::test_tangential_grazes_decide_like_scipy on the card: paths that touch a surface in the middle
    of the env-step (within 0.3 mm), where a step over the whole env-step must not be kept.  Decisions equal the fp64
    oracle's up to roots that are tangent within fp32 rounding; states and rewards within the tolerances except for a
    handful of exactly tangent terminal states; non-terminal env-steps all within them."""
    from cases import tangential_graze_cases
    o = Oracle(env_id, threads=16)
    s0, a, P, g = tangential_graze_cases(o, n=60000, seed=4)
    m = len(s0)
    env = make(env_id, m, seed=1, auto_reset=False)
    env.reset()
    env.set_state(ship=s0, planets=P, goal=g, elapsed=np.zeros(m, np.int32))
    obs, rew, done, info = env.step(a)
    s1 = env.get_state()["ship"]
    env.close()
    ref = o.step(s0.astype(np.float64), a, None if P is None else P.astype(np.float64), None if g is None else g.astype(np.float64))
    term = ref["done"] == 1
    assert 0.2 < term.mean() < 0.9 and m > 30000
    same = done == ref["done"].astype(bool)
    assert (~same).sum() <= 2
    rel = np.abs(rew - ref["reward"])[same] / np.maximum(1, np.abs(ref["reward"][same]))
    assert (rel > TOL_REWARD_REL).sum() <= 5 and rel.max() <= 1e-4 and not (rel > TOL_REWARD_REL)[~term[same]].any()
    ds = np.abs(s1[same][:, [0, 1, 3, 4, 5]] - ref["state1"][same][:, [0, 1, 3, 4, 5]]).max(1)
    assert (ds > TOL_STATE).sum() <= 3 and not (ds > TOL_STATE)[~term[same]].any()


@pytest.mark.parametrize("fam", ["goal2p", "goal3p", "goal4p"])
def test_reset_distribution_matches_reference_gpu(fam):
    """GPU-generated resets and goal-resample chains against the statistics of 1e5 resets of the reference
    (tests/golden/reset_*.npz): the reset kernel, then `hits` goal hits forced by parking the ship on its goal (engine off,
    no velocity) so that every env resamples in every step (goal.py:154-157 -> hexagonal_tiling.py:95-128)."""
    ref = load_golden("reset_" + fam)
    n, hits = int(ref["n_resets"]), int(ref["n_hits"])
    env = make(FAMILIES[fam], n, seed=777, auto_reset=False)
    env.reset()
    st0 = env.get_state()
    cols0 = env.snapshot_columns(env.save_state())
    goals, ship_tile, goal_tile, free = [st0["goal"].copy()], [], [], []

    def tiles(c):
        ship_tile.append((c["aux"][:, 1] & 0xff).astype(np.int64)); goal_tile.append(((c["aux"][:, 1] >> 8) & 0xff).astype(np.int64))
        free.append(c["aux"][:, 2].astype(np.uint64) | (c["aux"][:, 3].astype(np.uint64) << np.uint64(32)))
    tiles(cols0)
    off = np.zeros((n, 2), np.float32); off[:, 0] = -1.0  # engine off, no turn
    park = np.zeros((n, 6), np.float32)
    alive = np.ones(n, bool)
    for k in range(hits):
        park[:, :2] = goals[-1]
        env.set_state(ship=park)
        obs, rew, done, info = env.step(off)
        alive &= ~done
        c = env.snapshot_columns(env.save_state())
        assert (c["aux"][alive, 0] == k + 1).all()  # every parked env hit its goal and drew a new one
        goals.append(env.get_state()["goal"].copy())
        tiles(c)
    env.close()
    assert alive.mean() > 0.999  # (a ship parked on a goal next to a planet may drift into it within the step)
    flags = np.array([((cols0["aux"][:, 1] >> 16) & 1).sum(), ((cols0["aux"][:, 1] >> 17) & 1).sum()])
    check_reset_statistics(ref, fam, st0["ship"][alive], st0["planets"][alive], np.stack(goals, 1)[alive], np.stack(ship_tile, 1)[alive],
                           np.stack(goal_tile, 1)[alive], np.stack(free, 1)[alive], cols0["cshift"][alive], flags * alive.mean())


@pytest.mark.parametrize("env_id,kernel", [("GoalContinuous3P-v0", "pair"), ("GoalContinuous2P-v0", "single"),
                                           ("KeplerCircleOrbit-v0", "pair"), ("GoalDiscrete4-v0", "pair"),
                                           ("KeplerRandomOrbits-v0", "single"), ("GoalContinuous4P-v0", "unfused")])
def test_rollout_terminal_observations(env_id, kernel, monkeypatch):
    """sg_rollout_device_terminal: one record per finished env-step with the LAST observation of the episode that ended
    (what the reference's step returns with done=True, spaceship_env.py:75-78), bit-identical to the step kernel's
    terminal_obs rows; the ordinary outputs are those of sg_rollout_device.  Every plan: the wave-pair and one-wave K-step
    kernels of both families, and one launch per step with the rows compacted after each."""
    import torch
    if kernel != "unfused":
        monkeypatch.setenv("SPACEGYM_ROLLOUT_KERNEL", kernel)
    n, K = 8192, 160
    gen = torch.Generator(device="cuda").manual_seed(3)
    a = (torch.randint(0, 6, (K, n), device="cuda", generator=gen, dtype=torch.int32) if "Discrete" in env_id
         else torch.rand((K, n, 2), device="cuda", generator=gen) * 2 - 1)
    env = make(env_id, n, seed=6, max_episode_steps=45)
    env.set_unfused_rollout(kernel == "unfused")
    assert kernel == "unfused" or (kernel + "_rollout" in env.rollout_kernel(60)) == (kernel == "pair")
    env.reset_torch()
    obs, rew, done, trunc = _rollout_buffers(env, K)
    term = env.terminal_list_torch(capacity=n * K // 4)
    env.rollout_torch(a[:60], obs[:60], rew[:60], done[:60], trunc[:60], terminal=term)
    first = env.terminal_records(term)
    env.rollout_torch(a[60:], obs[60:], rew[60:], done[60:], trunc[60:], terminal=term)  # the count restarts with every call
    second = env.terminal_records(term)
    env.check_status()
    env.close()
    t_step = np.concatenate([first[0], second[0] + 60]); t_env = np.concatenate([first[1], second[1]])
    t_obs = np.concatenate([first[2], second[2]])
    # step by step with the step kernel's terminal_obs
    ref = make(env_id, n, seed=6, max_episode_steps=45)
    ref.reset_torch()
    tob = torch.full((n, ref.obs_dim), float("nan"), device="cuda")
    want_rows, want_obs = [], []
    for t in range(K):
        ob, rw, dn, tr = ref.step_torch(a[t].contiguous(), terminal_obs=tob)
        torch.cuda.synchronize()
        assert torch.equal(ob, obs[t]) and torch.equal(rw, rew[t]) and torch.equal(dn, done[t]) and torch.equal(tr, trunc[t])
        idx = torch.nonzero(dn).flatten().cpu().numpy()
        want_rows += [(t, int(i)) for i in idx]
        want_obs.append(tob[idx].cpu().numpy())
    ref.close()
    want_obs = np.concatenate(want_obs)
    assert len(want_rows) > n and len(want_rows) == len(t_step)
    assert np.array_equal(np.array(want_rows), np.stack([t_step, t_env], 1))
    assert np.array_equal(want_obs, t_obs)
    assert (trunc.sum() > 0) and (done.sum() > trunc.sum())  # both kinds of endings were covered


def test_terminal_list_overflow_is_reported():
    import torch
    n, K = 4096, 64
    env = make("GoalContinuous3P-v0", n, seed=6, max_episode_steps=10)
    env.reset_torch()
    a = torch.rand((K, n, 2), device="cuda") * 2 - 1
    obs, rew, done, trunc = _rollout_buffers(env, K)
    term = env.terminal_list_torch(capacity=100)
    env.rollout_torch(a, obs, rew, done, trunc, terminal=term)
    torch.cuda.synchronize()
    assert int(term["count"].item()) == int(done.sum().item()) > 100
    with pytest.raises(OverflowError):
        env.terminal_records(term)
    env.close()


@pytest.mark.parametrize("env_id", ["GoalContinuous2P-v0", "GoalContinuous4P-v0", "KeplerRandomOrbits-v0"])
def test_snapshot_restore_is_bit_identical(env_id):
    """save_state -> K steps -> load_state -> the same K steps again: every output identical, with goal hits (the tiling's
    free-tile multiset, ship / goal tile and goal-draw counter, hexagonal_tiling.py:99-128), restarts and truncations in
    between; and a snapshot loaded into a NEW handle continues the same way."""
    import torch
    n, K = 8192, 120
    a = torch.rand((2 * K, n, 2), device="cuda", generator=torch.Generator(device="cuda").manual_seed(8)) * 2 - 1
    env = make(env_id, n, seed=31, max_episode_steps=50)
    env.reset_torch()
    obs, rew, done, trunc = _rollout_buffers(env, K)
    env.rollout_torch(a[:K], obs, rew, done, trunc)  # some history first
    blob = env.save_state()
    cols = env.snapshot_columns(blob)
    if env_id.startswith("Goal"):
        assert (cols["aux"][:, 0] > 0).sum() > n // 50  # envs that have hit a goal in their current episode
    runs = []
    for rep in range(2):
        env.rollout_torch(a[K:], obs, rew, done, trunc)
        torch.cuda.synchronize()
        runs.append([x.cpu().numpy().copy() for x in (obs, rew, done, trunc)] + [env.save_state()])
        env.load_state(blob)
    other = make(env_id, n, seed=999, max_episode_steps=50)  # another seed: the snapshot carries the RNG key
    other.load_state(blob)
    rec = []
    for t in range(K):  # and through the one-launch-per-step kernel
        o_, r_, d_, info = other.step(a[K + t].cpu().numpy())
        rec.append((o_, r_, d_))
    for x, y in zip(runs[0], runs[1]):
        assert np.array_equal(x, y)
    for t in range(K):
        assert np.array_equal(rec[t][0], runs[0][0][t]) and np.array_equal(rec[t][1], runs[0][1][t])
        assert np.array_equal(rec[t][2], runs[0][2][t].astype(bool))
    assert np.array_equal(other.save_state()[48:], runs[0][4][48:])
    assert runs[0][2].sum() > n
    with pytest.raises(Exception):
        make("GoalContinuous3P-v0", n).load_state(blob)
    env.close(); other.close()


def test_host_calls_are_ordered_after_device_calls():
    """host-buffer calls (the handle's own stream) wait for the *_device work enqueued on the caller's stream: no manual
    synchronisation between step_torch / rollout_torch and get_state / reset / set_state"""
    import torch
    n, K = 65536, 200
    a = torch.rand((K, n, 2), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) * 2 - 1
    synced, unsynced = make("GoalContinuous3P-v0", n, seed=4), make("GoalContinuous3P-v0", n, seed=4)
    bufs = _rollout_buffers(synced, K)
    for env, sync in ((synced, True), (unsynced, False)):
        env.reset_torch()
        env.rollout_torch(a, *bufs)          # ~0.6 ms of work in flight
        if sync:
            torch.cuda.synchronize()
        st = env.get_state()
        env.step_torch(a[0].contiguous())
        if sync:
            torch.cuda.synchronize()
        st2 = env.get_state()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):       # a second caller stream
            env.rollout_torch(a[:50], *(b[:50] for b in bufs))
        if sync:
            torch.cuda.synchronize()
        blob = env.save_state()
        env.results = (st, st2, blob)
    for x, y in zip(synced.results[:2], unsynced.results[:2]):
        for k in ("ship", "planets", "goal", "elapsed"):
            assert np.array_equal(x[k], y[k]), k
    assert np.array_equal(synced.results[2], unsynced.results[2])
    unsynced.check_status()
    synced.close(); unsynced.close()


def test_large_batch_plan_branch_4p():
    """config 5's per-GPU plan at its real size on one card: GoalContinuous4P-v0 with 524 288 envs (2 048 workgroups, eight
    per CU: the wave-pair rollout kernel's workgroups take turns on the CUs, one at a time each).  The fused rollout equals
    the step kernel bit for bit and the invariants hold."""
    import torch
    n, K = 524288, 24
    env = make("GoalContinuous4P-v0", n, seed=2, max_episode_steps=12)
    assert env.rollout_kernel(K) == "goal_pair_rollout_kernel<4, false, 3, false>"
    a = torch.rand((K, n, 2), device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)) * 2 - 1
    outs = []
    for mode in (0, 1):
        env.set_unfused_rollout(mode)
        env.seed(2); env.reset_torch()
        bufs = _rollout_buffers(env, K)
        env.rollout_torch(a, *bufs)
        torch.cuda.synchronize()
        outs.append([b.cpu() for b in bufs] + [env.get_state()])
        del bufs
    for x, y in zip(outs[0][:4], outs[1][:4]):
        assert torch.equal(x, y)
    for k in ("ship", "planets", "goal", "elapsed"):
        assert np.array_equal(outs[0][4][k], outs[1][4][k])
    obs, rew, done, trunc = outs[0][:4]
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert int(done.sum()) > n and int(trunc.sum()) > n // 2
    assert (obs[..., :2].abs() <= 1.5).all() and ((obs[..., 2] ** 2 + obs[..., 3] ** 2 - 1).abs() < 1e-5).all()
    env.close()


def test_bench_two_ranks_as_a_plain_command():
    """`python bench.py --gpus 2` spawns its ranks itself (rehearsal mode: both on this card, gloo) and prints one line"""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["SG_BENCH_REHEARSE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                          "--batch", "8192", "--preroll", "200"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["config"]["global_batch"] == 16384 and "cpu_baseline" not in b
    assert abs(b["value"] - 2 * 8192 * 20 / (b["ms_per_step"] * 20e-3)) / b["value"] < 1e-6
    # the N > 1 line explains itself: backend and world size as torch.distributed reports them, every rank's device and
    # its own clock for the timed region, and the leg with the gather to rank 0
    d = b["distributed"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and [r["rank"] for r in d["ranks"]] == [0, 1]
    assert all(r["ms_per_step"] > 0 and r["device"] for r in d["ranks"])
    assert max(r["ms_per_step"] for r in d["ranks"]) == pytest.approx(b["ms_per_step"], rel=1e-9)
    assert b["ms_per_step_with_rccl_gather"] > 0 and b["value_with_rccl_gather"] > 0


@pytest.mark.parametrize("env_id", ["GoalContinuous3P-v0", "GoalDiscrete3-v0"])
def test_sharded_env_on_the_gpu_adapter(env_id):
    """ShardedVectorEnv over the real engine (_TorchEngineAdapter, device tensors), one rank: what the gloo tests cannot
    reach.  Equals a plain SpaceGymVectorEnv step for step, terminal observations included, for both action specs."""
    import os, socket
    import torch
    import torch.distributed as dist
    from space_gym_amd.sharded import ShardedVectorEnv
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        n = 4096
        sh = ShardedVectorEnv(env_id, n, seed=12, device=0, terminal_observation=True, max_episode_steps=30)
        ref = make(env_id, n, seed=12, max_episode_steps=30)
        assert sh.discrete == ref.discrete and (sh.lo, sh.hi) == (0, n)
        assert np.array_equal(sh.reset().cpu().numpy(), ref.reset())
        rng = np.random.default_rng(4)
        n_done = 0
        for t in range(80):
            a = rng.integers(0, 6, n).astype(np.int32) if ref.discrete else rng.uniform(-1, 1, (n, 2)).astype(np.float32)
            obs, rew, done, trunc, tobs = (x.cpu().numpy() for x in sh.step(a))
            o2, r2, d2, info = ref.step(a)
            assert np.array_equal(obs, o2) and np.array_equal(rew, r2) and np.array_equal(done.astype(bool), d2)
            assert np.array_equal(trunc.astype(bool), info["TimeLimit.truncated"])
            assert np.array_equal(tobs[d2], info["terminal_observation"][d2]) and np.isnan(tobs[~d2]).all()
            n_done += int(d2.sum())
        assert n_done > n
        sh.close(); ref.close()
    finally:
        dist.destroy_process_group()


def test_sg_create_sharded_blocks_equal_one_handle():
    """sg_create_sharded: contiguous blocks (ragged: 1000 envs over 3 handles, all on this card) with env_index_base set per
    block reproduce one handle of the whole batch, resets and steps"""
    import ctypes as C
    from space_gym_amd import _native
    lib = _native.load()
    n, nd = 1000, 3
    cfg = _native.SgConfig(env_id=b"GoalContinuous3P-v0", num_envs=n, seed=41, env_index_base=500, max_episode_steps=25, auto_reset=1, steering=0)
    handles = (C.c_void_p * nd)()
    devs = (C.c_int * nd)(0, 0, 0)
    assert lib.sg_create_sharded(C.byref(cfg), nd, devs, handles) == 0
    sizes = [int(lib.sg_num_envs(C.c_void_p(h))) for h in handles]
    assert sizes == [334, 333, 333]
    full = make("GoalContinuous3P-v0", n, seed=41, env_index_base=500, max_episode_steps=25)
    D = full.obs_dim
    parts = []
    for h, m in zip(handles, sizes):
        o = np.empty((m, D), np.float32)
        assert lib.sg_reset(C.c_void_p(h), o.ctypes.data_as(C.c_void_p)) == 0
        parts.append(o)
    assert np.array_equal(np.concatenate(parts), full.reset())
    rng = np.random.default_rng(0)
    for t in range(40):
        a = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
        of, rf, df, _ = full.step(a)
        lo, outs = 0, []
        for h, m in zip(handles, sizes):
            o, r = np.empty((m, D), np.float32), np.empty(m, np.float32)
            d, tr = np.empty(m, np.uint8), np.empty(m, np.uint8)
            ak = np.ascontiguousarray(a[lo:lo + m])
            assert lib.sg_step(C.c_void_p(h), ak.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                               d.ctypes.data_as(C.c_void_p), tr.ctypes.data_as(C.c_void_p), None) == 0
            outs.append((o, r, d))
            lo += m
        assert np.array_equal(np.concatenate([x[0] for x in outs]), of) and np.array_equal(np.concatenate([x[1] for x in outs]), rf)
        assert np.array_equal(np.concatenate([x[2] for x in outs]).astype(bool), df)
    for h in handles:
        lib.sg_destroy(C.c_void_p(h))
    bad = (C.c_int * 1)(99)
    one = (C.c_void_p * 1)()
    assert lib.sg_create_sharded(C.byref(cfg), 1, bad, one) != 0 and not one[0]
    full.close()


def test_actions_out_of_range_are_rejected_by_default():
    """the reference's step asserts action_space.contains(raw_action) (spaceship_env.py:71) / raises ValueError for a
    discrete index out of range (:201-202): the NumPy front end does the same unless validate_actions=False (device clamp)"""
    n = 256
    env = make("GoalContinuous2P-v0", n, seed=1)
    env.reset()
    a = np.zeros((n, 2), np.float32)
    a[7, 0] = 1.5
    with pytest.raises(AssertionError):
        env.step(a)
    env.close()
    env = make("GoalDiscrete2-v0", n, seed=1)
    env.reset()
    k = np.zeros(n, np.int32)
    k[3] = 6
    with pytest.raises(ValueError):
        env.step(k)
    env.close()
    env = make("GoalContinuous2P-v0", n, seed=1, validate_actions=False)
    env.reset()
    obs, rew, done, info = env.step(a)  # clamped into [-1, 1] on the device
    assert np.isfinite(obs).all() and np.isfinite(rew).all()
    env.close()


def test_multi_device_front_end_equals_one_handle():
    """MultiDeviceVectorEnv (single process, one native handle per listed device through sg_create_sharded_ex): three blocks --
    all on device 0 here, the only GPU of the box -- are the same envs as one handle of the whole batch, bit for bit: reset,
    steps (per-block streams, results gathered into the root's arrays) and K-step rollouts with one gather per K steps."""
    import torch
    import space_gym_amd as sg
    n, K = 5000, 12  # ragged blocks: 1667 + 1667 + 1666
    one = sg.make_vec("GoalContinuous3P-v0", n, device=0, seed=9, max_episode_steps=30, terminal_observation=False)
    multi = sg.make_vec("GoalContinuous3P-v0", n, devices=[0, 0, 0], seed=9, max_episode_steps=30, copy=False)
    assert [hi - lo for lo, hi in multi.bounds] == [1667, 1667, 1666] and len(multi.shards) == 3
    o1 = one.reset_torch().clone(); om = multi.reset_torch()
    assert torch.equal(o1, om)
    g = torch.Generator(device="cuda").manual_seed(4)
    seen = set()
    for t in range(40):
        a = torch.rand((n, 2), device="cuda", generator=g) * 2 - 1
        r1 = [x.clone() for x in one.step_torch(a)]
        rm = multi.step_torch(a)
        seen.add(rm[0].data_ptr())
        for x, y in zip(r1, rm):
            assert torch.equal(x, y)
    assert len(seen) == 2  # copy=False: two result sets that alternate
    a = torch.rand((K, n, 2), device="cuda", generator=g) * 2 - 1
    o = torch.empty((K, n, one.obs_dim), device="cuda"); r = torch.empty((K, n), device="cuda")
    d = torch.empty((K, n), dtype=torch.uint8, device="cuda"); tr = torch.empty_like(d)
    one.rollout_torch(a, o, r, d, tr)
    mo, mr, md, mt = multi.rollout_torch(a)
    assert torch.equal(o, mo) and torch.equal(r, mr) and torch.equal(d, md) and torch.equal(tr, mt)
    assert int(d.sum()) > 100  # episodes ended and restarted
    s1, sm = one.get_state(), multi.get_state()
    for k in s1:
        assert np.array_equal(s1[k], sm[k])
    # NumPy convenience path
    obs, rew, done, info = multi.step(a[0].cpu().numpy())
    o1, r1, d1, i1 = one.step(a[0].cpu().numpy())
    assert np.array_equal(obs, o1) and np.array_equal(rew, r1) and np.array_equal(done, d1)
    multi.check_status(); one.close(); multi.close()
    # a discrete id with constructor kwargs through the same front end
    m2 = sg.make_vec("GoalDiscrete3-v0", 777, devices=[0, 0], seed=2, max_engine_force=0.8)
    o2 = sg.make_vec("GoalDiscrete3-v0", 777, device=0, seed=2, max_engine_force=0.8, terminal_observation=False)
    assert torch.equal(m2.reset_torch(), o2.reset_torch())
    ai = torch.randint(0, 6, (777,), device="cuda", dtype=torch.int32, generator=g)
    for x, y in zip(m2.step_torch(ai), o2.step_torch(ai)):
        assert torch.equal(x, y)
    m2.close(); o2.close()


def test_step_end_pointers_stay_valid_until_the_begin_after_next():
    """include/spacegym.h: the result block of step t is written again by the kernel enqueued by sg_step_begin of step t + 2
    (not before): its contents are intact after step t + 1 has completed."""
    import space_gym_amd as sg
    n = 4096
    env = sg.make_vec("GoalContinuous2P-v0", n, device=0, seed=1, copy=False)
    env.reset()
    rng = np.random.default_rng(0)
    a = [rng.uniform(-1, 1, (n, 2)).astype(np.float32) for _ in range(3)]
    env.step_async(a[0]); obs0, rew0, done0, _ = env.step_wait()
    keep = (obs0.copy(), rew0.copy(), done0.copy())
    env.step_async(a[1]); obs1, rew1, done1, _ = env.step_wait()
    assert obs1.ctypes.data != obs0.ctypes.data  # the other block
    assert np.array_equal(obs0, keep[0]) and np.array_equal(rew0, keep[1]) and np.array_equal(done0, keep[2])  # step t intact after step t + 1
    env.step_async(a[2]); obs2, _, _, _ = env.step_wait()
    assert obs2.ctypes.data == obs0.ctypes.data and not np.array_equal(obs2, keep[0])  # written again by step t + 2
    env.close()
