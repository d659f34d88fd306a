#!/usr/bin/env python3
"""Generate golden input/output vectors from the UNMODIFIED reference at /root/reference.

Runs only in the build container (the reference never travels).  Commits DATA only
(tests/golden/*.npz): inputs quantised to fp32-representable values and the fp64 outputs
the reference produced for them.  No reference source text is stored.

Two fixture families, generated in two separate processes:

  env   (this process, `--stage env`)   the reference env layer (SpaceshipEnv.step:
        spaceship_env.py:68-78 -> dynamic_model.py:94-125 -> scipy RK45 + events ->
        _make_observation :113-131 -> GoalEnv._reward goal.py:147-158 /
        KeplerEnv._reward kepler.py:152-156).  `gym` is absent in this image, so the env
        modules are imported with the loader-only namespace in tools/_gym_loader_shim.py.
  core  (child process, `--stage core`)  `dynamic_model.make_step` alone on the SAME inputs,
        imported with NO shim at all (only numpy + scipy), package `__init__` bypassed by
        pre-seeding sys.modules["gym_space"].  The generator asserts core == env bitwise
        for (state', done), i.e. the integrator/event fixtures do not depend on the shim.

Usage:  python tools/gen_golden.py            # writes tests/golden/*.npz
"""
import argparse
import contextlib
import io
import os
import subprocess
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

FAMILIES = {
    "goal2p": "GoalContinuous2P-v0",
    "goal3p": "GoalContinuous3P-v0",
    "goal4p": "GoalContinuous4P-v0",
    "kepler_circle": "KeplerCircleOrbit-v0",
    "kepler_easy": "KeplerEllipseEasy-v0",
    "kepler_hard": "KeplerEllipseHard-v0",
}
# Discrete-action ids: registered only by keyboard_agent.py:10-74 (not by gym_space/__init__.py); same classes, kwargs below.
DISCRETE_FAMILIES = {
    "goal_discrete2": ("GoalDiscrete2-v0", "GoalDiscreteEnv", dict(n_planets=2, ship_steering=1, ship_moi=0.01, survival_reward_scale=0.2,
                                                                   goal_vel_reward_scale=5.0, safety_reward_scale=10.0,
                                                                   goal_sparse_reward=5.0, max_engine_force=1)),
    "goal_discrete3": ("GoalDiscrete3-v0", "GoalDiscreteEnv", dict(n_planets=3, ship_steering=1, ship_moi=0.01, survival_reward_scale=0.2,
                                                                   goal_vel_reward_scale=5.0, safety_reward_scale=10.0,
                                                                   goal_sparse_reward=5.0, max_engine_force=1)),
    "goal_discrete4": ("GoalDiscrete4-v0", "GoalDiscreteEnv", dict(n_planets=4, ship_steering=1, ship_moi=0.01, survival_reward_scale=0.2,
                                                                   goal_vel_reward_scale=5.0, safety_reward_scale=10.0,
                                                                   goal_sparse_reward=5.0, max_engine_force=1)),
    "kepler_discrete": ("KeplerDiscrete-v0", "KeplerDiscreteEnv", dict(ship_steering=1, ship_moi=0.01, max_engine_force=0.4, reward_value=0,
                                                                       rad_penalty_C=2, numerator_C=0.01, act_penalty_C=0.5, step_size=0.07,
                                                                       randomize=False, ref_orbit_a=1.2, ref_orbit_eccentricity=0,
                                                                       ref_orbit_angle=0)),
}
STEP_SIZE = 0.07
MAX_EPISODE_STEPS = 500
# thrust 0.4 over a 3..6 unit world bounds natural speeds at ~1.6..2.2; forced cases stay below this
MAX_FORCED_SPEED = 2.5


def q32(x):
    """Quantise to fp32-representable float64."""
    return np.asarray(x, dtype=np.float32).astype(np.float64)


# --------------------------------------------------------------------------- loaders
def load_core_only():
    """dynamic_model/helpers/planet/ship_params with numpy+scipy only (no gym, no shim)."""
    pkg = types.ModuleType("gym_space")
    pkg.__path__ = [os.path.join(REF, "gym_space")]
    sys.modules["gym_space"] = pkg
    from gym_space import dynamic_model, planet, ship_params  # noqa: F401
    return dynamic_model, planet, ship_params


def load_env_layer():
    sys.path.insert(0, HERE)
    import _gym_loader_shim
    _gym_loader_shim.install()
    sys.path.insert(0, REF)
    import gym_space  # noqa: F401  (runs the register() calls -> REGISTRY)
    import gym_space.envs as envs
    from gym_space import dynamic_model
    return _gym_loader_shim.REGISTRY, envs, dynamic_model


def make_env(registry, envs, env_id):
    spec = registry[env_id]
    cls = getattr(envs, spec["entry_point"].split(":")[1])
    with contextlib.redirect_stdout(io.StringIO()):  # constructor prints its config
        env = cls(**spec["kwargs"])
    assert spec["max_episode_steps"] == MAX_EPISODE_STEPS
    return env


# --------------------------------------------------------------------------- env helpers
class IvpRecorder:
    """Wraps scipy's solve_ivp at the reference's single call site to read back diagnostics
    (number of RK steps, nfev, which event fired and when).  Arithmetic is untouched."""

    def __init__(self, dynamic_model):
        self.real = dynamic_model.solve_ivp
        self.last = None
        dynamic_model.solve_ivp = self

    def __call__(self, *a, **k):
        self.last = self.real(*a, **k)
        return self.last

    def diag(self):
        r = self.last
        ev_idx, t_ev = -1, np.nan
        if r.status == 1:
            cands = [(te[-1], i) for i, te in enumerate(r.t_events) if len(te)]
            t_ev, ev_idx = max(cands)  # terminal root is the last one recorded
        return len(r.t) - 1, r.nfev, ev_idx, t_ev


def is_goal(env):
    return bool(env.with_goal)


def inject(env, state, planets_xy=None, goal=None):
    env._ship_state._state_vec = np.array(state, dtype=np.float64)
    if planets_xy is not None:
        for p, xy in zip(env.planets, planets_xy):
            p.center_pos = np.array(xy, dtype=np.float64)
    if goal is not None:
        env.goal_pos = np.array(goal, dtype=np.float64)


def quantise_env(env):
    inject(env, q32(env._ship_state._state_vec),
           [q32(p.center_pos) for p in env.planets] if is_goal(env) else None,
           q32(env.goal_pos) if is_goal(env) else None)


def snapshot_inputs(env):
    n = len(env.planets)
    d = dict(state0=env._ship_state._state_vec.copy())
    if is_goal(env):
        d["planets"] = np.array([p.center_pos for p in env.planets]).reshape(n, 2)
        d["goal"] = env.goal_pos.copy()
    return d


def is_discrete(env):
    return hasattr(env.action_space, "n")


def step_and_record(env, rec, action, kind):
    row = snapshot_inputs(env)
    goal_before = env.goal_pos.copy() if is_goal(env) else None
    if is_discrete(env):
        row["action"] = np.int32(action)
        obs, reward, done, _ = env.step(int(action))
    else:
        row["action"] = np.asarray(action, dtype=np.float32)
        obs, reward, done, _ = env.step(row["action"].copy())
    row["state1"] = np.array(env._ship_state._state_vec, dtype=np.float64)
    row["obs"] = np.array(obs, dtype=np.float64)
    row["reward"] = float(reward)
    row["done"] = bool(done)
    row["goal_changed"] = bool(is_goal(env) and not np.array_equal(goal_before, env.goal_pos))
    n_rk, nfev, ev_idx, t_ev = rec.diag()
    row["n_rk_steps"], row["nfev"], row["event_index"], row["t_event"] = n_rk, nfev, ev_idx, t_ev
    row["kind"] = kind
    return row, done


KINDS = ["rollout", "wall", "corner", "planet", "goal_hit", "danger", "border", "near_orbit", "extreme_action"]


def rollout_rows(env, rec, rng, n_steps, n_keep_nonterminal):
    rows_t, rows_n = [], []
    env.seed(int(rng.randint(1 << 30)))
    env.reset()
    quantise_env(env)
    elapsed = 0
    for _ in range(n_steps):
        a = rng.randint(6) if is_discrete(env) else rng.uniform(-1, 1, size=2).astype(np.float32)
        row, done = step_and_record(env, rec, a, KINDS.index("rollout"))
        elapsed += 1
        (rows_t if (done or row["goal_changed"]) else rows_n).append(row)
        if done or elapsed >= MAX_EPISODE_STEPS:
            env.reset()
            elapsed = 0
        quantise_env(env)
    keep = rng.choice(len(rows_n), size=min(n_keep_nonterminal, len(rows_n)), replace=False)
    return rows_t + [rows_n[i] for i in sorted(keep)]


def rand_action(rng, extreme=False):
    if extreme:
        return np.array(rng.choice([-1.0, 0.0, 1.0], size=2), dtype=np.float32)
    return rng.uniform(-1, 1, size=2).astype(np.float32)


def unit(angle):
    return np.array([np.cos(angle), np.sin(angle)])


def forced_goal_rows(env, rec, rng, n_each):
    rows = []
    R = env.planets[0].radius
    half = env.world_size / 2
    h = env.step_size

    def fresh_layout():
        env.reset()
        quantise_env(env)
        return [p.center_pos.copy() for p in env.planets], env.goal_pos.copy()

    def clear_of_planets(xy, planets, margin):
        return all(np.linalg.norm(xy - p) > R + margin for p in planets)

    def ship_state(xy, vel):
        return q32([xy[0], xy[1], rng.uniform(0, 2 * np.pi), vel[0], vel[1], rng.normal() * 1.4])

    def run(state, planets, goal, kind, extreme=False):
        inject(env, state, planets, q32(goal))
        a = rng.randint(6) if is_discrete(env) else rand_action(rng, extreme)
        row, _ = step_and_record(env, rec, a, KINDS.index(kind))
        rows.append(row)

    for _ in range(n_each):  # walls: +x, -x, +y, -y
        planets, goal = fresh_layout()
        axis, sign = rng.randint(2), rng.choice([-1.0, 1.0])
        for _try in range(100):
            xy = rng.uniform(-half + 0.1, half - 0.1, size=2)
            xy[axis] = sign * (half - rng.uniform(0.0, 0.06))
            if clear_of_planets(xy, planets, 0.03):
                break
        vel = rng.normal(size=2) * 0.4
        vel[axis] = sign * rng.uniform(0.05, 1.5)
        run(ship_state(xy, vel), planets, goal, "wall")
    for _ in range(n_each // 2):  # corners
        planets, goal = fresh_layout()
        sx, sy = rng.choice([-1.0, 1.0], size=2)
        xy = np.array([sx * (half - rng.uniform(0, 0.05)), sy * (half - rng.uniform(0, 0.05))])
        if not clear_of_planets(xy, planets, 0.03):
            continue
        vel = np.array([sx, sy]) * rng.uniform(0.1, 1.5, size=2)
        run(ship_state(xy, vel), planets, goal, "corner")
    for _ in range(2 * n_each):  # planets: head-on, oblique, grazing
        planets, goal = fresh_layout()
        j = rng.randint(len(planets))
        ang = rng.uniform(0, 2 * np.pi)
        xy = planets[j] + unit(ang) * (R + rng.uniform(0.0005, 0.07))
        if np.any(np.abs(xy) > half - 0.01) or not clear_of_planets(xy, planets, 0.0):
            continue
        vel = -unit(ang + np.deg2rad(rng.uniform(-85, 85))) * rng.uniform(0.05, MAX_FORCED_SPEED)
        run(ship_state(xy, vel), planets, goal, "planet")
    for _ in range(n_each):  # goal hits / near misses
        planets, _goal = fresh_layout()
        for _try in range(100):
            xy = rng.uniform(-half + 0.2, half - 0.2, size=2)
            if clear_of_planets(xy, planets, 0.05):
                break
        vel = rng.normal(size=2) * 0.4
        goal = xy + vel * h + unit(rng.uniform(0, 2 * np.pi)) * env.goal_radius * rng.uniform(0.0, 1.4)
        run(ship_state(xy, vel), planets, goal, "goal_hit")
    for _ in range(n_each):  # danger zone approach / retreat
        planets, goal = fresh_layout()
        j = rng.randint(len(planets))
        ang = rng.uniform(0, 2 * np.pi)
        xy = planets[j] + unit(ang) * (R + rng.uniform(0.08, 0.3))
        if np.any(np.abs(xy) > half - 0.05) or not clear_of_planets(xy, planets, 0.0):
            continue
        vel = unit(ang) * rng.uniform(-0.8, 0.8) + unit(ang + np.pi / 2) * rng.normal() * 0.3
        run(ship_state(xy, vel), planets, goal, "danger")
    for _ in range(n_each // 2):  # extreme actions from ordinary states
        planets, goal = fresh_layout()
        run(q32(env._ship_state._state_vec), planets, goal, "extreme_action", extreme=True)
    return rows


def forced_kepler_rows(env, rec, rng, n_each):
    rows = []
    Rb, Rp = env._border_radius, env._planet_radius

    def ship_state(xy, vel):
        return q32([xy[0], xy[1], rng.uniform(0, 2 * np.pi), vel[0], vel[1], rng.normal() * 0.84])

    def run(state, kind, action=None):
        inject(env, state)
        if is_discrete(env):
            a = rng.randint(6) if (action is None or kind != "near_orbit") else 0  # 0 = engine off, no thruster
        else:
            a = rand_action(rng) if action is None else np.asarray(action, dtype=np.float32)
        row, _ = step_and_record(env, rec, a, KINDS.index(kind))
        rows.append(row)

    for _ in range(2 * n_each):  # border circle, crossed from inside
        ang = rng.uniform(0, 2 * np.pi)
        xy = unit(ang) * (Rb - rng.uniform(0.0005, 0.07))
        vel = unit(ang + np.deg2rad(rng.uniform(-85, 85))) * rng.uniform(0.05, MAX_FORCED_SPEED)
        run(ship_state(xy, vel), "border")
    for _ in range(2 * n_each):  # central planet
        ang = rng.uniform(0, 2 * np.pi)
        xy = unit(ang) * (Rp + rng.uniform(0.0005, 0.07))
        vel = -unit(ang + np.deg2rad(rng.uniform(-85, 85))) * rng.uniform(0.05, MAX_FORCED_SPEED)
        run(ship_state(xy, vel), "planet")
    for _ in range(2 * n_each):  # on / near the reference orbit, with target velocity, weak actions
        nu = rng.uniform(0, 2 * np.pi)
        a_, e_, phi = env.ref_orbit_a, env.ref_orbit_eccentricity, env.ref_orbit_angle
        b_ = a_ * np.sqrt(1 - e_ * e_)
        c_ = np.sqrt(a_ * a_ - b_ * b_)
        w = np.array([a_ * np.cos(nu) + c_, b_ * np.sin(nu)])  # ellipse in the rotated frame
        rot_back = np.array([[np.cos(phi), -np.sin(phi)], [np.sin(phi), np.cos(phi)]])
        xy = rot_back @ w + rng.normal(size=2) * rng.choice([0.0, 1e-3, 3e-2])
        if not (Rp + 0.02 < np.linalg.norm(xy) < Rb - 0.02):
            continue
        vt = env._orbit_target_vel(xy.copy(), phi, a_, e_) + rng.normal(size=2) * rng.choice([0.0, 1e-3, 3e-2])
        action = [-1.0, 0.0] if rng.uniform() < 0.5 else rng.uniform(-1, 1, size=2) * 0.1 + [-0.9, 0.0]
        run(ship_state(xy, vt), "near_orbit", action=np.clip(action, -1, 1))
    for _ in range(n_each // 2):
        env.reset()
        run(q32(env._ship_state._state_vec), "extreme_action", action=rand_action(rng, extreme=True))
    return rows


def rows_to_arrays(rows, goal):
    out = {}
    keys = ["state0", "action", "state1", "obs", "reward", "done", "goal_changed",
            "n_rk_steps", "nfev", "event_index", "t_event", "kind"]
    if goal:
        keys += ["planets", "goal"]
    for k in keys:
        out[k] = np.array([r[k] for r in rows])
    out["done"] = out["done"].astype(np.uint8)
    out["goal_changed"] = out["goal_changed"].astype(np.uint8)
    for k in ("n_rk_steps", "nfev", "event_index", "kind"):
        out[k] = out[k].astype(np.int16)
    return out


def env_constants(env):
    c = dict(step_size=env.step_size, world_size=env.world_size, max_abs_vel_angle=env.max_abs_vel_angle,
             n_planets=len(env.planets), planet_mass=[p.mass for p in env.planets],
             planet_radius=[p.radius for p in env.planets],
             max_engine_force=env.ship_params.max_engine_force, ship_mass=env.ship_params.mass)
    if is_goal(env):
        t = env._hexagonal_tiling
        c.update(goal_radius=env.goal_radius, danger_zone=env.danger_zone,
                 survival_reward_scale=env.survival_reward_scale, goal_vel_reward_scale=env.goal_vel_reward_scale,
                 safety_reward_scale=env.safety_reward_scale, goal_sparse_reward=env.goal_sparse_reward,
                 distance_fctr=env._distance_fctr, tiling_rows=t._rows, tiling_cols=t._cols, tiling_a=t._a,
                 ship_radius=t.ship_radius)
    else:
        c.update(ref_orbit_a=env.ref_orbit_a, ref_orbit_eccentricity=env.ref_orbit_eccentricity,
                 ref_orbit_angle=env.ref_orbit_angle, numerator_C=env.numerator_C,
                 rad_penalty_C=env.rad_penalty_C, act_penalty_C=env.act_penalty_C)
    return {"const_" + k: np.asarray(v, dtype=np.float64) for k, v in c.items()}


# --------------------------------------------------------------------------- reset statistics
def hist2d(xy, half, bins):
    h, _, _ = np.histogram2d(xy[:, 0], xy[:, 1], bins=bins, range=[[-half, half], [-half, half]])
    return h.astype(np.int64)


def goal_reset_stats(env, n_resets, n_hits, seed):
    """Distributional fixtures for GoalEnv._reset (goal.py:133-145), HexagonalTiling.reset
    (hexagonal_tiling.py:53-93) and the goal-resample chain (hexagonal_tiling.py:95-128)."""
    t = env._hexagonal_tiling
    nt, half, N = t._n_tiles, env.world_size / 2, len(env.planets)
    env.seed(seed)
    ship = np.empty((n_resets, 2)); planets = np.empty((n_resets, N, 2)); kin = np.empty((n_resets, 4))
    goals = np.empty((n_resets, n_hits + 1, 2))
    ship_tile = np.zeros(nt, np.int64); flags = np.zeros(2, np.int64)
    goal_tile = np.zeros((n_hits + 1, nt), np.int64); same = np.zeros(n_hits + 1, np.int64)
    taxi = np.zeros((n_hits + 1, t._rows + t._cols), np.int64)
    free_len = np.zeros((n_hits + 1, nt + n_hits + 2), np.int64)
    col_shift_sum = np.zeros(t._cols)
    planet_tiles_pair = np.zeros((nt, nt), np.int64)  # (ship tile, tile of nearest-centre planet 0)
    min_clear = dict(ship_planet=np.inf, ship_wall=np.inf, planet_wall=np.inf, goal_planet=np.inf,
                     planet_planet=np.inf, goal_wall=np.inf)
    R, rs, rg = env.planets[0].radius, t.ship_radius, env.goal_radius
    for i in range(n_resets):
        env.reset()
        sv = env._ship_state._state_vec
        ship[i] = sv[:2]; kin[i] = sv[2:6]
        planets[i] = [p.center_pos for p in env.planets]
        flags += [int(t._case_b), int(t._flip_xy)]
        col_shift_sum += t._col_shift
        ship_tile[t._ship_tile_nr] += 1
        for k in range(n_hits + 1):
            if k > 0:
                env._resample_goal()  # what GoalEnv._reward does on a hit (goal.py:154-157)
            goals[i, k] = env.goal_pos
            goal_tile[k, t._goal_tile_nr] += 1
            same[k] += int(t._goal_tile_nr == t._ship_tile_nr)
            (r0, c0), (r1, c1) = t._tiles_coord[t._ship_tile_nr], t._tiles_coord[t._goal_tile_nr]
            taxi[k, abs(r0 - r1) + abs(c0 - c1)] += 1
            free_len[k, len(t._free_tiles_nrs)] += 1
            if k == 0:
                d = np.linalg.norm(planets[i] - ship[i], axis=1).min()
                min_clear["ship_planet"] = min(min_clear["ship_planet"], d - R)
                min_clear["ship_wall"] = min(min_clear["ship_wall"], half - np.abs(ship[i]).max())
                min_clear["planet_wall"] = min(min_clear["planet_wall"], half - np.abs(planets[i]).max() - R)
                pd = min(np.linalg.norm(planets[i][a] - planets[i][b]) for a in range(N) for b in range(a))
                min_clear["planet_planet"] = min(min_clear["planet_planet"], pd - 2 * R)
            d = np.linalg.norm(planets[i] - goals[i, k], axis=1).min()
            min_clear["goal_planet"] = min(min_clear["goal_planet"], d - R)
            min_clear["goal_wall"] = min(min_clear["goal_wall"], half - np.abs(goals[i, k]).max())
    out = dict(
        n_resets=n_resets, n_hits=n_hits, seed=seed,
        ship_hist=hist2d(ship, half, 24), planets_hist=hist2d(planets.reshape(-1, 2), half, 24),
        goal_hist=np.stack([hist2d(goals[:, k], half, 24) for k in range(n_hits + 1)]),
        ship_mean=ship.mean(0), ship_cov=np.cov(ship.T), planets_mean=planets.reshape(-1, 2).mean(0),
        planets_cov=np.cov(planets.reshape(-1, 2).T),
        goal_mean=goals.mean(0), goal_sq_mean=(goals ** 2).mean(0),
        ship_goal_dist_mean=np.linalg.norm(goals - ship[:, None], axis=2).mean(0),
        ship_goal_dist_hist=np.stack([np.histogram(np.linalg.norm(goals[:, k] - ship, axis=1), bins=30,
                                                   range=(0, 4.5))[0] for k in range(n_hits + 1)]),
        theta_hist=np.histogram(kin[:, 0], bins=16, range=(0, 2 * np.pi))[0],
        vel_hist=np.histogram(kin[:, 1:3].ravel(), bins=32, range=(-0.35, 0.35))[0],
        vel_mean=kin[:, 1:3].mean(), vel_std=kin[:, 1:3].std(),
        omega_hist=np.histogram(kin[:, 3], bins=32, range=(-4.2 - 1e-9, 4.2 + 1e-9))[0],
        omega_mean=kin[:, 3].mean(), omega_std=kin[:, 3].std(),
        omega_clipped=np.sum(np.abs(kin[:, 3]) >= 4.2),
        ship_tile=ship_tile, goal_tile=goal_tile, same_tile=same, taxi=taxi, free_len=free_len,
        case_b_flip=flags, col_shift_mean=col_shift_sum / n_resets,
        min_clear=np.array([min_clear[k] for k in sorted(min_clear)]),
        min_clear_keys=np.array(sorted(min_clear)),
        planet_radius=R, ship_radius=rs, goal_radius=rg,
    )
    return {k: np.asarray(v) for k, v in out.items()}


def kepler_reset_stats(env, n_resets, seed):
    """KeplerEnv._reset (kepler.py:233-267)."""
    env.seed(seed)
    sv = np.empty((n_resets, 6))
    for i in range(n_resets):
        env.reset()
        sv[i] = env._ship_state._state_vec
    rad = np.linalg.norm(sv[:, :2], axis=1)
    ang = np.arctan2(sv[:, 1], sv[:, 0]) % (2 * np.pi)
    out = dict(
        n_resets=n_resets, seed=seed,
        radius_hist=np.histogram(rad, bins=18, range=(0.7, 2.5))[0], radius_min=rad.min(), radius_max=rad.max(),
        angle_hist=np.histogram(ang, bins=16, range=(0, 2 * np.pi))[0],
        theta_hist=np.histogram(sv[:, 2], bins=16, range=(0, 2 * np.pi))[0],
        vel_hist=np.histogram(sv[:, 3:5].ravel(), bins=32, range=(-0.25, 0.25))[0],
        vel_mean=sv[:, 3:5].mean(), vel_std=sv[:, 3:5].std(),
        omega_hist=np.histogram(sv[:, 5], bins=32, range=(-4.2 - 1e-9, 4.2 + 1e-9))[0],
        omega_mean=sv[:, 5].mean(), omega_std=sv[:, 5].std(),
    )
    return {k: np.asarray(v) for k, v in out.items()}


# --------------------------------------------------------------------------- stages
def stage_env(args):
    registry, envs, dynamic_model = load_env_layer()
    rec = IvpRecorder(dynamic_model)
    os.makedirs(OUT, exist_ok=True)
    for fam, env_id in FAMILIES.items():
        rng = np.random.RandomState(sum(map(ord, fam)) * 7919 % (1 << 31))
        env = make_env(registry, envs, env_id)
        rows = rollout_rows(env, rec, rng, args.rollout_steps, args.keep_nonterminal)
        rows += (forced_goal_rows if is_goal(env) else forced_kepler_rows)(env, rec, rng, args.forced_each)
        arrs = rows_to_arrays(rows, is_goal(env))
        arrs.update(env_constants(env))
        arrs["kind_names"] = np.array(KINDS)
        arrs["env_id"] = np.array(env_id)
        path = os.path.join(OUT, f"step_{fam}.npz")
        np.savez_compressed(path, **arrs)
        print(f"{fam}: {len(rows)} transitions, {int(arrs['done'].sum())} terminal, "
              f"{int(arrs['goal_changed'].sum())} goal hits -> {path}", flush=True)


def stage_discrete(args):
    """Discrete-action variants (DiscreteSpaceshipEnv._translate_raw_action, spaceship_env.py:189-202)."""
    registry, envs, dynamic_model = load_env_layer()
    rec = IvpRecorder(dynamic_model)
    for fam, (env_id, cls_name, kwargs) in DISCRETE_FAMILIES.items():
        rng = np.random.RandomState(sum(map(ord, fam)) * 7919 % (1 << 31))
        with contextlib.redirect_stdout(io.StringIO()):
            env = getattr(envs, cls_name)(**kwargs)
        rows = rollout_rows(env, rec, rng, args.rollout_steps // 2, args.keep_nonterminal // 2)
        rows += (forced_goal_rows if is_goal(env) else forced_kepler_rows)(env, rec, rng, args.forced_each // 2)
        arrs = rows_to_arrays(rows, is_goal(env))
        arrs.update(env_constants(env))
        arrs["kind_names"] = np.array(KINDS)
        arrs["env_id"] = np.array(env_id)
        path = os.path.join(OUT, f"step_{fam}.npz")
        np.savez_compressed(path, **arrs)
        print(f"{fam}: {len(rows)} transitions, {int(arrs['done'].sum())} terminal, "
              f"{int(arrs['goal_changed'].sum())} goal hits -> {path}", flush=True)


def stage_random_orbits(args):
    """KeplerRandomOrbits-v0 (randomize=True, kepler.py:257-259): every row carries its own (angle, eccentricity)."""
    registry, envs, dynamic_model = load_env_layer()
    rec = IvpRecorder(dynamic_model)
    env = make_env(registry, envs, "KeplerRandomOrbits-v0")
    rng = np.random.RandomState(4242)
    np.random.seed(99)  # the reference draws the orbit from the GLOBAL numpy RNG
    rows = []
    env.seed(7); env.reset()
    elapsed = 0
    for _ in range(args.rollout_steps // 2):
        # fp32-representable orbit parameters, like every other engine input
        env.ref_orbit_angle = float(np.float32(env.ref_orbit_angle)); env.ref_orbit_eccentricity = float(np.float32(env.ref_orbit_eccentricity))
        inject(env, q32(env._ship_state._state_vec))
        a = rng.uniform(-1, 1, size=2).astype(np.float32)
        if rng.uniform() < 0.2:
            a = np.array([-1.0, 0.0], np.float32)  # engine off: the sensitive regime of the reward
        orbit = np.array([env.ref_orbit_angle, env.ref_orbit_eccentricity, env.ref_orbit_a])
        row, done = step_and_record(env, rec, a, KINDS.index("rollout"))
        row["orbit"] = orbit
        rows.append(row)
        elapsed += 1
        if done or elapsed >= MAX_EPISODE_STEPS:
            env.reset(); elapsed = 0
    keep_t = [r for r in rows if r["done"]]
    keep_n = [r for r in rows if not r["done"]]
    sel = rng.choice(len(keep_n), size=min(args.keep_nonterminal, len(keep_n)), replace=False)
    rows = keep_t + [keep_n[i] for i in sorted(sel)]
    arrs = rows_to_arrays(rows, False)
    arrs["orbit"] = np.array([r["orbit"] for r in rows])
    arrs.update(env_constants(env))
    arrs["kind_names"] = np.array(KINDS); arrs["env_id"] = np.array("KeplerRandomOrbits-v0")
    path = os.path.join(OUT, "step_kepler_random.npz")
    np.savez_compressed(path, **arrs)
    print(f"kepler_random: {len(rows)} transitions, {int(arrs['done'].sum())} terminal, ecc range "
          f"[{arrs['orbit'][:, 1].min():.3f}, {arrs['orbit'][:, 1].max():.3f}] -> {path}", flush=True)


def stage_acceleration(args):
    """Steering.acceleration (ship_steering=0, the constructor default of GoalEnv / KeplerEnv: goal.py:27, kepler.py:198):
    omega is a state variable (dynamic_model.py:138-141 does not overwrite it), the thruster gives an angular acceleration
    (:160-161) and the angular-velocity event (:210-212) is live.  No registered id uses it; fixtures use the kwargs of the
    3-planet / circle-orbit ids with ship_steering=0."""
    registry, envs, dynamic_model = load_env_layer()
    rec = IvpRecorder(dynamic_model)
    specs = {"goal3p_accel": ("GoalContinuousEnv", dict(registry["GoalContinuous3P-v0"]["kwargs"], ship_steering=0)),
             "kepler_circle_accel": ("KeplerContinuousEnv", dict(registry["KeplerCircleOrbit-v0"]["kwargs"], ship_steering=0))}
    for fam, (cls_name, kwargs) in specs.items():
        rng = np.random.RandomState(sum(map(ord, fam)) * 7919 % (1 << 31))
        with contextlib.redirect_stdout(io.StringIO()):
            env = getattr(envs, cls_name)(**kwargs)
        assert env.ship_params.steering.value == 0
        rows = rollout_rows(env, rec, rng, args.rollout_steps // 2, args.keep_nonterminal // 2)
        rows += (forced_goal_rows if is_goal(env) else forced_kepler_rows)(env, rec, rng, args.forced_each // 2)
        # angular-velocity event: |omega| close to the limit 6 with the thruster pushing either way
        for _ in range(args.forced_each * 2):
            env.reset(); quantise_env(env)
            sv = env._ship_state._state_vec.copy()
            sv[5] = rng.choice([-1.0, 1.0]) * (6.0 - rng.uniform(0.0, 0.5))
            inject(env, q32(sv))
            a = rng.uniform(-1, 1, size=2).astype(np.float32)
            row, _ = step_and_record(env, rec, a, KINDS.index("extreme_action"))
            rows.append(row)
        arrs = rows_to_arrays(rows, is_goal(env))
        arrs.update(env_constants(env))
        arrs["const_moi"] = np.asarray(env.ship_params.moi); arrs["const_max_thruster_force"] = np.asarray(env.ship_params.max_thruster_force)
        arrs["kind_names"] = np.array(KINDS)
        path = os.path.join(OUT, f"step_{fam}.npz")
        np.savez_compressed(path, **arrs)
        n_ev = int(arrs["const_n_planets"]) + 2
        print(f"{fam}: {len(rows)} transitions, {int(arrs['done'].sum())} terminal "
              f"({int((arrs['event_index'] == n_ev).sum())} by the angular-velocity event) -> {path}", flush=True)


# Constructor kwargs beyond the registered ones (GoalEnv.__init__ goal.py:18-31, KeplerEnv.__init__ kepler.py:189-203): what
# sg_params / make_vec(env_id, **kwargs) must reproduce.  (class, base id for the engine, kwargs)
KWARGS_SETS = {
    "goal_a": ("GoalContinuousEnv", "GoalContinuous3P-v0",
               dict(n_planets=3, ship_steering=1, ship_moi=0.01, max_engine_force=0.7, goal_vel_reward_scale=3.0,
                    safety_reward_scale=4.0, goal_sparse_reward=2.5, survival_reward_scale=0.05, danger_zone=0.4)),
    "goal_b": ("GoalContinuousEnv", "GoalContinuous2P-v0",
               dict(n_planets=2, ship_steering=1, ship_moi=0.01, max_engine_force=0.25, goal_vel_reward_scale=8.0,
                    safety_reward_scale=20.0, goal_sparse_reward=10.0, survival_reward_scale=0.0, danger_zone=0.1)),
    "goal_c": ("GoalContinuousEnv", "GoalContinuous2P-v0",  # another planet count than the base id's, Steering.acceleration, another moi
               dict(n_planets=4, ship_steering=0, ship_moi=0.02, max_engine_force=0.55, goal_vel_reward_scale=5.0,
                    safety_reward_scale=10.0, goal_sparse_reward=5.0, survival_reward_scale=0.2)),
    "kepler_a": ("KeplerContinuousEnv", "KeplerCircleOrbit-v0",  # the constructor's own step size
                 dict(ref_orbit_a=1.5, ref_orbit_eccentricity=0.3, ref_orbit_angle=2.0, step_size=0.1, ship_steering=1,
                      numerator_C=0.02, rad_penalty_C=1.0, act_penalty_C=0.25, max_engine_force=0.4)),
    "kepler_b": ("KeplerContinuousEnv", "KeplerEllipseEasy-v0",
                 dict(ref_orbit_a=0.9, ref_orbit_eccentricity=0.6, ref_orbit_angle=5.0, step_size=0.07, ship_steering=1,
                      numerator_C=0.05, rad_penalty_C=3.0, act_penalty_C=1.0, max_engine_force=0.7)),
    "kepler_c": ("KeplerContinuousEnv", "KeplerCircleOrbit-v0",  # every class default (Steering.acceleration, step 0.1, a 1.2, e 0.5, angle 3.75) but the moi
                 dict(ship_moi=0.02)),
    "kepler_d": ("KeplerContinuousEnv", "KeplerEllipseHard-v0",  # a shorter env-step than the registered one
                 dict(step_size=0.05, ship_steering=1, ref_orbit_a=1.1, ref_orbit_eccentricity=0.2, ref_orbit_angle=1.0)),
}


def stage_kwargs(args):
    """Non-registered parameter sets, constructed like gym.make(id, **kwargs) / the classes themselves would."""
    import json
    registry, envs, dynamic_model = load_env_layer()
    rec = IvpRecorder(dynamic_model)
    for name, (cls_name, base_id, kwargs) in KWARGS_SETS.items():
        rng = np.random.RandomState(sum(map(ord, name)) * 7919 % (1 << 31))
        with contextlib.redirect_stdout(io.StringIO()):
            env = getattr(envs, cls_name)(**kwargs)
        rows = rollout_rows(env, rec, rng, args.rollout_steps // 3, args.keep_nonterminal // 3)
        rows += (forced_goal_rows if is_goal(env) else forced_kepler_rows)(env, rec, rng, args.forced_each // 3)
        if env.ship_params.steering.value == 0:  # angular-velocity event: |omega| close to the limit
            for _ in range(args.forced_each):
                env.reset(); quantise_env(env)
                sv = env._ship_state._state_vec.copy()
                sv[5] = rng.choice([-1.0, 1.0]) * (6.0 - rng.uniform(0.0, 0.5))
                inject(env, q32(sv))
                row, _ = step_and_record(env, rec, rng.uniform(-1, 1, size=2).astype(np.float32), KINDS.index("extreme_action"))
                rows.append(row)
        arrs = rows_to_arrays(rows, is_goal(env))
        arrs.update(env_constants(env))
        arrs["const_moi"] = np.asarray(env.ship_params.moi)
        arrs["kind_names"] = np.array(KINDS)
        arrs["env_id"] = np.array(base_id); arrs["class_name"] = np.array(cls_name); arrs["kwargs_json"] = np.array(json.dumps(kwargs))
        path = os.path.join(OUT, f"step_kw_{name}.npz")
        np.savez_compressed(path, **arrs)
        print(f"kw_{name}: {len(rows)} transitions, {int(arrs['done'].sum())} terminal, {int(arrs['goal_changed'].sum())} goal hits, "
              f"step_size {env.step_size} -> {path}", flush=True)


def stage_vector_field(args):
    """SpaceshipEnv.vector_field(raw_action, state_vec) (spaceship_env.py:96-100): the RHS of the ODE, for model-based users."""
    registry, envs, _ = load_env_layer()
    out = {}
    for fam in ("goal3p", "kepler_easy"):
        env = make_env(registry, envs, FAMILIES[fam])
        rng = np.random.RandomState(5)
        rows = []
        for _ in range(300):
            env.reset(); quantise_env(env)
            a = rng.uniform(-1, 1, size=2).astype(np.float32)
            sv = env._ship_state._state_vec.copy()
            f = env.vector_field(a, sv.copy())  # mutates its argument's omega (dynamic_model.py:138-141)
            rows.append((sv, a, np.array([p.center_pos for p in env.planets]).reshape(-1, 2), np.array(f, dtype=np.float64)))
        out[fam + "_state"] = np.array([r[0] for r in rows]); out[fam + "_action"] = np.array([r[1] for r in rows])
        out[fam + "_planets"] = np.array([r[2] for r in rows]); out[fam + "_field"] = np.array([r[3] for r in rows])
    path = os.path.join(OUT, "vector_field.npz")
    np.savez_compressed(path, **out)
    print("vector_field fixtures ->", path, flush=True)


def stage_reset(args):
    registry, envs, _ = load_env_layer()
    os.makedirs(OUT, exist_ok=True)
    for fam in ("goal2p", "goal3p", "goal4p"):
        env = make_env(registry, envs, FAMILIES[fam])
        st = goal_reset_stats(env, args.n_resets, 5, seed=20240 + len(env.planets))
        np.savez_compressed(os.path.join(OUT, f"reset_{fam}.npz"), **st)
        print(f"reset stats {fam}: same_tile={st['same_tile'] / args.n_resets}", flush=True)
    env = make_env(registry, envs, FAMILIES["kepler_circle"])
    np.savez_compressed(os.path.join(OUT, "reset_kepler.npz"), **kepler_reset_stats(env, args.n_resets, seed=777))


def stage_core(_args):
    """No shim: reference dynamic_model.make_step + real scipy on the inputs of every step fixture."""
    dm, planet_mod, sp_mod = load_core_only()
    assert "gym" not in sys.modules
    for fam in FAMILIES:
        path = os.path.join(OUT, f"step_{fam}.npz")
        d = dict(np.load(path))
        n = int(d["const_n_planets"])
        planets = [planet_mod.Planet(mass=float(d["const_planet_mass"][i]), radius=float(d["const_planet_radius"][i]),
                                     center_pos=np.zeros(2)) for i in range(n)]
        ship = sp_mod.ShipParams(sp_mod.Steering(1), mass=float(d["const_ship_mass"]), moi=0.01,
                                 max_engine_force=float(d["const_max_engine_force"]), max_thruster_force=0.05)
        events = dm.make_termination_events(float(d["const_world_size"]), float(d["const_max_abs_vel_angle"]), planets)
        M = len(d["state0"])
        s1 = np.empty((M, 6)); dn = np.empty(M, np.uint8)
        for i in range(M):
            if "planets" in d:
                for p, xy in zip(planets, d["planets"][i]):
                    p.center_pos = xy.copy()
            raw = d["action"][i]  # float32
            # ContinuousSpaceshipEnv._translate_raw_action (spaceship_env.py:210-214), float32 arithmetic
            action = np.array(((raw[0] + 1) / 2, raw[1]))
            sv, done = dm.make_step(ship, planets, d["state0"][i].copy(), action, float(d["const_step_size"]), events)
            s1[i], dn[i] = sv, done
        assert np.array_equal(s1, d["state1"]) and np.array_equal(dn, d["done"]), fam
        d["core_state1"], d["core_done"] = s1, dn
        np.savez_compressed(path, **d)
        print(f"core {fam}: make_step without shim == env layer, bitwise, on {M} transitions", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", choices=["all", "env", "core", "reset", "discrete", "random_orbits", "acceleration", "vector_field", "kwargs"], default="all")
    ap.add_argument("--rollout-steps", type=int, default=20000)
    ap.add_argument("--keep-nonterminal", type=int, default=1200)
    ap.add_argument("--forced-each", type=int, default=80)
    ap.add_argument("--n-resets", type=int, default=100000)
    args = ap.parse_args()
    if args.stage == "all":
        for st in ("env", "core", "reset", "discrete", "random_orbits", "acceleration", "vector_field", "kwargs"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--stage", st,
                                   "--rollout-steps", str(args.rollout_steps),
                                   "--keep-nonterminal", str(args.keep_nonterminal),
                                   "--forced-each", str(args.forced_each), "--n-resets", str(args.n_resets)])
    elif args.stage == "env":
        stage_env(args)
    elif args.stage == "reset":
        stage_reset(args)
    elif args.stage == "discrete":
        stage_discrete(args)
    elif args.stage == "random_orbits":
        stage_random_orbits(args)
    elif args.stage == "acceleration":
        stage_acceleration(args)
    elif args.stage == "vector_field":
        stage_vector_field(args)
    elif args.stage == "kwargs":
        stage_kwargs(args)
    else:
        stage_core(args)


if __name__ == "__main__":
    main()
