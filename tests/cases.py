"""Input generators and statistics shared by the CPU (host twin) and GPU (-m gpu) tests."""
import numpy as np


def unit(a):
    return np.stack([np.cos(a), np.sin(a)], -1)


def adversarial_event_cases(o, n=60000, seed=3):
    """Terminal-step stress inputs for oracle `o`'s env id: ships 0.2 mm..7 cm from a planet / the border circle / a wall /
    a corner, heading within 87 degrees of the normal at up to 2.5 units/s (near-tangent grazes have ill-conditioned roots;
    the wall events min(W/2 -+ x, W/2 -+ y) of dynamic_model.py:196-205 have a kink at corners).
    Returns (state0 f32 [m, 6], action f32 [m, 2], planets f32 [m, N, 2] | None, goal f32 [m, 2] | None)."""
    rng = np.random.default_rng(seed)
    envs, _ = o.vec_reset(n, seed=1)
    P = g = None
    if o.is_goal:
        N, R = o.n_planets, o.params.planet_radius[0]
        P, g = envs["planets_xy"][:, :N].astype(np.float32), envs["goal_xy"].astype(np.float32)
        j, ang = rng.integers(0, N, n), rng.uniform(0, 2 * np.pi, n)
        pos = P[np.arange(n), j] + unit(ang) * (R + rng.uniform(0.0002, 0.07, n))[:, None]
        vel = -unit(ang + np.deg2rad(rng.uniform(-87, 87, n))) * rng.uniform(0.05, 2.5, n)[:, None]
        w = rng.uniform(size=n) < 0.4  # walls and corners
        k = int(w.sum())
        posw = rng.uniform(-1.45, 1.45, (k, 2)); ax = rng.integers(0, 2, k); sg = rng.choice([-1., 1.], k)
        posw[np.arange(k), ax] = sg * (1.5 - rng.uniform(0, 0.06, k))
        cor = rng.uniform(size=k) < 0.3
        posw[cor, 1 - ax[cor]] = rng.choice([-1., 1.], int(cor.sum())) * (1.5 - rng.uniform(0, 0.06, int(cor.sum())))
        velw = rng.normal(size=(k, 2)) * 0.7; velw[np.arange(k), ax] = sg * rng.uniform(0.05, 2.0, k)
        pos[w], vel[w] = posw, velw
        ok = np.all(np.abs(pos) < 1.5, axis=1) & (np.linalg.norm(P - pos[:, None], axis=2).min(1) > R)
    else:
        inner, ang = rng.uniform(size=n) < 0.5, rng.uniform(0, 2 * np.pi, n)
        rad = np.where(inner, 0.2 + rng.uniform(0.0002, 0.07, n), 3.0 - rng.uniform(0.0002, 0.07, n))
        pos = unit(ang) * rad[:, None]
        vel = np.where(inner, -1.0, 1.0)[:, None] * unit(ang + np.deg2rad(rng.uniform(-87, 87, n))) * rng.uniform(0.05, 2.5, n)[:, None]
        ok = np.ones(n, bool)
    s0 = np.concatenate([pos, rng.uniform(0, 2 * np.pi, (n, 1)), vel, rng.normal(size=(n, 1))], 1).astype(np.float32)[ok]
    a = rng.uniform(-1, 1, (len(s0), 2)).astype(np.float32)
    Pk, gk = (P[ok], g[ok]) if P is not None else (None, None)
    return s0, a, Pk, gk


def tangential_graze_cases(o, n=60000, seed=4):
    """Stress inputs for the "can a dip between the two ends of the env-step be excluded" decision (Integrator::no_graze):
    straight paths whose point of closest approach to a planet / the border circle / a wall lies INSIDE the env-step (at
    0.1..0.9 of it), -0.3..+0.3 mm from the surface (the engine and gravity bend the real path by up to 1 mm: some of these
    dip below the surface and come out again, some touch, some miss), at 0.15..2.5 units/s.  scipy sees a dip only at the
    end of one of its own RK steps; the engine has to decide the same.  Same return values as adversarial_event_cases."""
    rng = np.random.default_rng(seed)
    envs, _ = o.vec_reset(n, seed=2)
    h = 0.07
    tc = rng.uniform(0.1, 0.9, n) * h          # time of closest approach
    speed = rng.uniform(0.15, 2.5, n)
    depth = rng.uniform(-3e-4, 3e-4, n)        # > 0: the straight path penetrates the surface by this much
    ang = rng.uniform(0, 2 * np.pi, n)
    side = rng.choice([-1.0, 1.0], n)
    P = g = None
    if o.is_goal:
        N, R = o.n_planets, o.params.planet_radius[0]
        P, g = envs["planets_xy"][:, :N].astype(np.float32), envs["goal_xy"].astype(np.float32)
        j = rng.integers(0, N, n)
        nrm = unit(ang)                                           # outward normal at the closest point
        tan = np.stack([-nrm[:, 1], nrm[:, 0]], -1) * side[:, None]
        close = P[np.arange(n), j] + nrm * (R - depth)[:, None]
        w = rng.uniform(size=n) < 0.35                            # walls: closest approach to x or y = +-1.5
        k = int(w.sum())
        ax, sg = rng.integers(0, 2, k), rng.choice([-1.0, 1.0], k)
        cw = rng.uniform(-1.2, 1.2, (k, 2)); cw[np.arange(k), ax] = sg * (1.5 + depth[w])
        tw_ = np.zeros((k, 2)); tw_[np.arange(k), 1 - ax] = side[w]
        close[w], tan[w] = cw, tw_
        pos = close - tan * (speed * tc)[:, None]
        vel = tan * speed[:, None]
        ok = np.all(np.abs(pos) < 1.5, axis=1) & (np.linalg.norm(P - pos[:, None], axis=2).min(1) > R)
    else:
        inner = rng.uniform(size=n) < 0.5
        nrm = unit(ang)
        tan = np.stack([-nrm[:, 1], nrm[:, 0]], -1) * side[:, None]
        rad = np.where(inner, 0.2 - depth, 3.0 + depth)
        close = nrm * rad[:, None]
        pos = close - tan * (speed * tc)[:, None]
        vel = tan * speed[:, None]
        r0 = np.linalg.norm(pos, axis=1)
        ok = (r0 > 0.2) & (r0 < 3.0)
    s0 = np.concatenate([pos, rng.uniform(0, 2 * np.pi, (n, 1)), vel, rng.normal(size=(n, 1))], 1).astype(np.float32)[ok]
    a = rng.uniform(-1, 1, (len(s0), 2)).astype(np.float32)
    Pk, gk = (P[ok], g[ok]) if P is not None else (None, None)
    return s0, a, Pk, gk


def chi2_ok(obs, exp, sigmas=5.0):
    """Pearson chi-square of two count vectors (both sampled): |chi2 - dof| within `sigmas` of its spread."""
    obs, exp = np.asarray(obs, float).ravel(), np.asarray(exp, float).ravel()
    exp = exp * obs.sum() / exp.sum()
    m = (obs + exp) > 20
    chi2 = ((obs[m] - exp[m]) ** 2 / (obs[m] + exp[m])).sum()  # two-sample form
    dof = m.sum() - 1
    return abs(chi2 - dof) < sigmas * np.sqrt(2 * dof) + 5, (chi2, dof)


def check_reset_statistics(ref, fam, state, planets, goals, ship_tile, goal_tile, free_counts, col_shift, flags):
    """Distributional parity of a reset + goal-resample chain with statistics of 1e5 resets of the reference
    (GoalEnv._reset goal.py:133-145; HexagonalTiling hexagonal_tiling.py:53-134), tests/golden/reset_*.npz.
    state [n, 6]; planets [n, N, 2]; goals [n, hits+1, 2]; ship_tile / goal_tile int [n, hits+1]; free_counts uint64 [n, hits+1]
    (sixteen 4-bit counters); col_shift [n, >=cols]; flags = (#case_b, #flip)."""
    n, hits = len(state), goals.shape[1] - 1
    half, nt = 1.5, len(ref["ship_tile"])
    rows, cols = {"goal2p": (2, 2), "goal3p": (3, 3), "goal4p": (4, 4)}[fam]
    assert nt == rows * cols

    def h2(xy):
        return np.histogram2d(xy[:, 0], xy[:, 1], bins=24, range=[[-half, half]] * 2)[0]

    ok, info = chi2_ok(h2(state[:, :2]), ref["ship_hist"]); assert ok, ("ship", info)
    ok, info = chi2_ok(h2(planets.reshape(-1, 2)), ref["planets_hist"]); assert ok, ("planets", info)
    ok, info = chi2_ok(np.bincount(ship_tile[:, 0], minlength=nt), ref["ship_tile"]); assert ok, ("ship tile", info)
    n_ref = int(ref["n_resets"])
    assert np.abs(np.asarray(flags) / n - 0.5).max() < 0.01 and np.abs(ref["case_b_flip"] / n_ref - 0.5).max() < 0.01
    assert np.abs(col_shift[:, :cols].mean(0) - ref["col_shift_mean"]).max() < 5e-3
    for k in range(hits + 1):
        ok, info = chi2_ok(h2(goals[:, k]), ref["goal_hist"][k]); assert ok, ("goal", k, info)
        ok, info = chi2_ok(np.bincount(goal_tile[:, k], minlength=nt), ref["goal_tile"][k]); assert ok, ("goal tile", k, info)
        same = (goal_tile[:, k] == ship_tile[:, k]).mean()
        assert abs(same - ref["same_tile"][k] / n_ref) < 0.008, (k, same)
        taxi = np.abs(goal_tile[:, k] // cols - ship_tile[:, k] // cols) + np.abs(goal_tile[:, k] % cols - ship_tile[:, k] % cols)
        ok, info = chi2_ok(np.bincount(taxi, minlength=rows + cols), ref["taxi"][k]); assert ok, ("taxi", k, info)
        free_len = ((free_counts[:, k, None] >> (4 * np.arange(16, dtype=np.uint64))) & np.uint64(15)).sum(1)
        ok, info = chi2_ok(np.bincount(free_len.astype(int), minlength=ref["free_len"].shape[1]), ref["free_len"][k])
        assert ok, ("free list length", k, info)
    kin = state
    ok, info = chi2_ok(np.histogram(kin[:, 2], bins=16, range=(0, 2 * np.pi))[0], ref["theta_hist"]); assert ok, ("theta", info)
    ok, info = chi2_ok(np.histogram(kin[:, 3:5].ravel(), bins=32, range=(-0.35, 0.35))[0], ref["vel_hist"]); assert ok, ("vel", info)
    ok, info = chi2_ok(np.histogram(kin[:, 5], bins=32, range=(-4.2 - 1e-9, 4.2 + 1e-9))[0], ref["omega_hist"]); assert ok, ("omega", info)
    assert abs(kin[:, 3:5].std() - ref["vel_std"]) < 1e-3 and abs(kin[:, 5].std() - ref["omega_std"]) < 0.02
    # clearances the tiling guarantees by construction: never tighter than what the reference ever produced - eps
    R, rs = float(ref["planet_radius"]), float(ref["ship_radius"])
    mc = dict(zip([str(k) for k in ref["min_clear_keys"]], ref["min_clear"]))
    dist = np.linalg.norm(planets - state[:, None, :2], axis=2).min(1) - R
    assert dist.min() > rs - 1e-5 and mc["ship_planet"] > rs - 1e-9      # ship disc never overlaps a planet
    assert (half - np.abs(state[:, :2]).max(1)).min() > rs - 1e-5 and mc["ship_wall"] > rs - 1e-9
    assert (half - np.abs(planets).max(2).min(1) - R).min() > -1e-5 and mc["planet_wall"] > -1e-9
