#!/usr/bin/env python3
"""Kernel-time breakdown on the GPU box (measurement aid): young episodes (no events, no restarts) vs steady state,
plus a dump of finished envs whose terminal observation is not on a boundary (debugging aid)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import space_gym_amd as sg  # noqa: E402


def timed(env, acts, obs, rew, done, trunc, k0, k):
    env.set_profiling(True)
    env.rollout_torch(acts[k0:k0 + k], obs[k0:k0 + k], rew[k0:k0 + k], done[k0:k0 + k], trunc[k0:k0 + k])
    torch.cuda.synchronize()
    n, tot, mn, mx = env.get_profile()
    env.set_profiling(False)
    return dict(launches=n, avg_us=tot * 1e3 / n, min_us=mn * 1e3, max_us=mx * 1e3,
                done_per_step=float(done[k0:k0 + k].sum().item()) / k)


def main():
    env_id = sys.argv[1] if len(sys.argv) > 1 else "GoalContinuous3P-v0"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    dev = torch.device("cuda", 0)
    out = {}
    env = sg.make_vec(env_id, B, seed=0)
    K = 400
    acts = torch.rand((K, B, 2), device=dev) * 2 - 1
    obs = torch.empty((K, B, env.obs_dim), device=dev); rew = torch.empty((K, B), device=dev)
    done = torch.empty((K, B), dtype=torch.uint8, device=dev); trunc = torch.empty_like(done)
    env.reset_torch(); torch.cuda.synchronize()
    out["steps 0-7 after reset (no events, no restarts)"] = timed(env, acts, obs, rew, done, trunc, 0, 8)
    env.reset_torch(); torch.cuda.synchronize()
    out["same, second time (warm)"] = timed(env, acts, obs, rew, done, trunc, 0, 8)
    env.rollout_torch(acts[8:200], obs[8:200], rew[8:200], done[8:200], trunc[8:200]); torch.cuda.synchronize()
    out["steady state"] = timed(env, acts, obs, rew, done, trunc, 200, 200)
    env.set_auto_reset(False)
    out["steady state, one step with auto-reset off"] = timed(env, acts, obs, rew, done, trunc, 0, 1)
    print(json.dumps({env_id: out}, indent=1))
    env.close()

    # ---- debugging aid: terminal observations off the boundary
    if env_id.startswith("Goal"):
        n = B
        env = sg.make_vec(env_id, n, seed=3)
        env.reset()
        rng = np.random.default_rng(5)
        R = float(np.sqrt(sg.registration.ENV_SPECS[env_id]["n_planets"] * 0 + 1))  # placeholder, fixed below
        from space_gym_amd import registration  # noqa: F401
        bad = []
        for step in range(120):
            a = rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
            st0 = env.get_state()
            obs_, rew_, done_, info = env.step(a)
            term = done_ & ~info["TimeLimit.truncated"]
            if term.any():
                tob = info["terminal_observation"]
                p = st0["planets"]
                dist = np.linalg.norm(p - tob[:, None, :2], axis=2)
                Rp = {2: 0.45, 3: 0.32142857, 4: 0.25}[p.shape[1]]
                g = np.minimum(np.abs(dist - Rp).min(axis=1), np.abs(1.5 - np.abs(tob[:, :2]).max(axis=1)))
                idx = np.nonzero(term & (g > 2e-6))[0]
                for i in idx[:20]:
                    bad.append(dict(step=step, ship=st0["ship"][i], planets=p[i], goal=st0["goal"][i], action=a[i],
                                    tob=tob[i], g=g[i], reward=rew_[i]))
        print("off-boundary terminal observations:", len(bad))
        if bad:
            np.savez("gpurun_out/debug_cases.npz", **{k: np.array([b[k] for b in bad]) for k in bad[0]})
        env.close()


if __name__ == "__main__":
    main()
