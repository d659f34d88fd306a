#!/usr/bin/env python3
"""Measurement aid (GPU box): duration of the one-launch-per-step kernels (dispatch start/stop events) in steady state, for each
step-kernel plan (SPACEGYM_STEP_KERNEL=single: one wave per 64 envs, =pair: pilot + finisher waves) at several batch sizes.

    python tools/gpu_step_times.py ENV_ID BATCH[,BATCH...] [single,pair] [extra make_vec kwargs as k=v ...]

SPACEGYM_LIB selects another build of the library (tools/build_rev.sh) for same-box A/B runs.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(env_id, B, plan, kw, n=200):
    os.environ["SPACEGYM_STEP_KERNEL"] = plan
    import torch
    import space_gym_amd as sg
    dev = torch.device("cuda", 0)
    env = sg.make_vec(env_id, B, seed=0, **kw)
    K = 50
    if env.discrete:
        acts = torch.randint(0, 6, (K, B), device=dev, dtype=torch.int32)
    else:
        acts = torch.rand((K, B, 2), device=dev) * 2 - 1
    obs = torch.empty((K, B, env.obs_dim), device=dev); rew = torch.empty((K, B), device=dev)
    done = torch.empty((K, B), dtype=torch.uint8, device=dev); trunc = torch.empty_like(done)
    env.reset_torch()
    for _ in range(max(1, 3000 // K)):  # stationary episode ages (fused rollout), working clocks
        env.rollout_torch(acts, obs, rew, done, trunc)
    env.set_unfused_rollout(True)
    name = env.rollout_kernel(1)
    for _ in range(4):
        env.rollout_torch(acts, obs, rew, done, trunc)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(max(1, n // K)):
        env.rollout_torch(acts, obs, rew, done, trunc)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e6 / (max(1, n // K) * K)
    env.set_profiling(True)
    for _ in range(max(1, n // K)):
        env.rollout_torch(acts, obs, rew, done, trunc)
    torch.cuda.synchronize()
    cnt, tot, mn, mx = env.get_profile()
    env.set_profiling(False)
    env.check_status()
    env.close()
    bytes_per = (113 + 16 * env.n_planets) if env.spec["family"] == "goal" else 109
    avg = tot * 1e3 / cnt
    print("%-22s B=%-8d %-34s n=%4d avg %7.2f us  min %7.2f  max %7.2f   frac %.3f   %.2f G env-steps/s   wall %.2f us/step" % (
        env_id, B, name, cnt, avg, mn * 1e3, mx * 1e3, B * bytes_per / (avg * 1e-6) / 8e12, B / (avg * 1e-6) / 1e9, wall), flush=True)


def main():
    env_id = sys.argv[1] if len(sys.argv) > 1 else "GoalContinuous3P-v0"
    batches = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "65536,1048576").split(",")]
    plans = (sys.argv[3] if len(sys.argv) > 3 else "single,pair").split(",")
    kw = {}
    for a in sys.argv[4:]:
        k, v = a.split("=")
        kw[k] = v
    for B in batches:
        for plan in plans:
            measure(env_id, B, plan, kw)


if __name__ == "__main__":
    main()
