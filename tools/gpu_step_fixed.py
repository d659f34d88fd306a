#!/usr/bin/env python3
"""Measurement aid (GPU box): the one-launch-per-step kernel with FIXED output buffers (sg_step_device as a policy loop calls it:
the same obs / reward / flag tensors every step), against tools/gpu_step_times.py's [K, B, ...] rollout rows.
    python tools/gpu_step_fixed.py ENV_ID BATCH[,BATCH...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ.setdefault("SPACEGYM_STEP_KERNEL", "single")
    import torch
    import space_gym_amd as sg
    env_id = sys.argv[1] if len(sys.argv) > 1 else "GoalContinuous3P-v0"
    for B in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1048576").split(",")]:
        env = sg.make_vec(env_id, B, seed=0)
        dev = torch.device("cuda", 0)
        K = 50
        acts = torch.rand((K, B, 2), device=dev) * 2 - 1 if not env.discrete else torch.randint(0, 6, (K, B), device=dev, dtype=torch.int32)
        obs = torch.empty((K, B, env.obs_dim), device=dev); rew = torch.empty((K, B), device=dev)
        done = torch.empty((K, B), dtype=torch.uint8, device=dev); trunc = torch.empty_like(done)
        env.reset_torch()
        for _ in range(60):
            env.rollout_torch(acts, obs, rew, done, trunc)
        del obs, rew, done, trunc
        for t in range(20):
            env.step_torch(acts[t % K])
        torch.cuda.synchronize()
        env.set_profiling(True)
        for t in range(200):
            env.step_torch(acts[t % K])
        torch.cuda.synchronize()
        cnt, tot, mn, mx = env.get_profile()
        env.set_profiling(False)
        bytes_per = (113 + 16 * env.n_planets) if env.spec["family"] == "goal" else 109
        avg = tot * 1e3 / cnt
        print("%-22s B=%-8d fixed output buffers: n=%4d avg %7.2f us  min %7.2f  max %7.2f   frac %.3f" % (
            env_id, B, cnt, avg, mn * 1e3, mx * 1e3, B * bytes_per / (avg * 1e-6) / 8e12), flush=True)
        env.close()


if __name__ == "__main__":
    main()
