#!/usr/bin/env python3
"""Measurement aid (GPU box): kernel time of ONE sg_rollout_device launch as a function of the steps per launch K, in
steady state (envs pre-rolled to their stationary mix of episode ages).  The slope is the per-step cost, the intercept the
fixed cost of a launch (workgroup start, state load / store, pipeline fill and drain of the pilot + finisher pair).

    python tools/gpu_ksweep.py [env_id] [batch] [K,K,...]
SG_KSWEEP_TOBS=1: with a terminal-observation list (sg_rollout_device_terminal).
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import space_gym_amd as sg  # noqa: E402


def main():
    env_id = sys.argv[1] if len(sys.argv) > 1 else "GoalContinuous3P-v0"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    Ks = [int(k) for k in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8, 12, 16, 20, 24, 32, 48, 64, 100, 200]
    reps = int(os.environ.get("SG_KSWEEP_REPS", "12"))
    dev = torch.device("cuda", 0)
    env = sg.make_vec(env_id, B, seed=0)
    Kmax = max(max(Ks), 200)
    if env.discrete:
        acts = torch.randint(0, 6, (Kmax, B), device=dev, dtype=torch.int32)
    else:
        acts = torch.rand((Kmax, B, 2), device=dev) * 2 - 1
    obs = torch.empty((Kmax, B, env.obs_dim), device=dev); rew = torch.empty((Kmax, B), device=dev)
    done = torch.empty((Kmax, B), dtype=torch.uint8, device=dev); trunc = torch.empty_like(done)
    env.reset_torch()
    for _ in range(60):  # 12 000 steps: stationary episode ages, working clocks
        env.rollout_torch(acts[:200], obs[:200], rew[:200], done[:200], trunc[:200])
    torch.cuda.synchronize()
    rows = []
    tobs = bool(os.environ.get("SG_KSWEEP_TOBS"))
    for K in Ks:
        term = dict(terminal=env.terminal_list_torch(max(4096, K * B // 8))) if tobs else {}
        for _ in range(3):
            env.rollout_torch(acts[:K], obs[:K], rew[:K], done[:K], trunc[:K], **term)
        torch.cuda.synchronize()
        env.set_profiling(True)
        for _ in range(reps):
            env.rollout_torch(acts[:K], obs[:K], rew[:K], done[:K], trunc[:K], **term)
        torch.cuda.synchronize()
        n, tot, mn, mx = env.get_profile()
        env.set_profiling(False)
        if tobs:
            env.terminal_records(term["terminal"])  # (raises on overflow)
        rows.append(dict(K=K, kernel=env.rollout_kernel(K) + (" +terminal list" if tobs else ""), launches=n, avg_us=tot * 1e3 / n, min_us=mn * 1e3, max_us=mx * 1e3,
                         us_per_step=tot * 1e3 / n / K))
        print("K=%4d  %-40s avg %8.2f us  min %8.2f  max %8.2f   %.3f us/step" % (
            K, rows[-1]["kernel"], rows[-1]["avg_us"], rows[-1]["min_us"], rows[-1]["max_us"], rows[-1]["us_per_step"]), flush=True)
    k = np.array([r["K"] for r in rows if r["K"] >= 8], float); t = np.array([r["min_us"] for r in rows if r["K"] >= 8])
    if len(k) >= 2:
        slope, icpt = np.polyfit(k, t, 1)
        print("fit over K >= 8 (min times): %.3f us/step + %.2f us per launch" % (slope, icpt))
        rows.append(dict(fit_us_per_step=slope, fit_us_per_launch=icpt))
    print(json.dumps(rows))
    env.close()


if __name__ == "__main__":
    main()
