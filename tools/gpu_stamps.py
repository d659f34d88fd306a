#!/usr/bin/env python3
"""Diagnostic: where a wave of the step kernel spends its cycles (stamped build, -DSG_STAMPS).  Run on the GPU box:
    SPACEGYM_LIB=space_gym_amd/lib/libspacegym_hip_stamps.so python tools/gpu_stamps.py
Reads SHARES, not lengths: the stamped build drains memory counters at each stamp."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import space_gym_amd as sg  # noqa: E402
from space_gym_amd import _native  # noqa: E402

SLOTS, WAVES = 16, 4096


def read(lib):
    buf = np.zeros(SLOTS * WAVES, np.uint64)
    assert lib.sg_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int64(buf.size)) == 0
    return buf.reshape(WAVES, SLOTS)


def report(st, n_waves):
    st = st[:n_waves].astype(np.float64)
    d = {"loads (issue -> all back)": st[:, 1] - st[:, 0], "env_step (RK45 + events + reward + obs)": st[:, 2] - st[:, 1],
         "  translate + sincos(theta0) .. initial-step rule done": st[:, 9] - st[:, 8],
         "  initial-step rule -> end of 1st RK iteration (incl. its event check)": st[:, 10] - st[:, 9],
         "  rest of the RK loop (2nd/3rd iterations, event roots)": st[:, 11] - st[:, 10],
         "  reward (fp64)": st[:, 12] - st[:, 11], "  state update + observation": st[:, 2] - st[:, 12],
         "owner stores": st[:, 3] - st[:, 2], "service (ballot .. restarts/resamples done)": st[:, 4] - st[:, 3],
         "  [waves with a restart] entry -> Philox + word exchange": st[:, 13] - st[:, 3],
         "  [..] layout + tiles + goal tile": st[:, 14] - st[:, 13],
         "  [..] disc samples + Box-Muller + exchange": st[:, 15] - st[:, 14],
         "  [..] column/obs stores -> end of service": st[:, 4] - st[:, 15],
         "store drain": st[:, 5] - st[:, 4], "whole wave": st[:, 5] - st[:, 0]}
    real = (st[:, 7] - st[:, 6])
    had_restart = st[:, 13] > st[:, 3]  # slots 13..15 are only written (this launch) by waves that restarted an env
    out = {}
    for k, v in d.items():
        if k.startswith("  [") and had_restart.any():
            v = v[had_restart]
        out[k] = dict(mean=float(v.mean()), p50=float(np.median(v)), p95=float(np.percentile(v, 95)), max=float(v.max()), n=int(v.size))
    out["clock_GHz (cycles / 100MHz ticks)"] = float((st[:, 5] - st[:, 0]).sum() / (real.sum() * 10.0))
    out["kernel span: first wave start -> last wave end (us, 100 MHz clock)"] = float((st[:, 7].max() - st[:, 6].min()) / 100.0)
    out["wave start skew (us)"] = float((st[:, 6].max() - st[:, 6].min()) / 100.0)
    return out


def main():
    lib = _native.load()
    lib.sg_debug_read_stamps.argtypes = [C.c_void_p, C.c_int64]
    B = 65536
    env = sg.make_vec("GoalContinuous3P-v0", B, seed=0)
    env.set_unfused_rollout(True)  # the stamps live in the per-step kernel
    dev = torch.device("cuda", 0)
    K = 260
    acts = torch.rand((K, B, 2), device=dev) * 2 - 1
    obs = torch.empty((K, B, env.obs_dim), device=dev); rew = torch.empty((K, B), device=dev)
    done = torch.empty((K, B), dtype=torch.uint8, device=dev); trunc = torch.empty_like(done)
    env.reset_torch()
    env.rollout_torch(acts[:2], obs[:2], rew[:2], done[:2], trunc[:2]); torch.cuda.synchronize()
    res = {"young episodes (step 3)": None, "steady state (step 260)": None}
    env.rollout_torch(acts[2:3], obs[2:3], rew[2:3], done[2:3], trunc[2:3]); torch.cuda.synchronize()
    res["young episodes (step 3)"] = report(read(lib), B // 64)
    env.rollout_torch(acts[3:], obs[3:], rew[3:], done[3:], trunc[3:]); torch.cuda.synchronize()
    res["steady state (step 260)"] = report(read(lib), B // 64)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
