#!/usr/bin/env python3
"""Diagnostic: cycles per step and phase of the pilot and finisher waves of goal_pair_rollout_kernel
(build with -DSG_STAMPS -DSG_STAMPS_ACC_ONLY).  Run on the GPU box:
    SPACEGYM_LIB=space_gym_amd/lib/libspacegym_hip_accstamps.so python tools/gpu_pair_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import space_gym_amd as sg  # noqa: E402
from space_gym_amd import _native  # noqa: E402

SLOTS, WAVES = 16, 4096
PILOT = {0: "top of step (action, waits, loop)", 1: "begin: constants, f(t0), g(t0)", 2: "fast step (+ scipy's sequence)",
         3: "replay records + state update", 5: "ring record + publish", 8: "TimeLimit + restart"}
FIN = {0: "loop", 1: "wait for the pilot", 3: "read record, release slot", 4: "reward + update",
       6: "terminal obs (TOBS) + refill passes", 12: "restart: next episode out of the queue", 13: "observation", 5: "output stores",
       2: "goal resamples", 9: "replay passes"}


def main():
    lib = _native.load()
    lib.sg_debug_read_stamps.argtypes = [C.c_void_p, C.c_int64]
    B, K = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 500
    env = sg.make_vec("GoalContinuous3P-v0", B, seed=0)
    dev = torch.device("cuda", 0)
    acts = torch.rand((K, B, 2), device=dev) * 2 - 1
    obs = torch.empty((K, B, env.obs_dim), device=dev); rew = torch.empty((K, B), device=dev)
    done = torch.empty((K, B), dtype=torch.uint8, device=dev); trunc = torch.empty_like(done)
    env.reset_torch()
    for _ in range(2):
        env.rollout_torch(acts, obs, rew, done, trunc)
    torch.cuda.synchronize()
    buf = np.zeros(SLOTS * WAVES, np.uint64)
    assert lib.sg_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int64(buf.size)) == 0
    st = buf.reshape(WAVES, SLOTS)[: B // 256 * 8].reshape(-1, 8, SLOTS).astype(np.float64) / K
    pil, fin = st[:, :4].reshape(-1, SLOTS), st[:, 4:].reshape(-1, SLOTS)
    print("pilot waves: cycles per step (mean over %d waves)" % len(pil))
    for k, name in PILOT.items():
        print("  %-40s %8.0f" % (name, pil[:, k].mean()))
    print("  %-40s %8.0f" % ("total", sum(pil[:, k].mean() for k in PILOT)))
    print("  wave-steps with a terminal event %.3f, with a lane on scipy's sequence %.4f (more than one RK step: %.4f)" % (
        pil[:, 7].mean(), pil[:, 10].mean(), pil[:, 11].mean()))
    print("finisher waves: cycles per step")
    for k, name in FIN.items():
        print("  %-40s %8.0f" % (name, fin[:, k].mean()))
    print("  %-40s %8.0f" % ("total", sum(fin[:, k].mean() for k in FIN)))
    print("  queue refill passes per step %.4f, goal resample passes per step %.4f" % (fin[:, 10].mean(), fin[:, 11].mean()))
    # which phase makes the slowest waves slow: the 1 % of the waves with the longest step loops against the average wave
    for name, w, phases in (("pilot", pil, PILOT), ("finisher", fin, FIN)):
        tot = sum(w[:, k] for k in phases)
        slow = np.argsort(-tot)[: max(1, len(tot) // 100)]
        print("%s: slowest 1 %% of the waves, cycles per step above the average wave (total %+.0f)" % (name, tot[slow].mean() - tot.mean()))
        for k, nm in phases.items():
            print("  %-40s %+8.0f" % (nm, w[slow, k].mean() - w[:, k].mean()))


if __name__ == "__main__":
    main()
