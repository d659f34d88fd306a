#!/usr/bin/env python3
"""Diagnostic (build with -DSG_STAMPS -DSG_STAMPS_ACC_ONLY -DSG_STAMPS_PASSES; GPU box): wall-clock (100 MHz) start of every
subtile pass of the waves of goal_step_kernel, of the pooled pass and of the exit, relative to the launch's first stamp."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import space_gym_amd as sg  # noqa: E402
from space_gym_amd import _native  # noqa: E402

SLOTS, WAVES = 16, 4096
os.environ["SPACEGYM_STEP_KERNEL"] = "single"
lib = _native.load()
lib.sg_debug_read_stamps.argtypes = [C.c_void_p, C.c_int64]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
env = sg.make_vec("GoalContinuous3P-v0", B, seed=0)
dev = torch.device("cuda", 0)
K = 50
acts = torch.rand((K, B, 2), device=dev) * 2 - 1
obs = torch.empty((K, B, env.obs_dim), device=dev); rew = torch.empty((K, B), device=dev)
done = torch.empty((K, B), dtype=torch.uint8, device=dev); trunc = torch.empty_like(done)
env.reset_torch()
for _ in range(20):
    env.rollout_torch(acts, obs, rew, done, trunc)
env.set_unfused_rollout(True)
env.rollout_torch(acts, obs, rew, done, trunc)
torch.cuda.synchronize()
buf = np.zeros(SLOTS * WAVES, np.uint64)
assert lib.sg_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int64(buf.size)) == 0
st = buf.reshape(WAVES, SLOTS)[:2048].astype(np.float64)
t0 = st[:, 0].min()
for k in list(range(8)) + [12, 13]:
    d = (st[:, k] - t0) / 100.0
    print("slot %2d (%s): us since the first pass start  mean %7.2f  p5 %7.2f  p95 %7.2f  max %7.2f" % (
        k, "pass %d" % k if k < 12 else ("pooled pass" if k == 12 else "exit"), d.mean(), np.percentile(d, 5), np.percentile(d, 95), d.max()))
env.close()
