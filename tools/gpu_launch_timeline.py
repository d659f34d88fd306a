#!/usr/bin/env python3
"""Diagnostic: where the fixed cost of a wave-pair rollout launch goes.  Wall-clock (100 MHz) stamps of every wave at kernel
entry, after the one barrier, at the end of its step loop and at its exit (build with -DSG_STAMPS -DSG_STAMPS_ACC_ONLY).
    SPACEGYM_LIB=space_gym_amd/lib/libspacegym_hip_accstamps.so python tools/gpu_launch_timeline.py [K ...]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import space_gym_amd as sg  # noqa: E402
from space_gym_amd import _native  # noqa: E402


def main():
    lib = _native.load()
    lib.sg_debug_read_real.argtypes = [C.c_void_p, C.c_int64]
    Ks = [int(k) for k in sys.argv[1:]] or [1, 20]
    B = 65536
    env = sg.make_vec("GoalContinuous3P-v0", B, seed=0)
    dev = torch.device("cuda", 0)
    Kmax = max(max(Ks), 200)
    acts = torch.rand((Kmax, B, 2), device=dev) * 2 - 1
    obs = torch.empty((Kmax, B, env.obs_dim), device=dev); rew = torch.empty((Kmax, B), device=dev)
    done = torch.empty((Kmax, B), dtype=torch.uint8, device=dev); trunc = torch.empty_like(done)
    env.reset_torch()
    for _ in range(40):
        env.rollout_torch(acts[:200], obs[:200], rew[:200], done[:200], trunc[:200])
    for K in Ks:
        rows = []
        for rep in range(6):
            env.set_profiling(True)
            env.rollout_torch(acts[:K], obs[:K], rew[:K], done[:K], trunc[:K])
            torch.cuda.synchronize()
            n, tot, mn, mx = env.get_profile()
            env.set_profiling(False)
            buf = np.zeros(4 * 4096, np.uint64)
            assert lib.sg_debug_read_real(buf.ctypes.data_as(C.c_void_p), C.c_int64(buf.size)) == 0
            st = buf.reshape(4096, 4)[: B // 256 * 8].reshape(-1, 8, 4).astype(np.float64) * 0.01  # us
            t0 = st[:, :, 0].min()
            pil, fin = st[:, :4], st[:, 4:]
            rows.append([tot * 1e3, st[:, :, 0].max() - t0, st[:, :, 1].mean() - t0, st[:, :, 1].max() - t0,
                         pil[:, :, 2].mean() - t0, pil[:, :, 2].max() - t0, fin[:, :, 2].mean() - t0, fin[:, :, 2].max() - t0,
                         pil[:, :, 3].max() - t0, fin[:, :, 3].max() - t0])
        r = np.median(np.array(rows[1:]), axis=0)
        print("K=%d: kernel %.1f us | last wave enters %.1f | after barrier mean %.1f max %.1f | pilot loop ends mean %.1f max %.1f | "
              "finisher loop ends mean %.1f max %.1f | last pilot exit %.1f | last finisher exit %.1f" % ((K,) + tuple(r)))
    env.close()


if __name__ == "__main__":
    main()
