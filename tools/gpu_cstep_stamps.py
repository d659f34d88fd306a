#!/usr/bin/env python3
"""Diagnostic (stamped build, -DSG_STAMPS; GPU box): where the waves of goal_step_kernel spend their cycles.
    SPACEGYM_LIB=space_gym_amd/lib/libspacegym_hip_stamps.so python tools/gpu_cstep_stamps.py [batch]
Slots: 0 entry, 1 loads issued + barrier, 2 first step done, 3 owner stores + lists, 4 past barrier 1, 8 replay chunks done,
9 restart passes done, 10 resamples done, 11 past barrier 2, 5 exit; 6 / 7 wall clock at entry / exit."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import space_gym_amd as sg  # noqa: E402
from space_gym_amd import _native  # noqa: E402

SLOTS, WAVES = 16, 4096


def main():
    os.environ["SPACEGYM_STEP_KERNEL"] = "single"
    lib = _native.load()
    lib.sg_debug_read_stamps.argtypes = [C.c_void_p, C.c_int64]
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    env = sg.make_vec(sys.argv[2] if len(sys.argv) > 2 else "GoalContinuous3P-v0", B, seed=0)
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
    st = buf.reshape(WAVES, SLOTS)[:min(WAVES, B // 64, 2048)].astype(np.float64)
    w = np.arange(len(st)) % 4
    names = [("top of pass -> inputs here", 13, 14), ("previous stores + next loads issued", 14, 1), ("first step (begin + fast step)", 1, 2),
             ("reward + observation", 2, 15), ("records + lists (end of pass)", 15, 3), ("whole pass (last subtile)", 13, 3),
             ("barrier 1 (wait)", 3, 4), ("replay chunks", 4, 8), ("restart passes", 8, 9), ("resamples", 9, 10), ("exit", 10, 5), ("whole wave", 0, 5)]
    print("cycles per phase, by wave index within the workgroup (mean / p95 / max); %d waves stamped" % len(st))
    for name, a, b in names:
        d = st[:, b] - st[:, a]
        print("%-36s" % name + "".join("  w%d %7.0f %7.0f %7.0f" % (k, d[w == k].mean(), np.percentile(d[w == k], 95), d[w == k].max()) for k in range(4)))
    real = st[:, 7] - st[:, 6]
    print("clock GHz %.2f; wave lifetime us mean %.2f max %.2f; span first entry -> last exit %.2f us; entry skew %.2f us" % (
        (st[:, 5] - st[:, 0]).sum() / (real.sum() * 10.0), real.mean() / 100, real.max() / 100, (st[:, 7].max() - st[:, 6].min()) / 100,
        (st[:, 6].max() - st[:, 6].min()) / 100))
    # per-phase cycles summed over ALL passes of a wave (rows 2048..): inputs wait, stores + loads issued, first step, reward + observation, records + lists, loop back
    if len(st) <= 2048:
        acc = buf.reshape(WAVES, SLOTS)[2048:2048 + len(st), :8].astype(np.float64)
        n_pass = max(1.0, (B / 64) / len(st))
        names2 = ["inputs wait", "stores + next loads issued", "first step", "reward + observation", "records + lists", "loop back"]
        print("mean cycles per pass over all passes (%.1f passes per wave): " % n_pass + ", ".join("%s %.0f" % (n, acc[:, k].mean() / n_pass) for k, n in enumerate(names2)) +
              "; sum %.0f" % (acc[:, :6].sum(1).mean() / n_pass))
    # how long the workgroups lived (slot 11: records | restarts << 16 | resamples << 32 the wave had gathered when it reached the barrier)
    val = buf.reshape(WAVES, SLOTS)[:len(st), 11]
    n_wg = len(st) // 4
    life = (st[:, 7].reshape(n_wg, 4).max(1) - st[:, 6].reshape(n_wg, 4).min(1)) / 100
    passes = (st[:, 3].reshape(n_wg, 4).max(1) - st[:, 0].reshape(n_wg, 4).min(1))  # entry -> last wave through with its subtiles
    tail = (st[:, 5].reshape(n_wg, 4).max(1) - st[:, 3].reshape(n_wg, 4).max(1))
    recs = (val & np.uint64(0xffff)).astype(int).reshape(n_wg, 4).sum(1)
    rst = ((val >> np.uint64(16)) & np.uint64(0xffff)).astype(int).reshape(n_wg, 4).sum(1)
    print("workgroups: lifetime us mean %.2f p50 %.2f p95 %.2f max %.2f; passes (entry -> last wave done) cycles mean %.0f p95 %.0f max %.0f; tail cycles mean %.0f p95 %.0f max %.0f" % (
        life.mean(), np.median(life), np.percentile(life, 95), life.max(), passes.mean(), np.percentile(passes, 95), passes.max(), tail.mean(), np.percentile(tail, 95), tail.max()))
    print("records per workgroup mean %.1f max %d; restarts mean %.1f max %d; corr(lifetime, records) %.2f corr(lifetime, passes) %.2f corr(lifetime, tail) %.2f" % (
        recs.mean(), recs.max(), rst.mean(), rst.max(), np.corrcoef(life, recs)[0, 1], np.corrcoef(life, passes)[0, 1], np.corrcoef(life, tail)[0, 1]))
    env.close()


if __name__ == "__main__":
    main()
