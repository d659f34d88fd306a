#!/usr/bin/env python3
"""PCIe-inclusive rate of the NumPy path (sg_step: H2D actions + kernel + D2H obs/reward/done per call)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import space_gym_amd as sg

for env_id, B, copy in (("GoalContinuous3P-v0", 65536, True), ("GoalContinuous3P-v0", 65536, False),
                        ("GoalContinuous2P-v0", 4096, False), ("KeplerCircleOrbit-v0", 65536, False)):
    env = sg.make_vec(env_id, B, seed=0, terminal_observation=False, copy=copy)
    env.reset()
    rng = np.random.default_rng(0)
    acts = [rng.uniform(-1, 1, size=(B, 2)).astype(np.float32) for _ in range(16)]
    for i in range(20):
        env.step(acts[i % 16])
    t0 = time.perf_counter(); K = 300
    for i in range(K):
        env.step(acts[i % 16])
    dt = time.perf_counter() - t0
    print(f"{env_id} B={B} copy={copy}: NumPy path {dt / K * 1e6:.1f} us/step = {B * K / dt / 1e6:.1f} M env-steps/s "
          f"(obs {B * env.obs_dim * 4 / 1e6:.2f} MB D2H per step)")
    env.close()
