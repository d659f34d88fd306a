#!/usr/bin/env python3
"""Soak (GPU box): the one-launch-per-step kernels against the K-step rollout kernels at a large batch, bit for bit, over many steps.
Two handles of the same id and seed; chunk by chunk of 20 steps with fresh random actions one runs 20 launches of the step kernel
(every wave walking several subtiles), the other one launch of the wave-pair rollout kernel; every output of every step and, at
the end, the state must be identical.
    python tools/gpu_soak_step_kernels.py [BATCH] [STEPS] [ENV_ID[:steering] ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ["SPACEGYM_STEP_KERNEL"] = "single"
    import numpy as np
    import torch
    import space_gym_amd as sg
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    ids = sys.argv[3:] or ["GoalContinuous3P-v0", "GoalContinuous4P-v0", "GoalContinuous2P-v0", "GoalDiscrete3-v0",
                           "GoalContinuous3P-v0:acceleration", "KeplerCircleOrbit-v0", "KeplerRandomOrbits-v0", "KeplerDiscrete-v0"]
    K = 20
    dev = torch.device("cuda", 0)
    for spec in ids:
        env_id, _, steering = spec.partition(":")
        kw = {"steering": steering} if steering else {}
        a = sg.make_vec(env_id, B, seed=11, max_episode_steps=150, **kw)
        b = sg.make_vec(env_id, B, seed=11, max_episode_steps=150, **kw)
        a.set_unfused_rollout(1)
        assert "pair" not in a.rollout_kernel(K) and "pair" in b.rollout_kernel(K), (a.rollout_kernel(K), b.rollout_kernel(K))
        a.reset_torch(); b.reset_torch()
        D = a.obs_dim
        bufs = [[torch.empty((K, B, D), device=dev), torch.empty((K, B), device=dev), torch.empty((K, B), dtype=torch.uint8, device=dev),
                 torch.empty((K, B), dtype=torch.uint8, device=dev)] for _ in range(2)]
        gen = torch.Generator(device=dev).manual_seed(3)
        bad, finished = 0, 0
        for c in range(steps // K):
            if a.discrete:
                act = torch.randint(0, 6, (K, B), device=dev, generator=gen, dtype=torch.int32)
            else:
                act = torch.rand((K, B, 2), device=dev, generator=gen) * 2 - 1
            a.rollout_torch(act, *bufs[0]); b.rollout_torch(act, *bufs[1])
            for x, y in zip(bufs[0], bufs[1]):
                bad += int((x != y).sum().item()) if x.dtype != torch.float32 else int((x.view(torch.int32) != y.view(torch.int32)).sum().item())
            finished += int(bufs[0][2].sum().item())
        torch.cuda.synchronize()
        a.check_status(); b.check_status()
        sa, sb = a.get_state(), b.get_state()
        for k in ("ship", "planets", "goal", "elapsed"):
            if sa[k] is not None:
                bad += int((np.asarray(sa[k]) != np.asarray(sb[k])).sum())
        print("%-34s B=%d  %d steps = %.0f M env-steps, %d episodes finished: %d differing words" % (
            spec, B, steps // K * K, steps // K * K * B / 1e6, finished, bad), flush=True)
        a.close(); b.close()
        del bufs


if __name__ == "__main__":
    main()
