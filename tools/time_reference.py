#!/usr/bin/env python3
"""Times the UNMODIFIED reference's CPU step (SpaceshipEnv.step, gym_space/envs/spaceship_env.py:68-78 ->
dynamic_model.make_step :94-125 -> scipy RK45) on this container's cores: the workload of BASELINE.md §2 / SURVEY §8d(ii).

Runs only where /root/reference exists (the build container; the reference never travels).  `gym` is not installed, so the
env layer is imported with the loader-only namespace of tools/_gym_loader_shim.py, exactly as tools/gen_golden.py does; no
arithmetic on the step path comes from it.

    python tools/time_reference.py [--steps 3000] [--warmup 200] [--procs 1,8] [--ids GoalContinuous3P-v0,...]

Per config: one env per process, random U(-1,1) actions, reset on done or after 500 steps (gym TimeLimit), `steps` timed
steps after `warmup`; P processes run concurrently and the aggregate rate is the sum of their rates.  Prints a markdown
table and writes profiles/reference_cpu_timing.json."""
import argparse
import contextlib
import io
import json
import multiprocessing as mp
import os
import platform
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
MAX_EPISODE_STEPS = 500


def worker(env_id, steps, warmup, seed, out):
    sys.path.insert(0, HERE)
    import numpy as np
    import _gym_loader_shim
    _gym_loader_shim.install()
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):  # the constructors print their configuration
        import gym_space  # noqa: F401  (runs the register() calls)
        import gym_space.envs as envs
        spec = _gym_loader_shim.REGISTRY[env_id]
        env = getattr(envs, spec["entry_point"].split(":")[1])(**spec["kwargs"])
    rng = np.random.default_rng(seed)
    env.seed(seed)
    env.reset()
    elapsed = n_done = 0

    def run(n):
        nonlocal elapsed, n_done
        for _ in range(n):
            _, _, done, _ = env.step(rng.uniform(-1, 1, 2).astype(np.float32))
            elapsed += 1
            if done or elapsed >= MAX_EPISODE_STEPS:
                env.reset()
                elapsed = 0
                n_done += 1
    run(warmup)
    n_done = 0
    t0 = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t0
    out.put((steps / dt, n_done))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--procs", default=f"1,{os.cpu_count()}")
    ap.add_argument("--ids", default="GoalContinuous2P-v0,GoalContinuous3P-v0,GoalContinuous4P-v0,KeplerCircleOrbit-v0")
    args = ap.parse_args()
    if not os.path.isdir(REF):
        raise SystemExit(f"{REF} not found: the reference is only present in the build container")
    import numpy
    import scipy
    cpu = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), platform.processor())
    rows = []
    for env_id in args.ids.split(","):
        for p in [int(x) for x in args.procs.split(",")]:
            q = mp.Queue()
            ps = [mp.Process(target=worker, args=(env_id, args.steps, args.warmup, 100 + k, q)) for k in range(p)]
            for x in ps:
                x.start()
            res = [q.get() for _ in ps]
            for x in ps:
                x.join()
            rate = sum(r for r, _ in res)
            rows.append(dict(env_id=env_id, processes=p, env_steps_per_s=rate, us_per_step_per_process=1e6 * p / rate,
                             episodes_finished_per_1000_steps=1000.0 * sum(d for _, d in res) / (p * args.steps)))
            print(f"{env_id:24s} P={p:2d}  {rate:9.0f} env-steps/s  {1e6 * p / rate:7.0f} us/step/process", flush=True)
    host = dict(cpu=cpu, logical_cpus=os.cpu_count(), python=platform.python_version(), numpy=numpy.__version__, scipy=scipy.__version__,
                steps=args.steps, warmup=args.warmup)
    out = dict(what="unmodified reference CPU step, one env per process (tools/time_reference.py)", host=host, rows=rows)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", "reference_cpu_timing.json"), "w"), indent=1)
    print(f"\nHost: {cpu}, {os.cpu_count()} logical CPUs, Python {host['python']}, numpy {host['numpy']}, scipy {host['scipy']}\n")
    print("| config | processes | env-steps/s | µs/step/process |\n|---|---|---|---|")
    for r in rows:
        print(f"| {r['env_id']} | {r['processes']} | {r['env_steps_per_s']:,.0f}{' (aggregate)' if r['processes'] > 1 else ''} | {r['us_per_step_per_process']:.0f} |")


if __name__ == "__main__":
    main()
