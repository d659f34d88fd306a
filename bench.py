#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the batched step() at batch = 65536 per GPU on GoalContinuous3P-v0
(BASELINE.json `metric`, configs[2]); one process per GPU, weak scaling, no data-path collective.

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python bench.py --gpus N ...            (spawns N ranks itself through torch.distributed.run, relays rank 0's line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over the whole per-GPU batch (integrate + events + observation + reward + goal
resample + TimeLimit + auto-reset for every env).  Inputs (U(-1,1) action blocks) are resident in HBM before the timed
region; outputs go to a [steps, B, ...] rollout buffer in HBM.  The timed region is ONE sg_rollout_device call for the K
steps: for the Goal ids that is one launch of the K-step rollout kernel (env state in registers across steps); the same K
steps as K launches of the per-step kernel (what a policy-in-the-loop user gets) are timed too and reported beside it with
their own roofline block.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_env_step(env_id):
    """fp32 SoA, each live field moved once; reset / goal-resample traffic excluded."""
    from space_gym_amd.registration import ENV_SPECS
    s = ENV_SPECS[env_id]
    return 113 + 16 * s["n_planets"] if s["family"] == "goal" else 109


def measured_traffic(env_id, batch, kernel_name, steps_per_launch):
    """HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 for 16 B/lane streams, WRITE_SIZE), taken
    offline with tools/gpu_profile.sh and committed under profiles/; null unless a measurement of THIS kernel on this
    workload exists (the per-step part scales with the steps per launch)."""
    try:
        ms = [t for t in json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["measurements"]
              if t["env_id"] == env_id and t["batch"] == batch and t["kernel"] == kernel_name]
        if ms:  # the measurement taken at these steps per launch, else the nearest one scaled (a launch's fixed part is small)
            t = min(ms, key=lambda t: abs(math.log(t["steps_per_launch"] / max(1, steps_per_launch))))
            return t["hbm_bytes_per_launch"] * steps_per_launch / t["steps_per_launch"]
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return None


def obs_dim_of(env_id):
    from space_gym_amd.registration import obs_dim
    return obs_dim(env_id)


def usable_cores():
    """CPU share of this process: the cgroup quota when there is one (a 1-GPU box gets 16 of the host's cores),
    else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 64)


def cpu_baseline(env_id, batch, budget_s=12.0):
    """The fp64 CPU oracle (oracle/, a restatement of the reference's NumPy/scipy path pinned to its golden vectors)
    on the same workload on this box's host cores, bounded sample: all usable cores (`value`) and one core
    (`single_core`).  Reported beside the GPU number, never as it."""
    import numpy as np
    from oracle import Oracle
    rng = np.random.default_rng(1)

    def sample(threads, n, budget):
        o = Oracle(env_id, threads=threads)
        envs, _ = o.vec_reset(n, seed=0)
        acts = [rng.uniform(-1, 1, size=(n, 2)).astype(np.float32) for _ in range(8)]
        for i in range(2):  # warm-up
            o.vec_step(envs, acts[i], seed=0)
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < budget:
            o.vec_step(envs, acts[steps % 8], seed=0)
            steps += 1
        dt = time.perf_counter() - t0
        return n * steps / dt, f"{steps} vector steps of {n} envs ({env_id}, random U(-1,1) actions, auto-reset on), {dt:.1f} s"

    cores = usable_cores()
    v_all, s_all = sample(cores, min(batch, 65536), budget_s * 0.65)
    v_one, s_one = sample(1, min(batch, 4096), budget_s * 0.35)
    return {"value": v_all, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": s_all + f", OpenMP over {cores} threads, fp64 RK45 oracle",
            "single_core": {"value": v_one, "cores": 1, "sample": s_one + ", one thread"}}


def spawn_ranks(n):
    """`python bench.py --gpus N` as a plain command: start N fresh ranks through torch.distributed.run BEFORE anything in
    this process touches the GPU, relay their output (rank 0 prints the JSON line) and leave with their exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # RCCL's intra-node transport shares device buffers between the ranks' processes; this pool's host driver only supports
    # dmabuf IPC handles, and with the legacy mode RCCL's set-up fails with `hipIpcGetMemHandle: invalid argument` (the image
    # exports the variable already; kept for a shell that does not)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--env", default="GoalContinuous3P-v0")
    ap.add_argument("--batch", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--steering", choices=["velocity", "acceleration"], default="velocity",
                    help="ship_steering of the reference's constructor: 1 (velocity, every registered id) or 0 (acceleration)")
    ap.add_argument("--action-ring", type=int, default=0, help="distinct pre-generated action blocks (0: one per step)")
    ap.add_argument("--action-source", choices=["torch", "engine"], default="torch",
                    help="who draws the U(-1,1) action tape: torch.rand (per-rank generator) or sg_random_actions_device")
    ap.add_argument("--repeats", type=int, default=5, help="the timed K-step region is run this many times (SURVEY 8d); `value` is "
                    "the median region, every region is in ms_per_step_repeats, the first one also as value_first_region")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall budget of the cpu_baseline sample")
    ap.add_argument("--preroll", type=int, default=12000,
                    help="untimed steps run before the warm-up steps, as part of the set-up: ~40 ms of the same kernel, so that "
                         "the GPU has left its idle clocks and the envs have their stationary mix of episode ages before "
                         "anything is measured (a 100-step warm-up alone is 0.3 ms)")
    ap.add_argument("--chunk", type=int, default=2000, help="max steps per sg_rollout_device call / rollout buffer")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the leg that also gathers every step's (obs, reward, done, "
                    "truncated) to rank 0 (the single-process VectorEnv view: the only exchange the path has)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the NumPy host-buffer path sample (sg_step, PCIe-inclusive)")
    ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph replay of the one-launch-per-step chain")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    import space_gym_amd as sg

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    # Rehearsal aid: SG_BENCH_REHEARSE=1 runs all ranks on GPU 0 with the gloo backend, to exercise the multi-rank logic on a
    # one-GPU box (RCCL refuses two ranks on one device).  Never set by the driver.
    rehearse = os.environ.get("SG_BENCH_REHEARSE") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    B, K, W = args.batch, args.steps, args.warmup
    env = sg.make_vec(args.env, B, device=dev_index, seed=args.seed, env_index_base=rank * B, copy=False, terminal_observation=False, validate_actions=False,
                      steering=args.steering)
    D = env.obs_dim
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)
    # rollout buffers hold at most `chunk` steps (2 000 steps of 65 536 envs = 7.9 GB of observations); longer runs reuse them
    chunk = max(1, min(max(K, W, 1), args.chunk, max(1, int(24e9 // (B * (obs_dim_of(args.env) * 4 + 14))))))
    ring = max(1, min(args.action_ring if args.action_ring > 0 else chunk, chunk))
    if args.action_source == "engine":  # sg_random_actions_device: Philox keyed by (seed, global env index, step)
        actions = env.random_actions_torch(ring, seed=1)
    elif env.discrete:
        actions = torch.randint(0, 6, (ring, B), generator=gen, device=dev, dtype=torch.int32)
    else:
        actions = torch.rand((ring, B, 2), generator=gen, device=dev, dtype=torch.float32) * 2 - 1
    nbuf = chunk
    # rollout buffers in HBM (3.9 GB of observations at K=1000, B=65536, D=15)
    act_seq = actions.repeat((nbuf + ring - 1) // ring, *([1] * (actions.dim() - 1)))[:nbuf].contiguous()
    obs = torch.empty((nbuf, B, D), device=dev, dtype=torch.float32)
    rew = torch.empty((nbuf, B), device=dev, dtype=torch.float32)
    done = torch.empty((nbuf, B), device=dev, dtype=torch.uint8)
    trunc = torch.empty((nbuf, B), device=dev, dtype=torch.uint8)

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            if rehearse:
                dist.barrier()
            else:
                dist.barrier(device_ids=[dev_index])
            torch.cuda.synchronize(dev)

    prepared = {}

    def run_steps(n):
        """n steps as ceil(n / chunk) sg_rollout_device calls into the (reused) rollout buffers; the buffers of a call are
        validated once (prepare_rollout), the call itself is one C function"""
        left = n
        while left > 0:
            k = min(left, chunk)
            if k not in prepared:
                prepared[k] = env.prepare_rollout(act_seq[:k], obs[:k], rew[:k], done[:k], trunc[:k])
            prepared[k]()
            left -= k

    def prepare_steps(n):  # build (and validate) the calls that run_steps(n) will make, outside any timed region
        left = n
        while left > 0:
            k = min(left, chunk)
            if k not in prepared:
                prepared[k] = env.prepare_rollout(act_seq[:k], obs[:k], rew[:k], done[:k], trunc[:k])
            left -= k

    for n in (args.preroll, W, K, min(K, 50)):
        prepare_steps(n)
    env.reset_torch()
    run_steps(args.preroll)  # set-up: brings the envs to their stationary mix of episode ages and the GPU to its working clocks
    sync_all()
    run_steps(W)
    sync_all()

    # ---- timed region: exactly K steps, nothing else between the two synchronisations (an event pair recorded around the
    # launch cost 10 us of host time, 13 % of a 20-step region)
    timing = not args.no_kernel_timing
    sync_all()
    t0 = time.perf_counter()
    run_steps(K)
    sync_all()
    dt = time.perf_counter() - t0
    env.check_status()  # a rollout whose wave hand-off timed out is not a measurement
    # ---- the same K-step region repeated (SURVEY 8d asks for >= 5 repeats and their median: that is `value`)
    repeats = [dt]
    for _ in range(max(0, args.repeats - 1)):
        sync_all()
        tr = time.perf_counter()
        run_steps(K)
        sync_all()
        repeats.append(time.perf_counter() - tr)

    # ---- the same K steps again with start/stop events on every dispatch (hipExtLaunchKernelGGL): the kernel's own
    # duration, as rocprofv3 --kernel-trace reports it.  Straight after the timed regions, on the same clocks (after the
    # one-launch-per-step passes below the GPU needs ~20 ms of rollout launches to come back to them)
    launches, kern_ms, kmin, kmax, dt_events = 0, 0.0, 0.0, 0.0, None
    kernel_name = env.rollout_kernel(min(K, chunk))
    if timing:
        run_steps(K)
        env.set_profiling(True)
        sync_all()
        t1 = time.perf_counter()
        for _ in range(max(1, args.repeats)):
            run_steps(K)
        sync_all()
        dt_events = (time.perf_counter() - t1) / max(1, args.repeats)
        launches, kern_ms, kmin, kmax = env.get_profile()
        env.set_profiling(False)
    env.check_status()

    # ---- the same K-step launch returning what the reference's step() returns with done=True as well: the LAST observation of
    # every episode that ends (spaceship_env.py:75-78; sg_rollout_device_terminal, one record per finished env-step).  Extra keys,
    # never `value`.
    tobs_ms = tobs_kernel_us = tobs_records = None
    if world == 1 and min(K, chunk) == K:
        cap = int(B * K * 0.05) + 4096
        tl = env.terminal_list_torch(cap)
        for _ in range(2):
            env.rollout_torch(act_seq[:K], obs[:K], rew[:K], done[:K], trunc[:K], terminal=tl)
        sync_all()
        tt = time.perf_counter()
        for _ in range(max(1, args.repeats)):
            env.rollout_torch(act_seq[:K], obs[:K], rew[:K], done[:K], trunc[:K], terminal=tl)
        sync_all()
        tobs_ms = (time.perf_counter() - tt) * 1e3 / max(1, args.repeats)
        if timing:
            env.set_profiling(True)
            for _ in range(max(1, args.repeats)):
                env.rollout_torch(act_seq[:K], obs[:K], rew[:K], done[:K], trunc[:K], terminal=tl)
            sync_all()
            n_l, t_ms, _, _ = env.get_profile()
            env.set_profiling(False)
            tobs_kernel_us = t_ms * 1e3 / max(1, n_l)
        tobs_records = int(tl["count"].item())
        env.check_status()

    # ---- A/B: the same K steps as K launches of the per-step kernel (what a policy-in-the-loop user gets)
    env.set_unfused_rollout(True)
    step_kernel_name = env.rollout_kernel(1)
    run_steps(min(K, 50))
    sync_all()
    t_u = time.perf_counter()
    run_steps(K)
    sync_all()
    dt_unfused = time.perf_counter() - t_u
    u_launches, u_ms, u_min, u_max = 0, 0.0, 0.0, 0.0
    if timing:  # the per-step kernel's own duration (dispatch start/stop events)
        env.set_profiling(True)
        sync_all()
        run_steps(K)
        sync_all()
        u_launches, u_ms, u_min, u_max = env.get_profile()
        env.set_profiling(False)
    # ---- the same chain of step-kernel launches captured once into a hipGraph and replayed (what a policy-in-the-loop user does
    # with the policy's kernels in between): launch entry and completion of neighbouring kernels overlap as far as the hardware lets them
    graph_ms = None
    if world == 1 and not args.no_graph:
        Kg = min(K, 50)
        g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                env.rollout_torch(act_seq[:Kg], obs[:Kg], rew[:Kg], done[:Kg], trunc[:Kg])
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(3):
            g.replay()
        sync_all()
        t_g = time.perf_counter()
        n_rep = max(1, 200 // Kg)
        for _ in range(n_rep):
            g.replay()
        sync_all()
        graph_ms = (time.perf_counter() - t_g) * 1e3 / (n_rep * Kg)
        del g
    env.set_unfused_rollout(False)

    # ---- N > 1: the same steps with the only exchange the path has -- every rank's (obs, reward, done, truncated) of each step
    # received into rank 0's [N * B, ...] arrays, one batch of point-to-point transfers per step (space_gym_amd/sharded.py)
    gather_ms = None
    if world > 1 and not args.no_gather:
        Kg = min(K, 50)
        xdev = torch.device("cpu") if rehearse else dev  # (gloo moves host tensors; RCCL device tensors)
        root = None
        if rank == 0:
            root = [torch.empty((world * B, D), dtype=torch.float32, device=xdev), torch.empty(world * B, dtype=torch.float32, device=xdev),
                    torch.empty(world * B, dtype=torch.uint8, device=xdev), torch.empty(world * B, dtype=torch.uint8, device=xdev)]

        def step_and_gather(j):
            env.rollout_torch(act_seq[j:j + 1], obs[j:j + 1], rew[j:j + 1], done[j:j + 1], trunc[j:j + 1])
            mine = [x[j].to(xdev) for x in (obs, rew, done, trunc)]
            if rank == 0:
                for f, x in zip(root, mine):
                    f[:B].copy_(x)
                ops = [dist.P2POp(dist.irecv, f[r * B:(r + 1) * B], r) for r in range(1, world) for f in root]
            else:
                ops = [dist.P2POp(dist.isend, x, 0) for x in mine]
            for q in dist.batch_isend_irecv(ops):
                q.wait()

        step_and_gather(0)
        sync_all()
        t2 = time.perf_counter()
        for t in range(Kg):
            step_and_gather(t % chunk)
        sync_all()
        gather_ms = (time.perf_counter() - t2) * 1e3 / Kg

    host_us = host_async_us = None
    if world == 1 and not args.no_host_path:  # NumPy in / out: H2D + kernel + D2H per step (PCIe-inclusive)
        a_host = act_seq[0].cpu().numpy()
        for _ in range(3):
            env.step(a_host)
        n_host, t_async = 20, 0.0
        th = time.perf_counter()
        for _ in range(n_host):
            ta = time.perf_counter()
            env.step_async(a_host)  # everything enqueued: the host is free from here ...
            t_async += time.perf_counter() - ta
            env.step_wait()         # ... to here
        host_us = (time.perf_counter() - th) * 1e6 / n_host
        host_async_us = t_async * 1e6 / n_host

    red_dev = torch.device("cpu") if rehearse else dev
    dt_median = sorted(repeats)[len(repeats) // 2]
    stats = torch.tensor([dt, dt_unfused, dt_events or 0.0, gather_ms or 0.0, dt_median], device=red_dev, dtype=torch.float64)
    k_last = K - (K - 1) // chunk * chunk  # steps in the last chunk, whose outputs are still in the buffers
    n_done = (done[:k_last].sum(dtype=torch.float64) * (K / k_last)).reshape(1).to(red_dev)
    ranks_info = None
    if world > 1:
        # what every rank saw: its clock for the timed region, its device -- so that the line shows N ranks on N devices
        mine = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "device": torch.cuda.get_device_name(dev_index),
                "ms_per_step": sorted(repeats)[len(repeats) // 2] * 1e3 / K, "ms_per_step_first_region": dt * 1e3 / K,
                "kernel_avg_us": (kern_ms * 1e3 / launches) if launches else None}
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, mine)
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        dist.all_reduce(n_done, op=dist.ReduceOp.SUM)
    dt_first, dt_unfused, dt_events = float(stats[0]), float(stats[1]), (float(stats[2]) or None)
    gather_ms = float(stats[3]) or None
    dt_max = float(stats[4])  # max over ranks of each rank's median region

    if rank == 0:
        bytes_per = algorithmic_bytes_per_env_step(args.env)
        out = {
            "metric": f"env-steps/sec at batch={B}, {args.env}, 1/2/4/8 MI355X",
            "value": world * B * K / dt_max, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt_max * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.env}, batch={B} per GPU, i.i.d. U(-1,1) actions resident in HBM "
                                   f"({ring} distinct blocks), auto-reset on (termination or 500-step truncation), "
                                   f"the {K} steps in {-(-K // chunk)} sg_rollout_device call(s) (one launch of "
                                   f"{kernel_name} each, env state in registers), outputs to a [steps, B, ...] rollout buffer in HBM",
                       "env_id": args.env, "steering": args.steering, "batch_per_gpu": B, "global_batch": world * B, "obs_dim": D,
                       "parallelism": f"env-sharded x{world}, no data-path collective",
                       "episodes_finished_per_step": float(n_done.item()) / K, "preroll_steps": args.preroll},
        }
        if timing and launches:
            avg_us = kern_ms * 1e3 / launches
            steps_per_launch = K * max(1, args.repeats) / launches
            achieved = steps_per_launch * B * bytes_per / (avg_us * 1e-6) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS,
                               "traffic": measured_traffic(args.env, B, kernel_name, steps_per_launch),
                               "kernel": kernel_name,
                               "env_steps_per_launch": steps_per_launch * B, "steps_per_launch": steps_per_launch,
                               "kernel_avg_us": avg_us, "kernel_min_us": kmin * 1e3, "kernel_max_us": kmax * 1e3,
                               "launches": launches, "timing": "hipExtLaunchKernelGGL start/stop events on each of the "
                               f"{launches} dispatches of {max(1, args.repeats)} further, identical {K}-step passes",
                               "algorithmic_bytes_per_env_step": bytes_per,
                               "algorithmic_bytes_per_launch": steps_per_launch * B * bytes_per}
            out["value_with_dispatch_events"] = world * B * K / dt_events
        out["value_is"] = f"median of the {len(repeats)} timed regions (max over ranks); value_first_region is the first one"
        out["value_first_region"] = world * B * K / dt_first
        out["ms_per_step_first_region"] = dt_first * 1e3 / K
        out["ms_per_step_repeats"] = [r * 1e3 / K for r in repeats]  # rank 0's clock
        out["ms_per_step_median"] = sorted(out["ms_per_step_repeats"])[len(repeats) // 2]
        if tobs_ms is not None:  # the same launch with terminal observations (TOBS kernel variant)
            out["with_terminal_observations"] = {
                "ms_per_step": tobs_ms / K, "value": B * K / (tobs_ms * 1e-3), "unit": "env-steps/s",
                "kernel": kernel_name.replace("false>", "true>") if "pair_rollout" in kernel_name and kernel_name.startswith("goal") else kernel_name,
                "kernel_avg_us": tobs_kernel_us, "records_per_launch": tobs_records,
                "what": "sg_rollout_device_terminal: the same K steps in one launch, plus one record (step, env, last observation) per "
                        "finished env-step -- what SpaceshipEnv.step returns with done=True (spaceship_env.py:75-78)"}
        out["value_one_launch_per_step"] = world * B * K / dt_unfused
        out["ms_per_step_one_launch_per_step"] = dt_unfused * 1e3 / K
        if graph_ms is not None:  # the same launches replayed from one hipGraph (wall per step)
            out["ms_per_step_one_launch_per_step_hipgraph"] = graph_ms
        if timing and u_launches:
            u_avg = u_ms * 1e3 / u_launches
            u_ach = B * bytes_per / (u_avg * 1e-6) / 1e9
            out["roofline_one_launch_per_step"] = {
                "bound": "hbm", "achieved": u_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": u_ach / HBM_PEAK_GBS,
                "traffic": measured_traffic(args.env, B, step_kernel_name, 1), "kernel": step_kernel_name,
                "kernel_avg_us": u_avg, "kernel_min_us": u_min * 1e3, "kernel_max_us": u_max * 1e3, "launches": u_launches}
        if host_us is not None:
            out["host_numpy_path"] = {"us_per_step": host_us, "us_in_step_async": host_async_us, "value": B / (host_us * 1e-6),
                                      "unit": "env-steps/s",
                                      "what": "step_async + step_wait (sg_step_begin / sg_step_end) with NumPy arrays over page-locked "
                                              "memory, no copies on the host side, no terminal observations: the step kernel reads the actions "
                                              "from and stores its outputs into page-locked host memory itself, enqueued by step_async "
                                              "(us_in_step_async of host time; the host is free until step_wait); PCIe-inclusive, never `value`"}
        if world > 1:
            out["distributed"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                  "devices_visible": torch.cuda.device_count(), "ranks": ranks_info,
                                  "distinct_devices": len({r["device_index"] for r in ranks_info})}
        if gather_ms is not None:
            out["ms_per_step_with_rccl_gather"] = gather_ms  # max over ranks; one launch per step + the gather to rank 0
            out["value_with_rccl_gather"] = world * B / (gather_ms * 1e-3)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.env, B, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
