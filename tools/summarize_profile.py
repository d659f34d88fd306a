#!/usr/bin/env python3
"""Condense a tools/gpu_profile.sh output directory into a small text summary for profiles/ (the raw rocprofv3 CSVs are
large and stay in gpurun_out/), and print the profiles/traffic.json entries that follow from its PMC passes.

    python3 tools/summarize_profile.py DIR STEPS

bench.py's rollout launches under the profiler, in launch order: the untimed pre-roll (2000-step launches), the warm-up steps
(the shortest dispatch), then the timed STEPS-step launch and every later STEPS-step launch (repeats of the timed region and
the passes with dispatch events) -- the "STEPS-step class" below is everything after the shortest dispatch."""
import csv
import glob
import json
import os
import sys


def newest(pattern):
    f = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return f[-1:]  # a directory may hold the files of an earlier run of the same tag


def short(name):
    return name.split("(")[0].replace("void ", "")


def kernel_rows(path, needle):
    f = newest(os.path.join(path, "**", "*_kernel_stats.csv"))
    return [r for r in csv.DictReader(open(f[0])) if needle in r["Name"]] if f else []


def dispatches(path):
    """[(dispatch id, kernel, start, duration_us)] in launch order"""
    f = newest(os.path.join(path, "**", "*_kernel_trace.csv"))
    if not f:
        return []
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
    return [(r.get("Dispatch_Id", ""), short(r["Kernel_Name"]), int(r["Start_Timestamp"]),
             (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]


def counter_rows(path):
    f = newest(os.path.join(path, "**", "*_counter_collection.csv"))
    return list(csv.DictReader(open(f[0]))) if f else []


def k_step_class(rows):
    """rollout-kernel dispatches after the shortest one (the warm-up launch)"""
    roll = [r for r in rows if "rollout_kernel" in r[1]]
    if not roll:
        return []
    roll = [r for r in roll if r[1] == roll[0][1]]  # (the terminal-observation variant of the kernel, bench.py's extra leg, is another kernel)
    i = min(range(len(roll)), key=lambda k: roll[k][3])
    # the timed region, its four repeats, one more pass and the five with dispatch events: what follows are bench.py's extra
    # legs (Kepler's terminal-observation launches are the same kernel)
    return roll[i + 1:i + 12]


def main():
    d, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20
    print(f"# profile summary of {os.path.basename(d.rstrip('/'))}: python3 bench.py --gpus 1 --steps {steps} --warmup ... (tools/gpu_profile.sh)")
    b = None
    try:
        b = json.loads(open(os.path.join(d, "bench.json")).read().strip().splitlines()[-1])
        print("bench.py line (un-profiled run of the same command):", json.dumps(b))
    except Exception as e:  # noqa: BLE001
        print("bench.json unreadable:", e)
    print("\n## rocprofv3 --kernel-trace --stats of the same command (--no-cpu-baseline --no-host-path)")
    for r in kernel_rows(os.path.join(d, "trace"), "_kernel"):
        print({k: (short(r[k]) if k == "Name" else r[k]) for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")})
    rows = dispatches(os.path.join(d, "trace"))
    cls = k_step_class(rows)
    if cls:
        us = [r[3] for r in cls]
        print(f"\n{steps}-step launches of {cls[0][1]} (everything after the warm-up launch): n={len(us)} avg {sum(us) / len(us):.2f} us "
              f"min {min(us):.2f} max {max(us):.2f}; the TIMED launch (first of them): {us[0]:.2f} us; per step {sum(us) / len(us) / steps:.3f} us")
        if b and "roofline" in b:
            print(f"bench.py's own dispatch-event figure for the same kernel: avg {b['roofline']['kernel_avg_us']:.2f} us "
                  f"(frac {b['roofline']['frac']:.4f}); by the rocprofv3 average: frac "
                  f"{b['roofline']['algorithmic_bytes_per_launch'] / (sum(us) / len(us) * 1e-6) / 1e9 / 8000.0:.4f}")
    step = [r[3] for r in rows if "step_kernel" in r[1]]
    if step:
        print(f"one-launch-per-step kernel ({[r[1] for r in rows if 'step_kernel' in r[1]][0]}): n={len(step)} avg {sum(step) / len(step):.2f} us "
              f"min {min(step):.2f} max {max(step):.2f}")
    print("\n## PMC, separate passes (FETCH_SIZE / WRITE_SIZE in KB per dispatch; gfx950: FETCH_SIZE counts 128-B requests at 64 B")
    print("## for 16-B-per-lane coalesced loads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact)")
    traffic = {}
    for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        rows_c = counter_rows(os.path.join(d, sub))
        disp = dispatches(os.path.join(d, sub))
        cls_ids = {r[0] for r in k_step_class(disp)}
        by = {}
        for r in rows_c:
            if r["Counter_Name"] != ctr:
                continue
            k = short(r["Kernel_Name"])
            if "rollout_kernel" in k and r.get("Dispatch_Id", "") not in cls_ids:
                continue  # pre-roll and warm-up launches
            by.setdefault(k, []).append(float(r["Counter_Value"]))
        for k, v in by.items():
            print(f"{sub}: {k}: {ctr}: mean {sum(v) / len(v):.1f} KB max {max(v):.1f} (n={len(v)})")
            traffic.setdefault(k, {})[ctr] = sum(v) / len(v)
    for kern, cs in counters_sq(os.path.join(d, "pmc_sq")).items():
        for k, v in cs.items():
            print(f"pmc_sq: {kern}: {k}: mean {sum(v) / len(v):.1f} (n={len(v)})")
    if b:
        meas = []
        for k, t in traffic.items():
            if "FETCH_SIZE" in t and "WRITE_SIZE" in t and ("rollout_kernel" in k or "step_kernel" in k):
                spl = steps if "rollout_kernel" in k else 1
                meas.append({"env_id": b["config"]["env_id"], "batch": b["config"]["batch_per_gpu"], "kernel": k, "steps_per_launch": spl,
                             "FETCH_SIZE_KB_per_launch": t["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": t["WRITE_SIZE"],
                             "hbm_bytes_per_launch": (2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024})
        print("\n## profiles/traffic.json entries from this run")
        print(json.dumps(meas, indent=1))


def counters_sq(path):
    out = {}
    for r in counter_rows(path):
        k = short(r["Kernel_Name"])
        if "rollout_kernel" in k or "step_kernel" in k:
            out.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return out


if __name__ == "__main__":
    main()
