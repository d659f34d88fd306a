#!/usr/bin/env python3
"""Condense a tools/gpu_profile.sh output directory into a small text summary for profiles/ (the raw rocprofv3 CSVs
are large and stay in gpurun_out/)."""
import csv
import glob
import json
import os
import sys


def newest(pattern):
    f = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return f[-1:]  # a directory may hold the files of an earlier run of the same tag


def kernel_rows(path, needle):
    f = newest(os.path.join(path, "**", "*_kernel_stats.csv"))
    return [r for r in csv.DictReader(open(f[0])) if needle in r["Name"]] if f else []


def dispatch_durations(path, needle):
    """[(kernel, duration_us)] of every dispatch whose name contains `needle`, in launch order"""
    f = newest(os.path.join(path, "**", "*_kernel_trace.csv"))
    if not f:
        return []
    rows = [r for r in csv.DictReader(open(f[0])) if needle in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]


def short(name):
    return name.split("(")[0].replace("void ", "")


def counters_by_kernel(path, needle):
    """{kernel: {counter: [values per dispatch, in dispatch order]}}"""
    f = newest(os.path.join(path, "**", "*_counter_collection.csv"))
    out = {}
    if not f:
        return out
    for r in csv.DictReader(open(f[0])):
        if needle in r["Kernel_Name"]:
            out.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return out


def main():
    d, needle = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "goal_")
    print(f"# profile summary of {os.path.basename(d.rstrip('/'))}; kernel filter: {needle}")
    try:
        b = json.loads(open(os.path.join(d, "bench.json")).read().strip().splitlines()[-1])
        print("bench.py line:", json.dumps(b))
    except Exception as e:  # noqa: BLE001
        print("bench.json unreadable:", e)
    print("\n## rocprofv3 --kernel-trace --stats (python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-kernel-timing)")
    print("## (rollout kernel launches, in order: bench.py's untimed pre-roll in 1000-step launches, the 100 warm-up steps (the short")
    print("##  dispatch), the 1000 TIMED steps = the dispatch right after it, then bench.py's repeats of the same region; the step")
    print("##  kernel's 1000 calls are the one-launch-per-step A/B pass)")
    for r in kernel_rows(os.path.join(d, "trace"), needle):
        print({k: (short(r[k]) if k == "Name" else r[k]) for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")})
    roll = [(k, us) for k, us in dispatch_durations(os.path.join(d, "trace"), "rollout_kernel")]
    if roll:
        short_i = min(range(len(roll)), key=lambda i: roll[i][1])  # the warm-up launch
        timed = roll[short_i + 1] if short_i + 1 < len(roll) else roll[-1]
        print("rollout dispatches in launch order, us:", ", ".join(f"{us:.1f}" for _, us in roll), f"  -> timed launch: {timed[1]:.1f} us ({timed[0]})")
    print("\n## PMC, separate passes; per kernel: mean per dispatch and the largest dispatch (rollout kernel: a 1000-step launch)")
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        for kern, cs in counters_by_kernel(os.path.join(d, sub), needle).items():
            for k, v in cs.items():
                print(f"{sub}: {kern}: {k}: mean {sum(v) / len(v):.1f} max {max(v):.1f} (n={len(v)})")


if __name__ == "__main__":
    main()
