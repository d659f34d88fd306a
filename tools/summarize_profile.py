#!/usr/bin/env python3
"""Condense a tools/gpu_profile.sh output directory into a small text summary for profiles/ (the raw rocprofv3 CSVs
are large and stay in gpurun_out/)."""
import csv
import glob
import json
import os
import sys


def kernel_rows(path, needle):
    f = glob.glob(os.path.join(path, "**", "*_kernel_stats.csv"), recursive=True)
    return [r for r in csv.DictReader(open(f[0])) if needle in r["Name"]] if f else []


def short(name):
    return name.split("(")[0].replace("void ", "")


def counters_by_kernel(path, needle):
    """{kernel: {counter: [values per dispatch, in dispatch order]}}"""
    f = glob.glob(os.path.join(path, "**", "*_counter_collection.csv"), recursive=True)
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
    print("## (the rollout kernel is launched twice: 100 warm-up steps, then the 1000 timed steps = MaxNs; the step kernel's")
    print("##  1000 calls are bench.py's one-launch-per-step A/B pass)")
    for r in kernel_rows(os.path.join(d, "trace"), needle):
        print({k: (short(r[k]) if k == "Name" else r[k]) for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")})
    print("\n## PMC, separate passes; per kernel: mean per dispatch and the largest dispatch (rollout kernel: the 1000-step launch)")
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        for kern, cs in counters_by_kernel(os.path.join(d, sub), needle).items():
            for k, v in cs.items():
                print(f"{sub}: {kern}: {k}: mean {sum(v) / len(v):.1f} max {max(v):.1f} (n={len(v)})")


if __name__ == "__main__":
    main()
