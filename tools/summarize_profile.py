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


def counter_avg(path, needle):
    f = glob.glob(os.path.join(path, "**", "*_counter_collection.csv"), recursive=True)
    if not f:
        return {}
    agg = {}
    for r in csv.DictReader(open(f[0])):
        if needle in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    d, needle = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "goal_step_kernel")
    print(f"# profile summary of {os.path.basename(d.rstrip('/'))}; kernel filter: {needle}")
    try:
        b = json.loads(open(os.path.join(d, "bench.json")).read().strip().splitlines()[-1])
        print("bench.py line:", json.dumps(b))
    except Exception as e:  # noqa: BLE001
        print("bench.json unreadable:", e)
    print("\n## rocprofv3 --kernel-trace --stats (python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-kernel-timing)")
    for r in kernel_rows(os.path.join(d, "trace"), needle):
        print({k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")})
    print("\n## PMC, separate passes, averages per launch of the step kernel (launch count)")
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        for k, (v, n) in counter_avg(os.path.join(d, sub), needle).items():
            print(f"{sub}: {k} = {v:.1f}  (n={n})")


if __name__ == "__main__":
    main()
