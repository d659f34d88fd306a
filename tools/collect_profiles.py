#!/usr/bin/env python3
"""Copies the round's measurement summaries from gpurun_out/ (scratch) into profiles/ (tracked) and rebuilds
profiles/traffic.json from the "traffic.json entries" blocks of the tools/gpu_profile.sh summaries.

    python tools/collect_profiles.py r4
"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
src, dst = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
names = {f"{tag}_k20": f"{tag}_k20_summary.txt", f"{tag}_k1000": f"{tag}_k1000_summary.txt",
         f"{tag}_kepler_k20": f"{tag}_kepler_k20_summary.txt", f"{tag}_kepler_k1000": f"{tag}_kepler_k1000_summary.txt",
         f"{tag}_4p_k20": f"{tag}_4p_k20_summary.txt", f"{tag}_4p_k1000": f"{tag}_4p_k1000_summary.txt",
         f"{tag}_2p4096_k20": f"{tag}_2p4096_k20_summary.txt", f"{tag}_2p4096_k1000": f"{tag}_2p4096_k1000_summary.txt"}
meas = []
for d, out in names.items():
    p = os.path.join(src, d, "summary.txt")
    if not os.path.exists(p):
        print("missing", p)
        continue
    shutil.copyfile(p, os.path.join(dst, out))
    txt = open(p).read()
    m = re.search(r"## profiles/traffic.json entries from this run\n(\[.*\])", txt, re.S)
    if m:
        for e in json.loads(m.group(1)):
            e["source"] = "profiles/" + out
            meas.append(e)
path = os.path.join(dst, "traffic.json")
old = json.load(open(path))
keep = [e for e in old["measurements"] if not any((e["env_id"], e["batch"], e["kernel"], e["steps_per_launch"]) ==
                                                  (n["env_id"], n["batch"], n["kernel"], n["steps_per_launch"]) for n in meas)]
old["measurements"] = meas + keep
json.dump(old, open(path, "w"), indent=1)
print(len(meas), "entries from", tag, "+", len(keep), "kept")
for f in (f"{tag}_ids_and_batches.txt",):
    if os.path.exists(os.path.join(src, f)):
        shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
