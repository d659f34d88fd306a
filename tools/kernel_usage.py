#!/usr/bin/env python3
"""Build aid (no GPU): registers, spills, LDS and occupancy of every kernel of the library as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.

    python tools/kernel_usage.py [substring ...]      only kernels whose demangled name contains every substring
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from space_gym_amd import build  # noqa: E402


def main():
    cmd = [build.hipcc(), *build.flags(("-Rpass-analysis=kernel-resource-usage",)), "-o", "/dev/null",
           os.path.join(build.CSRC, "sg_engine.hip")]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: \s*(.+?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        k, _, v = m.group(1).partition(":")
        k, v = k.strip(), v.strip()
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("%-66s %5s %5s %7s %7s %8s %4s" % ("kernel", "VGPR", "SGPR", "Sspill", "Vspill", "LDS", "occ"))
    for r, n in zip(rows, names):
        n = re.sub(r"^void ", "", n).split("(")[0]
        if all(a in n for a in sys.argv[1:]):
            print("%-66s %5s %5s %7s %7s %8s %4s" % (n[:66], r.get("VGPRs"), r.get("SGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"),
                                                 r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))


if __name__ == "__main__":
    main()
