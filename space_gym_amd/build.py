"""Builds libspacegym_hip.so (HIP kernels for gfx950 + the C ABI of include/spacegym.h) in-tree with hipcc."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libspacegym_hip.so")
SOURCES = ["sg_engine.hip"]
HEADERS = ["sg_device.hpp", "sg_host_config.hpp", "sg_config.h", os.path.join(ROOT, "include", "spacegym.h")]
ARCH = "gfx950"


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the engine is HIP-only and cannot be built without ROCm")
    return exe


def flags(extra=()):
    # -ffp-contract=on: a*b+c fuses per source expression, identically in every kernel that inlines the same device
    # function (the default, fast, fuses across statements depending on context, so the per-step kernel and the fused
    # rollout kernel would round differently)
    return ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-ffp-contract=on",
            "-I", os.path.join(ROOT, "include"), "-I", CSRC, *extra]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES] + [f if os.path.isabs(f) else os.path.join(CSRC, f) for f in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=(), out=None):
    """out=None builds the product library; a different `out` (with e.g. extra=("-DSG_STAMPS",)) a diagnostic one."""
    if out is None and not force and not is_stale():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    out = out or LIB
    cmd = [hipcc(), *flags(extra), "-o", out, *[os.path.join(CSRC, f) for f in SOURCES]]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True,
                extra=("-Rpass-analysis=kernel-resource-usage",) if "--usage" in sys.argv else ()))
