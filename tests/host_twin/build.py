"""Builds the TEST-ONLY host twin (g++ compile of the device math header). Never used by the product."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "space_gym_amd", "csrc")
OUT = os.path.join(HERE, "_build", "libsg_host_twin.so")


def build(force=False, defines=(), tag=""):
    """defines / tag: a variant of the twin (e.g. the probe-step build option), kept beside the default one"""
    out = OUT.replace(".so", tag + ".so")
    srcs = [os.path.join(HERE, "twin.cpp")] + [os.path.join(CSRC, f) for f in
                                                ("sg_device.hpp", "sg_host_config.hpp", "sg_config.h")]
    if force or not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        # -ffp-contract=off + explicit fmaf: the device code relies on explicit FMAs only where written
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-march=haswell", "-ffp-contract=off",
                               *["-D" + d for d in defines], "-I", CSRC, "-o", out, srcs[0]])
    return out
