#!/usr/bin/env python3
"""Builds libpcbenv.so (HIP kernels + C ABI) for gfx950, in-tree.

hipcc cross-compiles without a GPU.  -ffp-contract=off: one IEEE operation per
written operator (the reward path must match the reference bit for bit).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", "pcbenv_kernels.hip"), os.path.join(HERE, "csrc", "instance_gen.cpp")]
DEPS = SRC + [os.path.join(REPO, "include", "pcbenv.h")] + sorted(
    os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith(".h"))
OUT = os.path.join(HERE, "libpcbenv.so")


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in DEPS):
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
           "-Wno-unused-value", "-pthread", "-I", os.path.join(REPO, "include"), "-o", OUT] + SRC
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
