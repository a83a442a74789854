#!/usr/bin/env python3
"""Builds libpcbenv.so (HIP kernels + C ABI) for gfx950, in-tree.

hipcc cross-compiles without a GPU.  -ffp-contract=off: one IEEE operation per written operator (the reward path
must match the reference bit for bit).  The kernels of the four environment kinds are separate translation units
(csrc/pcb_kind_*.hip) compiled in parallel, then linked with the host side (csrc/pcbenv_kernels.hip) and the host
instance generator (csrc/instance_gen.cpp).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
UNITS = (["pcb_kind_square.hip", "pcb_kind_rect.hip"] + [f"pcb_kind_{k}_{p}.hip" for p in (2, 3, 0, 1) for k in ("spatial", "pin")]  # slowest first
         + ["pcbenv_kernels.hip", "instance_gen.cpp"])
SRC = [os.path.join(CSRC, u) for u in UNITS]
DEPS = SRC + [os.path.join(REPO, "include", "pcbenv.h")] + sorted(
    os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc")))
OUT = os.path.join(HERE, "libpcbenv.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wno-unused-value", "-pthread"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def build(force: bool = False, verbose: bool = False, out: str = OUT, extra_flags=(), jobs: int = 0) -> str:
    """`out` / `extra_flags`: diagnostic builds (tools/build_stamps.sh) next to the shipped library."""
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in DEPS):
        return out
    objdir = os.path.join(HERE, "build", os.path.splitext(os.path.basename(out))[0])
    os.makedirs(objdir, exist_ok=True)
    cc, inc = hipcc(), ["-I", os.path.join(REPO, "include"), "-I", CSRC]

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [cc] + FLAGS + list(extra_flags) + inc + ["-c", src, "-o", obj]
        import time
        t0 = time.time()
        subprocess.check_call(cmd)
        if verbose:
            print(f"{time.time() - t0:6.1f} s  " + " ".join(cmd[-4:]), flush=True)
        return obj
    with ThreadPoolExecutor(max_workers=jobs or min(len(SRC), os.cpu_count() or 1, 8)) as ex:
        objs = list(ex.map(compile_one, SRC))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", out] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a for a in sys.argv[1:] if a.endswith(".so")]
    print(build(force="--force" in sys.argv or bool(flags) or bool(outs), verbose=True, out=os.path.abspath(outs[0]) if outs else OUT, extra_flags=flags))
