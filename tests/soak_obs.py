#!/usr/bin/env python3
"""Observation-level soak: every observation tensor, reward, done and info of every environment at every step against
the CPU oracle, at large batches, in the explicit / fused auto-reset / incremental modes (test infrastructure: this
drives test_gpu_parity.py:_oracle_rollout; it lives under tests/ because it uses the oracle).  python tests/soak_obs.py [scale]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "rl-environment-for-component-placement_amd"), ROOT]
import test_gpu_parity as t  # noqa: E402
from pcbenv import named_config  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 1
t0, total = time.time(), 0
for name, B, eps in (("c3", 512 * scale, 3), ("c4", 256 * scale, 3), ("c2", 512 * scale, 4), ("c5", 32 * scale, 1)):
    for mode in (dict(), dict(auto_reset=True, fused=True), dict(incremental=True), dict(auto_reset=True, fused=True, incremental=True)):
        n = t._oracle_rollout(named_config(name), B, episodes=eps, queue_depth=2 if name != "c5" else 1,
                              p_bad=0.0 if mode.get("fused") or name == "c5" else 0.02, **mode)
        total += n
        print(name, B, mode, "env-steps", n, "ok", round(time.time() - t0, 1), "s", flush=True)
print("total env-steps compared:", total)
