#!/usr/bin/env python3
"""Observation-level soak of the terminal list and its helper teams: one launch per step, episode phases spread over
the batch, external actions with a few corrupted ones, at the BASELINE batch sizes -- every observation tensor, reward,
done and info of every environment at every step against the CPU oracle (test infrastructure: drives
test_gpu_parity.py:_oracle_rollout; it lives under tests/ because it uses the oracle).  python tests/soak_stagger.py [episodes]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "rl-environment-for-component-placement_amd"), ROOT]
import test_gpu_parity as t  # noqa: E402
from pcbenv import named_config  # noqa: E402

eps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
t0, total = time.time(), 0
for name, reward, B in (("c3", "centroid", 4096), ("c4", "centroid", 4096), ("c3", "both", 2048), ("c4", "beam", 2048), ("c5", "centroid", 1024)):
    cfg = named_config(name, reward)
    L = cfg.max_num_components
    for mode in (dict(auto_reset=True), dict(auto_reset=True, fused=True), dict(), dict(auto_reset=True, num_slots=5, compact=True)):
        stats = {}
        n = t._oracle_rollout(cfg, B, episodes=eps, queue_depth=3, p_bad=0.0 if mode.get("fused") else 0.005, cpu_threads=16, stats=stats,
                              max_steps=(eps + 2) * L, stagger=L, **mode)
        total += n
        print(name, reward, B, mode, "env-steps", n, "routed terminals", stats.get("routed_terminals"), "worst-case terminals", stats.get("worst_case_terminals"),
              "ok", round(time.time() - t0, 1), "s", flush=True)
print("total env-steps compared:", total)
