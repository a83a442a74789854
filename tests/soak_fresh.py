#!/usr/bin/env python3
"""Long soak of the fresh-instance path (test infrastructure: drives test_gpu_parity.py:_oracle_rollout, which uses the
oracle): on-device generator + fused auto-reset steps, EVERY tensor / reward / done / info of every environment at every
step against the oracle resetting from the host generator's records, many episodes per environment.
python tests/soak_fresh.py [episodes_per_env]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "rl-environment-for-component-placement_amd"), ROOT]
import test_gpu_parity as t  # noqa: E402

eps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
t0, total = time.time(), 0
for name, B in (("c3", 1024), ("c4", 512), ("small_spatial", 2048), ("small_pin", 2048), ("mid_spatial", 1024), ("c2", 2048), ("c5", 64)):
    cfg = t.GEN_CASES[name]()
    steps_per_ep = cfg.max_num_components
    stats = {}
    n = t._oracle_rollout(cfg, B, episodes=eps, queue_depth=8, p_bad=0.0, auto_reset=True, fused=True, device_instances=True,
                          cpu_threads=16, stats=stats, max_steps=eps * steps_per_ep + 8)
    total += n
    print(name, "envs", B, "env-steps compared", n, stats, "%.1f s" % (time.time() - t0), flush=True)
print("total env-steps compared on fresh on-device instances:", total)
