#!/usr/bin/env python3
"""CPU restatement (oracle, OpenMP) timed for c1-c5 at 1 thread and at the box's share of cores, with the
action stream recorded from the GPU and a parity check.  Run on the GPU box:
    python tools/cpu_baseline_sweep.py > gpurun_out/cpu_sweep.json"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "rl-environment-for-component-placement_amd"))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from pcbenv import named_config  # noqa: E402

out = {"cpu_model": bench.host_cpu_model(), "host_logical_cpus": os.cpu_count(), "configs": {}}
for name, (envs, steps) in {"c1": (256, 64), "c2": (4096, 64), "c3": (1024, 64), "c4": (1024, 48), "c5": (128, 64)}.items():
    res = bench.cpu_baseline(named_config(name), 0, envs, steps, 2)
    out["configs"][name] = {f"threads_{t}": {"env_steps_per_s": round(v[0], 1), "parity_with_gpu": v[1]} for t, v in res.items()}
    out["configs"][name]["sample"] = f"{envs} envs x {steps} steps"
    print(name, out["configs"][name], file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
