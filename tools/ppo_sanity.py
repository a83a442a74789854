#!/usr/bin/env python3
"""Learning-curve sanity of the PPO loop on the reference's shipped spatial config
(agent/config/rectangle_pin_spatial_model.json: 10x10 grid, 5 components 2x2, 3 nets x 6 pins, centroid reward).
Writes the mean episode return per iteration as JSON.  Parity with RLlib is unpinned; the check is only that
the return of the trained policy rises above that of the initial (near-uniform) one."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "rl-environment-for-component-placement_amd"))
import torch  # noqa: E402
from pcbenv import EnvConfig  # noqa: E402
from pcbenv.batched_env import BatchedPlacementEnv  # noqa: E402
from pcbenv.policy import SpatialPolicy  # noqa: E402
from pcbenv.ppo import PPOConfig, PPOTrainer  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
torch.manual_seed(0)
cfg = EnvConfig.spatial(10, 10, 9, 9, 2, 2, 2, 2, 5, 5, 3, 3, 6, 6, "centroid", 2, 0.75)
env = BatchedPlacementEnv(cfg, 1024, queue_depth=32, auto_reset=True, num_slots=11, compact_features=True)  # trajectory layout: no observation copies; compact feature tensors
env.enable_device_instances()  # a fresh instance for every episode, generated on the GPU
env.reset()
policy = SpatialPolicy(cfg).to(env.device)
tr = PPOTrainer(env, policy, PPOConfig(rollout_steps=10, lr=1e-3))
t0 = time.time()
curve = []
tr.train(iters, log=lambda it, r, s: (curve.append(r), print(f"iter {it:3d} mean_return {r:8.4f} entropy {s['entropy']:.3f}", flush=True)))
out = {"config": "10x10 spatial, 5 comps 2x2, 3 nets x 6 pins, centroid", "envs": 1024, "iterations": iters,
       "mean_return": curve, "first5": sum(curve[:5]) / 5, "last5": sum(curve[-5:]) / 5, "seconds": time.time() - t0, "episodes_per_env": env.queue_cursors()[0], "fresh_instances": "on-device generator", "generator_errors": env.device_instance_errors(),
       "observation_copies_per_step": 0 if tr.in_place else 1}
print(json.dumps(out))
