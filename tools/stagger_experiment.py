"""Lock-step vs staggered episode phases (1/C of the environments terminal in every launch): one launch per step, and
the persistent rollout (16 steps per launch), where wavefronts drift apart anyway.  python tools/stagger_experiment.py c3"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rl-environment-for-component-placement_amd"))
import torch
from pcbenv import named_config
from pcbenv.batched_env import BatchedPlacementEnv
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
cfg = named_config(name); B = {"c2": 4096, "c3": 4096, "c4": 4096, "c5": 8192}[name]
L = cfg.max_num_components
for stagger in (False, True):
    env = BatchedPlacementEnv(cfg, B, queue_depth=2, auto_reset=True)
    env.generate_instances(); env.reset()
    acts = torch.empty((B, 3), dtype=torch.int32, device="cuda")
    idx = torch.arange(B, device="cuda")
    for t in range(2 * L):
        env.rollout_step(t, out=acts)
        if stagger and t < L:
            env.reset((idx % L == t).to(torch.uint8))
    torch.cuda.synchronize()
    K = 320 if name != "c5" else 64
    t0 = time.perf_counter()
    for k in range(K):
        env.rollout_step(100 + k, out=acts)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, "stagger", stagger, "one launch per step: %.1fM env-steps/s" % (B * K / dt / 1e6), "%.2f us/step" % (dt / K * 1e6), "done frac last step %.3f" % float(env.done.float().mean()), flush=True)
    out = torch.empty((16, B, 3), dtype=torch.int32, device="cuda")
    env.rollout_steps(1000, 16, out=out); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K // 16):
        env.rollout_steps(2000 + 16 * k, 16, out=out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, "stagger", stagger, "persistent rollout, 16 steps per launch (in place): %.1fM env-steps/s" % (B * (K // 16) * 16 / dt / 1e6), "%.2f us/step" % (dt / ((K // 16) * 16) * 1e6), flush=True)
    env.close()
