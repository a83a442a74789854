"""Duration of every step launch over three episodes (HIP events around each launch, a synchronise between them):
python tools/launch_times.py c3 [terminal_teams] [stagger]  -> per launch: us, and the share of environments that ended an episode"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rl-environment-for-component-placement_amd"))
import torch
from pcbenv import named_config
from pcbenv.batched_env import BatchedPlacementEnv
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
teams = sys.argv[2] if len(sys.argv) > 2 else ""
stagger = len(sys.argv) > 3 and sys.argv[3] == "stagger"
cfg = named_config(name); B = {"c2": 4096, "c3": 4096, "c4": 4096, "c5": 8192}[name]
L = cfg.max_num_components
env = BatchedPlacementEnv(cfg, B, queue_depth=2, auto_reset=True, options={"terminal_teams": int(teams)} if teams != "" else None)
env.generate_instances(); env.reset()
acts = torch.empty((B, 3), dtype=torch.int32, device="cuda")
idx = torch.arange(B, device="cuda")
for t in range(2 * L):
    env.rollout_step(t, out=acts)
    if stagger and t < L:
        env.reset((idx % L == t).to(torch.uint8))
torch.cuda.synchronize()
out = []
for mode in ("fused", "external"):
    times = []
    for k in range(2 * L):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if mode == "external":
            env.sample_actions(500 + k, out=acts)
        torch.cuda.synchronize()
        e0.record()
        if mode == "fused":
            env.rollout_step(100 + k, out=acts)
        else:
            env.step(acts)
        e1.record(); torch.cuda.synchronize()
        times.append((e0.elapsed_time(e1) * 1e3, float(env.done.float().mean())))
    print(name, "teams", teams or "default", "stagger" if stagger else "lockstep", mode, " ".join("%.1f%s" % (t, "*" if d > 0.5 else "+" if d > 0 else "") for t, d in times), flush=True)
env.close()
