"""Lock-step vs staggered episode phases (1/C of the environments terminal in every launch): one launch per step -- with
the fused sampler and with EXTERNAL actions (a policy between the steps: pcbenv_sample_actions + pcbenv_step here) --
and the persistent rollout (16 steps per launch), where wavefronts drift apart anyway.
python tools/stagger_experiment.py c3 [terminal_teams [reward_type]]   (terminal_teams: 0 = plain k_step, omitted or "-" = the default)"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rl-environment-for-component-placement_amd"))
import torch
from pcbenv import named_config
from pcbenv.batched_env import BatchedPlacementEnv
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
cfg = named_config(name, sys.argv[3]) if len(sys.argv) > 3 else named_config(name); B = {"c2": 4096, "c3": 4096, "c4": 4096, "c5": 8192}[name]
L = cfg.max_num_components
opts = {"terminal_teams": int(sys.argv[2])} if len(sys.argv) > 2 and sys.argv[2] != "-" else None
for stagger in (False, True):
    env = BatchedPlacementEnv(cfg, B, queue_depth=2, auto_reset=True, options=opts)
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
    # external actions: the action tensor comes from another kernel (here the stand-alone sampler), as a policy's would
    for k in range(32):
        env.sample_actions(5000 + k, out=acts); env.step(acts)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    step_ms = 0.0
    t0 = time.perf_counter()
    for k in range(K):
        env.sample_actions(6000 + k, out=acts)
        env.step(acts)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    for k in range(64):  # the step launch alone, by events
        env.sample_actions(7000 + k, out=acts)
        ev0.record(); env.step(acts); ev1.record(); torch.cuda.synchronize()
        step_ms += ev0.elapsed_time(ev1)
    print(name, "stagger", stagger, "external actions (sampler kernel + step): %.1fM env-steps/s" % (B * K / dt / 1e6), "%.2f us/step, step launch alone %.2f us" % (dt / K * 1e6, step_ms / 64 * 1e3), flush=True)
    out = torch.empty((16, B, 3), dtype=torch.int32, device="cuda")
    env.rollout_steps(1000, 16, out=out); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K // 16):
        env.rollout_steps(2000 + 16 * k, 16, out=out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, "stagger", stagger, "persistent rollout, 16 steps per launch (in place): %.1fM env-steps/s" % (B * (K // 16) * 16 / dt / 1e6), "%.2f us/step" % (dt / ((K // 16) * 16) * 1e6), flush=True)
    env.close()
