import sys, time, os
sys.path.insert(0, "rl-environment-for-component-placement_amd")
import torch
from pcbenv import named_config
from pcbenv.batched_env import BatchedPlacementEnv
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c3"
cfg = named_config(cfgname)
Btot = {"c3": 4096, "c4": 4096, "c5": 8192}[cfgname]
for S in (1, 2, 4):
    B = Btot // S
    streams = [torch.cuda.Stream() for _ in range(S)]
    envs = []
    for i in range(S):
        with torch.cuda.stream(streams[i]):
            e = BatchedPlacementEnv(cfg, B, queue_depth=2, first_env_index=i * B, auto_reset=True)
            e.generate_instances(); e.reset()
            envs.append(e)
    acts = [torch.empty((B, 3), dtype=torch.int32, device="cuda") for _ in range(S)]
    def run(n, t0):
        for t in range(n):
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    envs[i].rollout_step(t0 + t, out=acts[i])
    run(32, 0); torch.cuda.synchronize()
    K = 320
    t = time.perf_counter(); run(K, 32); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(cfgname, "streams", S, "env-steps/s %.1fM" % (Btot * K / dt / 1e6), "us/step %.2f" % (dt / K * 1e6), flush=True)
    for e in envs: e.close()
