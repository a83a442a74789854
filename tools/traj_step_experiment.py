"""The trajectory layout with ONE launch per step -- the layout a PPO collect uses ([T + 1, B, ...] tensors, every step's
observation kept, a policy between the steps): env-steps/s of the environment side alone, float64 vs compact feature
tensors, replayed vs fresh on-device instances, write-through vs streaming stores.
python tools/traj_step_experiment.py c4 [envs [threads_per_env [quick]]]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rl-environment-for-component-placement_amd"))
import torch
from pcbenv import named_config
from pcbenv.batched_env import BatchedPlacementEnv
name = sys.argv[1] if len(sys.argv) > 1 else "c4"
cfg = named_config(name)
B = int(sys.argv[2]) if len(sys.argv) > 2 else {"c2": 1024, "c3": 4096, "c4": 4096, "c5": 8192}[name]
T = int(os.environ.get('TRAJ_T', '16'))  # slots - 1
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 0
quick = len(sys.argv) > 4
for compact in ((True,) if quick else (False, True)):
    for fresh in ((False,) if quick else (False, True)):
        for stream in ((None,) if quick else (None, 0)):
            opts = {} if stream is None else {"stream_threshold_bytes": stream}
            env = BatchedPlacementEnv(cfg, B, queue_depth=64 if fresh else 2, auto_reset=True, num_slots=T + 1, compact_features=compact, options=opts, threads_per_env=threads)
            if fresh:
                env.enable_device_instances()
            else:
                env.generate_instances()
            env.reset()
            acts = torch.empty((B, 3), dtype=torch.int32, device="cuda")
            for t in range(2 * T):
                env.select_slot((t + 1) % (T + 1)); env.rollout_step(t, out=acts)
            torch.cuda.synchronize()
            K = 20 * T
            t0 = time.perf_counter()
            for t in range(K):
                env.select_slot((t + 1) % (T + 1)); env.rollout_step(1000 + t, out=acts)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"{name} x{B}{f' ({threads} threads per environment)' if threads else ''} trajectory layout, one launch per step: features {'compact' if compact else 'float64'}, instances {'fresh (device)' if fresh else 'replayed'}, "
                  f"stores {'streaming' if stream == 0 else 'by size'}, {T + 1} slots: {B * K / dt / 1e6:.1f} M env-steps/s, {dt / K * 1e6:.1f} us/step", flush=True)
            env.close()
