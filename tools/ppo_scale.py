#!/usr/bin/env python3
"""PPO at scale around the device-resident path (SURVEY.md §8f row 2 / BASELINE config 4):

    python tools/ppo_scale.py [--config c4] [--envs 4096] [--iters 3]            # one GPU
    python -m torch.distributed.run --nproc-per-node N tools/ppo_scale.py ...    # one rank per GPU, RCCL

c4 (64x64 pin_spatial) x 4096 environments per GPU, 16-step rollouts written straight into the [17, B, ...]
trajectory tensors (no observation copies), fresh on-device instances at every reset, the policy network of
agent/models/rectangle_pin_spatial_model.py restated in PyTorch (its dense logits layer over 4*64*64 actions is what
dominates at this size), advantage standardisation by RCCL all-gather and gradient all-reduce across ranks.  Prints
one JSON line: env-steps/s inside the training loop (rollout = policy forward + env step; update = PPO epochs) next to
the env-only rate of the same loop without a policy.  Parity with RLlib is unpinned (not in the reference tree)."""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "rl-environment-for-component-placement_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from pcbenv import named_config  # noqa: E402
from pcbenv.batched_env import BatchedPlacementEnv  # noqa: E402
from pcbenv.policy import SpatialPolicy  # noqa: E402
from pcbenv.ppo import PPOConfig, PPOTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c4")
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--rollout-steps", type=int, default=16)
ap.add_argument("--minibatches", type=int, default=8)
ap.add_argument("--epochs", type=int, default=2)
ap.add_argument("--backend", default="nccl")
args = ap.parse_args()
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
dev = local % torch.cuda.device_count()
torch.cuda.set_device(dev)
if world > 1:
    dist.init_process_group(args.backend, device_id=torch.device("cuda", dev)) if args.backend == "nccl" else dist.init_process_group(args.backend)
torch.manual_seed(rank)  # different on purpose: the trainer must make the ranks agree
cfg = named_config(args.config)
B, T = args.envs, args.rollout_steps
env = BatchedPlacementEnv(cfg, B, device=f"cuda:{dev}", queue_depth=64, auto_reset=True, first_env_index=rank * B, num_slots=T + 1, compact_features=True)
env.enable_device_instances()
env.reset()
policy = SpatialPolicy(cfg).to(env.device)
tr = PPOTrainer(env, policy, PPOConfig(rollout_steps=T, epochs=args.epochs, minibatches=args.minibatches, lr=1e-4))
assert tr.in_place
if world > 1:  # the ranks agree on the weights after the broadcast
    w = torch.cat([p.detach().reshape(-1)[:64] for p in policy.parameters()])
    ws = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    assert all(torch.equal(ws[0], x) for x in ws)
tr.train(1)  # warm-up (allocator, autotuning)
tr.timing = {"collect_s": 0.0, "update_s": 0.0, "env_steps": 0}
tr.train(args.iters)
tm = tr.timing
# env-only: the same number of steps with uniformly sampled legal actions, same trajectory layout
# (one untimed pass first -- the update has just had the GPU to itself -- then at least twenty: 48 launches were a 4 ms sample)
env_iters = max(args.iters, 20)
for it in range(-1, env_iters):
    if it == 0:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    for t in range(T):
        env.select_slot((t + 1) % (T + 1))
        env.rollout_step(10_000 + (it + 1) * T + t)
torch.cuda.synchronize()
env_only = B * T * env_iters / (time.perf_counter() - t0)
params = sum(p.numel() for p in policy.parameters())
out = {"config": args.config, "envs_per_gpu": B, "n_gpus": world, "rollout_steps": T, "iterations": args.iters,
       "policy_parameters": params, "epochs": args.epochs, "minibatches": args.minibatches,
       "train_env_steps_per_sec_per_gpu": round(tm["env_steps"] / (tm["collect_s"] + tm["update_s"]), 1),
       "rollout_env_steps_per_sec_per_gpu": round(tm["env_steps"] / tm["collect_s"], 1),
       "collect_s_per_iter": round(tm["collect_s"] / args.iters, 4), "update_s_per_iter": round(tm["update_s"] / args.iters, 4),
       "env_only_env_steps_per_sec_per_gpu": round(env_only, 1), "env_only_launches": env_iters * T,
       "mean_return_per_iter": tr.returns[-args.iters:], "generator_errors": env.device_instance_errors(),
       "collectives": ("rccl all-gather of advantages + gradient all-reduce + BatchNorm statistics all-reduce" if world > 1 else "none (1 rank)"),
       "observation_copies_per_step": 0}
if world > 1:
    t = torch.tensor([out["train_env_steps_per_sec_per_gpu"]], device=env.device if args.backend == "nccl" else "cpu")
    dist.all_reduce(t)
    out["train_env_steps_per_sec_total"] = round(float(t.item()), 1)
if rank == 0:
    print(json.dumps(out), flush=True)
env.close()
if world > 1:
    dist.destroy_process_group()
