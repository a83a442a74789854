#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the batched reset/step/mask/reward hot path.

    python bench.py --gpus N --steps K --warmup W [--config c3] [--reward centroid] [--envs 4096]

A "step" is one batched transition of all environments on a GPU, one kernel launch: on-device uniform
sampling of a legal action, the transition (place + legal mask + observations + the terminal routing
reward) and, after a terminal transition, the reset inside the same launch.  The headline replays a
queue of instances filled before the timed region from reference-exact RNG streams (as in round 1);
`fresh_instances` in the line is the same loop where every reset takes a NEW instance of the
environment's stream, generated on the GPU while the steps run (the reference's reset() semantics,
`--instances device`), `rollout` the persistent kernel that runs 16 steps per launch and keeps every
step's tensors (trajectory layout).  Environments shard
by global index over ranks with no data-path collective (weak scaling); `python bench.py --gpus N`
starts the N ranks itself.

One JSON line on rank 0: metric/value (whole-job env-steps/s), `roofline` for the dominant kernel
(k_step: algorithmic bytes per launch / mean launch duration from HIP events on the launch stream), and
`cpu_baseline` (the oracle restatement of the reference, timed on this box's host cores over a bounded
sample of the same instances and the same action stream, with a parity check of rewards/dones).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "rl-environment-for-component-placement_amd"))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402


# the UNMODIFIED reference Python env timed in the survey container (Icelake-class Xeon VM, 1 thread,
# Python 3.10.12, numpy 2.2.6, scipy 1.15.3; step() wall time under uniform-random legal actions).  The reference cannot
# travel to the GPU box, so these are quoted, labelled, next to the port's numbers measured in this run.
def _ref(step_ms, reset_ms, per_core, entry):
    return {"label": "reference Python, survey container, 1 core (BASELINE.md section 2; not measured in this run)", "step_ms": step_ms,
            "reset_ms": reset_ms, "env_steps_per_sec_per_core": per_core, "entry_point": entry}


REFERENCE_PYTHON = {
    ("c1", "centroid"): _ref(0.011, 0.013, 88000, "environment/dummy_env_square.py:115"),
    ("c2", "centroid"): _ref(0.128, 0.197, 7800, "environment/dummy_env_rectangular.py:353"),
    ("c3", "centroid"): _ref(0.72, 1.04, 1380, "environment/dummy_env_rectangular_pin.py:1599"),
    ("c3", "both"): _ref(0.90, 1.01, 1110, "environment/dummy_env_rectangular_pin.py:1599"),
    ("c4", "centroid"): _ref(0.89, 1.14, 1120, "environment/dummy_env_rectangular_pin_spatial.py:1551"),
    ("c5", "centroid"): _ref(3.73, 4.97, 270, "environment/dummy_env_rectangular_pin_spatial.py:1551"),
    ("c5", "both"): _ref(4.16, 4.99, 240, "environment/dummy_env_rectangular_pin_spatial.py:1551"),
}


def algorithmic_bytes_per_env_step(cfg) -> int:
    """1 byte per cell for grid / action_mask / pin_grid, 8-byte table records."""
    from pcbenv.config import KIND_SPATIAL
    HW = cfg.height * cfg.width
    b = HW + HW + cfg.num_orientations * HW
    if cfg.kind == KIND_SPATIAL:
        b += (cfg.max_num_nets + 1) * HW
    b += 2 * 8 * (cfg.max_num_components + cfg.max_total_pins)
    return b


def trajectory_bytes_per_env_step(cfg, compact=False) -> int:
    """Bytes one step writes in the trajectory layout: every tensor of the slot, whole (cells + the feature tensors,
    float64 or compact, + component_grid), plus the state tables it reads and writes once per launch (not counted per step)."""
    from pcbenv.batched_env import COMPACT_DTYPES, FEATURE_KEYS, obs_spec
    import math
    total = 0
    for k, (shape, dt) in obs_spec(cfg).items():
        if compact and k in FEATURE_KEYS:
            dt = COMPACT_DTYPES[k]
        total += math.prod(shape) * torch.empty((), dtype=dt).element_size()
    return total + 8 + 1 + 16  # reward, done, info


def rollout_leg(cfg, args, B, dev_index, rank, T, compact=False):
    """The persistent rollout (pcbenv_rollout_sampled): T steps per launch, state held in LDS, every step's tensors kept
    in their own slot of [T + 1, B, ...] buffers -- the on-device counterpart of the reference's simulate() loop."""
    from pcbenv.batched_env import BatchedPlacementEnv
    S = T + 1
    env = BatchedPlacementEnv(cfg, B, device=f"cuda:{dev_index}", queue_depth=4 * T if args.instances == "device" else max(args.queue_depth, 4),
                              run_seed=args.run_seed, first_env_index=rank * B, auto_reset=True, threads_per_env=args.threads_per_env, num_slots=S,
                              compact_features=compact)
    if args.instances == "device":
        env.enable_device_instances()
    else:
        env.generate_instances()
    env.reset()
    acts = torch.empty((T, B, 3), dtype=torch.int32, device=env.device)
    launches = max(8, args.steps // T)  # at least eight launches whatever --steps is: a spread, not a single sample
    pos = 1
    for _ in range(2):  # warm-up launches
        env.select_slot(pos); env.rollout_steps(0, T, out=acts); pos = (pos + T) % S
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[0].record()
    for k in range(launches):
        env.select_slot(pos); env.rollout_steps((2 + k) * T, T, out=acts); pos = (pos + T) % S
        evs[k + 1].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_launch = [evs[k].elapsed_time(evs[k + 1]) for k in range(launches)]
    kernel_ms = evs[0].elapsed_time(evs[launches]) / launches
    nbytes = trajectory_bytes_per_env_step(cfg, compact)
    gen_errors = env.device_instance_errors() if args.instances == "device" else None
    env.close()
    gbps = nbytes * B * T / (kernel_ms * 1e-3) / 1e9
    return {"value": round(B * T * launches / dt, 1), "unit": "env-steps/s", "steps_per_launch": T, "num_slots": S,
            "launches": launches, "ms_per_step": round(dt / (launches * T) * 1e3, 5), "kernel_ms_per_launch": round(kernel_ms, 4),
            "kernel_ms_per_launch_min_median_max": [round(min(per_launch), 4), round(float(np.median(per_launch)), 4), round(max(per_launch), 4)],
            "bytes_written_per_env_step": nbytes, "achieved_GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000.0, 4),
            "algorithmic_GBps": round(algorithmic_bytes_per_env_step(cfg) * B * T / (kernel_ms * 1e-3) / 1e9, 1),
            "algorithmic_frac_of_8TBps": round(algorithmic_bytes_per_env_step(cfg) * B * T / (kernel_ms * 1e-3) / 1e9 / 8000.0, 4),
            "feature_tensors": "compact (int16 / int8 / uint8: pcbenv_bind_compact_features)" if compact else "float64",
            "instances": args.instances, "generator_errors": gen_errors,
            "note": "every tensor of every step kept (trajectory layout), feature tensors included; this rank only"}


def external_actions_staggered_leg(cfg, args, B, dev_index, rank):
    """The loop a policy runs (reference: utils/agent/utils.py:221-256, agent/random/random_policy_square.py:38-56): the
    action tensor comes from ANOTHER kernel between the steps (here the stand-alone uniform sampler, as a policy's
    forward would produce it), one pcbenv_step launch per step, and the episodes end at different times -- the phases
    are spread evenly, 1 / max_num_components of the batch ends an episode in every launch.  The terminal list and
    its helper wavefronts are what keeps such a launch as short as a lock-step one."""
    from pcbenv.batched_env import BatchedPlacementEnv
    from pcbenv.config import KIND_SQUARE
    L = max(1, cfg.max_num_components if cfg.kind != KIND_SQUARE else 4)
    steps = max(args.steps, 8 * L)
    env = BatchedPlacementEnv(cfg, B, device=f"cuda:{dev_index}", queue_depth=max(args.queue_depth, 2), run_seed=args.run_seed,
                              first_env_index=rank * B, auto_reset=True, threads_per_env=args.threads_per_env)
    env.generate_instances()
    env.reset()
    acts = torch.empty((B, 3), dtype=torch.int32, device=env.device)
    idx = torch.arange(B, device=env.device)
    for t in range(2 * L):  # spread the phases: environment i restarts once more after step i % L
        env.sample_actions(t, out=acts); env.step(acts)
        if t < L:
            env.reset((idx % L == t).to(torch.uint8))
    for t in range(max(args.warmup, 2 * L)):
        env.sample_actions(100 + t, out=acts); env.step(acts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        env.sample_actions(1000 + k, out=acts)
        env.step(acts)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    done_frac = float(env.done.float().mean())
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
    for k, (e0, e1) in enumerate(ev):  # the step launch alone, by events on the launch stream
        env.sample_actions(5000 + k, out=acts)
        e0.record(); env.step(acts); e1.record()
    torch.cuda.synchronize()
    step_ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    env.close()
    b_alg = algorithmic_bytes_per_env_step(cfg)
    return {"value": round(B * steps / dt, 1), "unit": "env-steps/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 5),
            "step_kernel_ms": round(step_ms, 5), "frac_of_8TBps": round(b_alg * B / (step_ms * 1e-3) / 1e9 / 8000.0, 4),
            "frac_of_8TBps_whole_loop": round(b_alg * B * steps / dt / 1e9 / 8000.0, 4),
            "terminal_fraction_per_launch": round(done_frac, 4),
            "note": "actions from a separate kernel between the steps (k_sample), one pcbenv_step launch per step, episode phases spread "
                    "evenly; ms_per_step includes the sampler launch, step_kernel_ms is the step launch alone (median of 64, HIP events); this rank only"}


def fresh_instances_leg(cfg, args, B, dev_index, rank):
    """The same fused loop with the reference's reset() semantics: every reset takes a FRESH instance of the
    environment's stream, generated on the GPU (k_gen_fill on a side stream) while the steps run.  Timed over at
    least 256 steps so that the region holds several refills whatever --steps is."""
    from pcbenv.batched_env import BatchedPlacementEnv
    steps = max(args.steps, 256)
    env = BatchedPlacementEnv(cfg, B, device=f"cuda:{dev_index}", queue_depth=64, run_seed=args.run_seed, first_env_index=rank * B,
                              auto_reset=True, threads_per_env=args.threads_per_env)
    env.enable_device_instances()
    env.reset()
    actions = torch.empty((B, 3), dtype=torch.int32, device=env.device)
    for t in range(max(args.warmup, 64)):
        env.rollout_step(t, out=actions)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        env.rollout_step(64 + k, out=actions)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    errors = env.device_instance_errors()
    env.close()
    return {"value": round(B * steps / dt, 1), "unit": "env-steps/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 5),
            "frac_of_8TBps": round(algorithmic_bytes_per_env_step(cfg) * B * steps / dt / 1e9 / 8000.0, 4),
            "queue_depth": 64, "generator_errors": errors,
            "note": "same loop and kernel; a new instance at every reset, generated on the GPU inside the timed region; this rank only"}


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def measured_copy_ceiling_gbps(device):
    """Device-to-device copy of 1 GiB (read + write), best of 5: the practical HBM ceiling next to the 8 TB/s spec."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=device)
    b = torch.empty(n, dtype=torch.uint8, device=device)
    best = 0.0
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    return best


def measured_fill_ceiling_gbps(device):
    """Write-only stream (1 GiB `fill_`), best of 5: the path is ~95 % stores, so this is the closer ceiling."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=device)
    best = 0.0
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        a.fill_(1)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a
    return best


def cpu_baseline(cfg, run_seed, sample_envs, sample_steps, queue_depth):
    """Time the CPU oracle (port of the reference) on a bounded sample; parity-check it against the GPU."""
    from oracle import oracle as orc
    from pcbenv import pack_instances
    from pcbenv.batched_env import BatchedPlacementEnv
    from pcbenv.config import KIND_SQUARE
    env = BatchedPlacementEnv(cfg, sample_envs, queue_depth=queue_depth, run_seed=run_seed)
    packed = env.generate_instances(verify=16) if cfg.kind != KIND_SQUARE else None
    env.reset()
    acts, dones, rewards = [], [], []
    for t in range(sample_steps):  # record the action stream on the GPU
        a = env.sample_actions(t)
        _, r, d, _ = env.step(a)
        acts.append(a.cpu().numpy().copy()); dones.append(d.cpu().numpy().copy()); rewards.append(r.cpu().numpy().copy())
        env.reset_done()
    n_obs = min(32, sample_envs)  # full observations of the first environments after the last step, for the parity check
    final_obs = {k: v[:n_obs].cpu().numpy().astype(np.float64) for k, v in env.obs.items()}
    env.close()
    ob = orc.OracleBatch(cfg, sample_envs)
    out = {}
    # the GPU box gives one GPU's share of the host: 16 cores (do not oversubscribe the shared machine)
    for threads in sorted({1, max(1, min(16, ob.max_threads, os.cpu_count() or 1))}):
        cursor = np.zeros(sample_envs, np.int64)

        def do_reset(mask):
            if cfg.kind == KIND_SQUARE:
                for i in np.flatnonzero(mask):
                    ob.env(i).reset()
                return
            rec = np.stack([packed[cursor[i] % queue_depth][i] for i in range(sample_envs)])
            ob.reset_packed(rec, mask.astype(np.uint8), threads)
            cursor[mask.astype(bool)] += 1
        do_reset(np.ones(sample_envs, np.uint8))
        ok = True
        t0 = time.perf_counter()
        for t in range(sample_steps):
            r, d, _ = ob.step(acts[t], threads)
            ok &= bool(np.array_equal(d, dones[t]) and np.array_equal(r.view(np.uint64), rewards[t].view(np.uint64)))
            do_reset(d)
        dt = time.perf_counter() - t0
        for i in range(n_obs):  # observations too (the oracle has reset the finished environments like the GPU did)
            want = ob.env(i).obs()
            ok &= all(np.array_equal(final_obs[k][i], want[k]) for k in final_obs)
        out[threads] = (sample_envs * sample_steps / dt, ok)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=320)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--config", default="c3", help="c2 | c3 | c4 | c5 (SURVEY.md §8)")
    ap.add_argument("--reward", default="centroid", choices=["centroid", "beam", "both"])
    ap.add_argument("--envs", type=int, default=0, help="environments per GPU (default: the config's batch)")
    ap.add_argument("--queue-depth", type=int, default=0, help="instances queued per environment (default: 2 for replay, 64 for device)")
    ap.add_argument("--instances", default="replay", choices=["device", "replay"],
                    help="replay (the headline, comparable with round 1): the queue is filled once by the host generator before "
                         "the timed region and replayed round robin; device: every reset takes a FRESH instance of the "
                         "environment's reference-exact stream, generated on the GPU while the steps run (the reference's "
                         "reset() semantics) -- with the default `replay` that variant is timed as well and reported as "
                         "`fresh_instances`")
    ap.add_argument("--run-seed", type=int, default=0)
    ap.add_argument("--incremental", action="store_true", help="PCBENV_FLAG_INCREMENTAL_OBS")
    ap.add_argument("--chunk", type=int, default=1,
                    help="fused loop only: issue this many steps per host call (pcbenv_rollout_sampled)")
    ap.add_argument("--loop", default="fused", choices=["fused", "explicit"],
                    help="fused: one launch per step (pcbenv_step_sampled + PCBENV_FLAG_AUTO_RESET); "
                         "explicit: sample_actions, step, reset_done as three launches (reference-style loop)")
    ap.add_argument("--threads-per-env", type=int, default=0)
    ap.add_argument("--event-steps", type=int, default=128, help="extra steps timed per kernel launch with HIP events")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank path on one GPU)")
    ap.add_argument("--device-index", type=int, default=-1, help="GPU index of this rank (default LOCAL_RANK)")
    ap.add_argument("--adv-allgather", dest="adv_allgather", action="store_true", default=None,
                    help="N > 1 (default there): every 16 steps all-gather + standardise a [16*B] float32 advantage tensor "
                         "(the PPO-side collective, 256 KiB per rank at 4 096 envs) over RCCL")
    ap.add_argument("--no-adv-allgather", dest="adv_allgather", action="store_false")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; `value` comes from the first, the spread of all is reported")
    ap.add_argument("--rollout-steps", type=int, default=16,
                    help="also time the persistent rollout kernel with this many steps per launch in the trajectory layout "
                         "(reported as `rollout`; 0 = skip)")
    ap.add_argument("--no-fresh-leg", action="store_true", help="skip the extra leg with a fresh on-device instance at every reset")
    ap.add_argument("--no-staggered-leg", action="store_true", help="skip the extra leg with external actions and staggered episode phases")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start N fresh ranks (one per GPU) and relay rank 0's line.
        # Nothing in this process has touched the GPU (importing torch does not), and it is not replaced: the ranks
        # are children, their stdout is ours.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if args.backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} over RCCL needs {world} visible GPUs, this node shows {torch.cuda.device_count()} "
                         "(one rank per GPU; --backend gloo only rehearses the multi-rank path on fewer cards)")
    if args.adv_allgather is None:
        args.adv_allgather = world > 1
    dev_index = local_rank if args.device_index < 0 else args.device_index
    if args.backend != "nccl" and args.device_index < 0:
        dev_index = local_rank % torch.cuda.device_count()  # rehearsal: more ranks than GPUs share the cards
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)

    from pcbenv import named_config
    from pcbenv.batched_env import BatchedPlacementEnv
    cfg = named_config(args.config, args.reward)
    if args.queue_depth <= 0:
        args.queue_depth = 64 if args.instances == "device" else 2
    default_B = {"c1": 1, "c2": 1024, "c3": 4096, "c4": 4096, "c5": 8192}[args.config]
    B = args.envs or default_B
    env = BatchedPlacementEnv(cfg, B, device=f"cuda:{dev_index}", queue_depth=args.queue_depth,
                              run_seed=args.run_seed, first_env_index=rank * B, incremental_obs=args.incremental,
                              auto_reset=(args.loop == "fused"), threads_per_env=args.threads_per_env)
    t_gen = time.perf_counter()
    if args.instances == "device":
        env.enable_device_instances()
    else:
        env.generate_instances()
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    env.reset()
    actions = torch.empty((B, 3), dtype=torch.int32, device=env.device)

    def one_step(t, ev=None):
        if args.loop == "fused":
            if ev is not None:
                ev[0].record()
            env.rollout_step(t, out=actions)
            if ev is not None:
                ev[1].record()
            return
        env.sample_actions(t, out=actions)
        if ev is not None:
            ev[0].record()
        env.step(actions)
        if ev is not None:
            ev[1].record()
        env.reset_done()

    for t in range(args.warmup):
        one_step(t)
    # HIP events on torch's current stream (= the stream every kernel is launched on).  In the fused loop a
    # step IS one k_step launch, so one event pair around the timed region gives the mean launch duration with
    # no per-launch event overhead; in the explicit loop a pair brackets each k_step launch (this costs ~5 us
    # per event and is therefore done in a second, untimed-for-`value` pass).
    chunk = max(1, args.chunk) if args.loop == "fused" else 1
    chunk_actions = torch.empty((chunk, B, 3), dtype=torch.int32, device=env.device) if chunk > 1 else None
    adv = torch.randn(16 * B, device=env.device) if (dist and args.adv_allgather) else None
    adv_stream = None
    if adv is not None:
        from pcbenv.distributed import normalize_advantages
        # The collective does not depend on the environment steps that follow it (in PPO it belongs to the rollout that
        # has just ended), so it runs on a side stream and overlaps them; the timed region ends with a device-wide
        # synchronize, which covers it.
        adv_stream = torch.cuda.Stream(device=env.device)

    def timed_region(step0):
        """EXACTLY args.steps steps between barrier + synchronize on both sides -> (own wall seconds, event ms)."""
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        t0 = time.perf_counter()
        ev[0].record()
        if chunk > 1:
            for k in range(0, args.steps, chunk):
                env.rollout_steps(step0 + k, min(chunk, args.steps - k), out=chunk_actions)
        else:
            for k in range(args.steps):
                one_step(step0 + k)
                if adv is not None and k % 16 == 15:  # the PPO-side collective: 64 KiB * B / 1024 per rank over RCCL
                    with torch.cuda.stream(adv_stream):
                        normalize_advantages(adv if args.backend == "nccl" else adv.cpu(), "all_gather")
        ev[1].record()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        return time.perf_counter() - t0, ev[0].elapsed_time(ev[1])

    def over_ranks(x):
        """-> (max over ranks, list of per-rank values)"""
        if not dist:
            return x, [x]
        tt = torch.tensor([x], dtype=torch.float64, device=env.device if args.backend == "nccl" else "cpu")
        allv = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allv, tt)
        vals = [float(v.item()) for v in allv]
        return max(vals), vals

    own, region_ms = timed_region(args.warmup)
    elapsed, per_rank_s = over_ranks(own)
    repeats_ms = [elapsed / args.steps * 1e3]
    for r in range(1, max(1, args.repeats)):  # the spread: further regions of the same length, not part of `value`
        own_r, _ = timed_region(args.warmup + r * args.steps)
        repeats_ms.append(over_ranks(own_r)[0] / args.steps * 1e3)
    steps_done = args.warmup + max(1, args.repeats) * args.steps
    step_kernel_ms = None
    if not args.no_kernel_events:
        if args.loop == "fused" and cfg.reward_type == "centroid":
            step_kernel_ms = region_ms / args.steps
        elif args.event_steps > 0:
            events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.event_steps)]
            for k in range(args.event_steps):
                one_step(steps_done + k, events[k])
            torch.cuda.synchronize()
            step_kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))

    gen_errors_main = env.device_instance_errors() if args.instances == "device" and cfg.kind != 0 else None
    env.close()
    # N > 1: BASELINE.json's config 4 (pin_spatial 64x64, 4 096 environments per GPU, sharded) next to the headline workload,
    # which stays c3 for every N so that the per-N values of one scaling run are comparable
    config4 = None
    if dist and args.config == "c3" and not args.incremental:
        cfg4 = named_config("c4", args.reward)
        env4 = BatchedPlacementEnv(cfg4, B, device=f"cuda:{dev_index}", queue_depth=args.queue_depth, run_seed=args.run_seed,
                                   first_env_index=rank * B, auto_reset=True, threads_per_env=args.threads_per_env)
        env4.generate_instances(); env4.reset()
        a4 = torch.empty((B, 3), dtype=torch.int32, device=env4.device)
        n4 = max(64, args.steps)
        for t in range(32):
            env4.rollout_step(t, out=a4)
        torch.cuda.synchronize(); dist.barrier()
        t4 = time.perf_counter()
        for t in range(n4):
            env4.rollout_step(32 + t, out=a4)
        torch.cuda.synchronize(); dist.barrier()
        t4 = over_ranks(time.perf_counter() - t4)[0]
        env4.close()
        config4 = {"workload": "c4: pin_spatial 64x64, 16 components, 48 pins", "envs_per_gpu": B, "steps": n4,
                   "value": round(world * B * n4 / t4, 1), "unit": "env-steps/s (all ranks)", "ms_per_step": round(t4 / n4 * 1e3, 5)}
    staggered = None
    if args.loop == "fused" and not args.incremental and cfg.kind != 0 and not args.no_staggered_leg:
        staggered = external_actions_staggered_leg(cfg, args, B, dev_index, rank)
    fresh = None
    if args.instances == "replay" and args.loop == "fused" and not args.incremental and cfg.kind != 0 and not args.no_fresh_leg:
        fresh = fresh_instances_leg(cfg, args, B, dev_index, rank)
    rollout = None
    if args.rollout_steps > 0 and not args.incremental and args.config != "c1":
        if dist:
            dist.barrier()
        rollout = rollout_leg(cfg, args, B, dev_index, rank, args.rollout_steps)
        if cfg.kind != 0:  # the same with the compact feature tensors a rollout that keeps every step would bind
            rollout["compact_features"] = rollout_leg(cfg, args, B, dev_index, rank, args.rollout_steps, compact=True)
        if dist:
            dist.barrier()
    if rank == 0:
        b_alg = algorithmic_bytes_per_env_step(cfg)
        if args.incremental:
            # only the rows of the placed rectangle are rewritten in grid / pin_grid: count what is moved
            from pcbenv.config import KIND_SPATIAL
            rows = (cfg.min_component_h + cfg.max_component_h + cfg.min_component_w + cfg.max_component_w) / 4.0
            per_row = cfg.width * (1 + (cfg.max_num_nets + 1 if cfg.kind == KIND_SPATIAL else 0))
            b_alg = int(cfg.num_orientations * cfg.height * cfg.width + rows * per_row
                        + 2 * 8 * (cfg.max_num_components + cfg.max_total_pins) + 2 * 3 * cfg.height * ((cfg.width + 63) // 64) * 8)
        value = world * B * args.steps / elapsed
        roof = None
        if step_kernel_ms:
            achieved = b_alg * B / (step_kernel_ms * 1e-3) / 1e9
            # HBM bytes per launch from the PMC counters: collected in separate rocprofv3 --pmc passes of this same
            # command (tools/profile_all.sh) and committed; a bench run cannot read the counters itself.
            traffic, traffic_source = None, None
            tname = f"traffic_{args.config}.json" if B == default_B else f"traffic_{args.config}_{B}.json"
            tpath = os.path.join(REPO, "profiles", tname)
            if os.path.exists(tpath) and not args.incremental and args.loop == "fused":
                try:
                    traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
                    traffic_source = "profiles/" + tname + " (rocprofv3 --pmc passes of this command, not this run)"
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "kernel": "k_step", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(achieved / 8000.0, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "measured_copy_ceiling": round(measured_copy_ceiling_gbps(env.device), 1),
                    "measured_fill_ceiling": round(measured_fill_ceiling_gbps(env.device), 1),
                    "algorithmic_bytes_per_env_step": b_alg, "units_per_launch": B,
                    "kernel_ms": round(step_kernel_ms, 5)}
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            sample_envs, sample_steps = {"c2": (4096, 128), "c3": (2048, 64), "c4": (2048, 48), "c5": (256, 64)}.get(args.config, (64, 64))
            res = cpu_baseline(cfg, args.run_seed, sample_envs, sample_steps, args.queue_depth)
            nthr = max(res)
            cpu = {"value": round(res[nthr][0], 1), "unit": "env-steps/s", "cores": nthr, "kind": "port",
                   "sample": f"{sample_envs} envs x {sample_steps} steps of the same instances and action stream (oracle/pcbenv_oracle.c, OpenMP)",
                   "single_thread_value": round(res[1][0], 1), "parity_with_gpu": bool(all(v[1] for v in res.values())),
                   "parity_scope": "reward + done of every sampled env-step, all observation tensors of 32 environments after the last step",
                   "cpu_model": host_cpu_model(), "host_logical_cpus": os.cpu_count(),
                   "threads_note": "1 and 16 threads: 16 is one GPU's share of this host (the pool's limit for a 1-GPU lease), not all physical cores",
                   "reference_python": REFERENCE_PYTHON.get((args.config, args.reward))}
        L = cfg.max_num_components if cfg.kind != 0 else None
        lockstep = cfg.kind != 0 and cfg.min_num_components == cfg.max_num_components and args.loop == "fused"
        term_in_region = sum(1 for k in range(args.warmup, args.warmup + args.steps) if k % L == L - 1) if lockstep else None
        rccl_version = None
        if dist and args.backend == "nccl":
            try:
                rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rccl_version = "unknown"
        line = {"metric": "env_steps_per_sec", "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
                "ms_per_step_repeats": {"n": len(repeats_ms), "median": round(float(np.median(repeats_ms)), 5),
                                        "min": round(min(repeats_ms), 5), "max": round(max(repeats_ms), 5)},
                "per_rank_env_steps_per_sec": [round(B * args.steps / t, 1) for t in per_rank_s],
                "world_size_observed": (dist.get_world_size() if dist else 1),
                "backend": (("rccl" if args.backend == "nccl" else args.backend) if dist else None), "rccl_version": rccl_version,
                "per_rank_first_env_index": [r * B for r in range(world)],
                "terminal_launches_in_region": term_in_region,
                "terminal_launches_note": (f"episodes are {L} steps and in lock-step: a region of {args.steps} steps from step {args.warmup} holds "
                                           f"{term_in_region} terminal launches (reward + reset of every environment), {args.steps / L:.2f} on average") if lockstep else None,
                "adv_allgather_bytes_per_rank": (16 * B * 4 if adv is not None else 0),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
                "data": "synthetic (reference-exact instance streams, seeds 1000003*run_seed+env" + (", generated on the GPU, a new instance at every reset" if args.instances == "device" else ", queued once and replayed") + "; uniform legal actions drawn on device)",
                "config": {"workload": f"{args.config}: {cfg.height}x{cfg.width} grid, {cfg.max_num_components} components, "
                                       f"{cfg.max_total_pins} pins, reward={args.reward}", "envs_per_gpu": B,
                           "queue_depth": args.queue_depth, "instances": args.instances,
                           "fresh_instance_every_reset": args.instances == "device", "generator_errors": gen_errors_main, "incremental_obs": bool(args.incremental), "loop": args.loop, "steps_per_host_call": chunk,
                           "instance_generation_s": round(t_gen, 2),
                           "store_policy": "sc1 nt (streaming)" if cfg.cell_tensor_bytes_per_step(args.incremental) * B
                           > 256 * (1 << 20) else "sc1 (write-through)"},
                "roofline": roof, "cpu_baseline": cpu, "north_star_config4": config4, "external_actions_staggered": staggered, "fresh_instances": fresh, "rollout": rollout}
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()  # rank 0 has printed: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
