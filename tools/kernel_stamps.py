#!/usr/bin/env python3
"""Phase profile of k_step from in-kernel s_memtime stamps (diagnostic build, never the shipped library):

    tools/build_stamps.sh            # -DPCBENV_STAMPS build into stamps_tmp/ (travels to the GPU box)
    PCBENV_STAMPS=1 PCBENV_LIB=$GRAFT_REPO_ROOT/stamps_tmp/libpcbenv_stamps.so python tools/kernel_stamps.py [c3|c4|c5] [envs]

Prints, for one launch without and one with terminal environments (episodes in lockstep) and for one launch with
staggered episode phases: the launch timeline from s_memrealtime (100 MHz) and the median shader cycles per phase."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rl-environment-for-component-placement_amd"))
import torch  # noqa: E402
from pcbenv import named_config  # noqa: E402
from pcbenv.batched_env import BatchedPlacementEnv  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
reward = sys.argv[3] if len(sys.argv) > 3 else "centroid"
opts = {"terminal_teams": int(sys.argv[4])} if len(sys.argv) > 4 and sys.argv[4] != "-" else None  # 0 = plain k_step, omitted / "-" = the default
traj = len(sys.argv) > 5 and sys.argv[5] == "traj"  # trajectory layout [17, B, ...] with compact feature tensors, one launch per step
cfg = named_config(name, reward)
B, L = int(sys.argv[2]) if len(sys.argv) > 2 else 4096, cfg.max_num_components
TERM = [("load", 0, 1), ("sample", 1, 2), ("update", 2, 3), ("fold+emit", 3, 4), ("reward:offsets+centroids", 4, 5),
        ("reward:segments (centroid) / beam search", 5, 6 if reward == "centroid" else 24), ("beam: count", 24, 25), ("both: centroid route + count", 25, 8), ("reward:terms+prefix", 6, 22), ("reward:pair walk", 12, 13), ("reward:tail batch", 13, 14),
        ("reward:reduce", 14, 15), ("reward:wirelength", 15, 8 if reward == "centroid" else 15), ("reset:instance+Q1", 9, 16), ("reset:fold+emit", 16, 17),
        ("reset:component features", 17, 18), ("reset:pin features", 18, 19), ("reset:rest", 19, 10), ("presample", 10, 20),
        ("store state", 20, 11)]
NONT = [("load", 0, 1), ("sample", 1, 2), ("update", 2, 3), ("fold+emit grid/mask", 3, 23), ("emit pin_grid", 23, 4), ("rest", 4, 10), ("presample", 10, 20),
        ("store state", 20, 11)]
if traj and name in ("c4", "c5"):  # spatial, trajectory layout: the feature part of the slot (it follows the cell tensors) in detail
    NONT[3:6] = [("cache tag + fold+emit grid/mask", 3, 23), ("emit pin_grid", 23, 4), ("compact feature tensors", 5, 6), ("copy from the episode's cache", 6, 7), ("rest", 7, 10)]


def report(title, s, done):
    a, b = s[:, 30], s[:, 31]  # s_memrealtime at wave start / end
    q = lambda v, f: float(np.quantile(v - a.min(), f)) * 10e-3
    clk = np.median((s[:, 11] - s[:, 0]) / np.maximum(b - a, 1)) * 100.0
    print(f"{title}: {int(done.sum())} terminal environments; presampled action used by {int((s[:, 21] > s[:, 0]).sum())}")
    print(f"  wave starts p50/max {q(a, .5):.2f}/{q(a, 1):.2f} us, ends p1/p50/p99/max {q(b, .01):.2f}/{q(b, .5):.2f}/"
          f"{q(b, .99):.2f}/{q(b, 1):.2f} us, median wave {np.median(b - a) * 10e-3:.2f} us, shader clock ~{clk:.0f} MHz")
    e = np.arange(len(a))
    # environment e runs on XCD e // (B/8), as the (e % (B/8))-th workgroup of that XCD (k_step: xcd_contiguous_env)
    n8 = len(a) // 8
    by_xcd = [float(np.median(b[e // n8 == x] - a.min())) * 10e-3 for x in range(8)]
    print("  median end by XCD (e // (B/8)): " + " ".join(f"{t:.1f}" for t in by_xcd) + " us; by dispatch order within the XCD, in eighths: " +
          " ".join(f"{float(np.median(b[(e % n8) // max(n8 // 8, 1) == x] - a.min())) * 10e-3:.1f}" for x in range(8)))
    for label, rows, table in (("terminal", s[done], TERM), ("non-terminal", s[~done], NONT)):
        if len(rows):
            print(f"  {label}: total {int(np.median(rows[:, 11] - rows[:, 0]))} cycles")
            for n, x, y in table:
                if (rows[:, x] == 0).all() or (rows[:, y] == 0).all():
                    continue  # a phase this configuration does not run (stamp never written)
                print(f"    {n:28s} {int(np.median(rows[:, y] - rows[:, x])):7d}")
            if label == "terminal" and os.environ.get("PCBENV_STAMPS_BEAM"):
                print("    beam search, lane 0 (net 0): setup %d, selection %d, expand %d, tie %d cycles" % tuple(int(np.median(rows[:, k])) for k in (26, 27, 28, 29)))
            elif label == "terminal" and rows[:, 28].any():
                print(f"    pair count detail: sweep steps {int(np.median(rows[:, 29]))}, candidates after the extent filter "
                      f"{int(np.median(rows[:, 28]))}, in-sweep dense batches {int(np.median(rows[:, 27]))} taking "
                      f"{int(np.median(rows[:, 26]))} cycles")


for stagger in (False, True):
    env = BatchedPlacementEnv(cfg, B, queue_depth=2, auto_reset=True, options=opts, num_slots=17 if traj else 1, compact_features=traj)
    env.generate_instances(); env.reset()
    env._L.pcbenv_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    acts = torch.empty((B, 3), dtype=torch.int32, device="cuda")
    idx = torch.arange(B, device="cuda")
    for t in range(2 * L):
        env.select_slot(t + 1)
        env.rollout_step(t, out=acts)
        if stagger and t < L:
            env.reset((idx % L == t).to(torch.uint8))
    seen = set()
    for k in range(3 * L):
        env.select_slot(k)
        env.rollout_step(100 + k, out=acts)
        torch.cuda.synchronize()
        done = env.done.cpu().numpy().astype(bool)
        kind = "staggered" if stagger else ("lockstep, terminal launch" if done.any() else "lockstep, non-terminal launch")
        if k < 2 or kind in seen:
            continue
        seen.add(kind)
        cap = 4096 * 3
        buf = np.zeros((B + cap, 32), np.uint64)
        assert env._L.pcbenv_debug_stamps(env._h, buf.ctypes.data) == 0
        allrows = buf.astype(np.int64)
        report(f"{name} x{B} {kind}", allrows[:B], done)
        hs = allrows[B:]
        hs = hs[hs[:, 30] > 0]
        if len(hs):  # reward helpers of this launch: when they started / ended relative to the first wavefront of the launch
            t0 = allrows[:B, 30].min()
            q = lambda v, f: float(np.quantile(v - t0, f)) * 10e-3
            print(f"  reward helpers: {len(hs)} ran; starts p1/p50/p99/max {q(hs[:, 30], .01):.2f}/{q(hs[:, 30], .5):.2f}/{q(hs[:, 30], .99):.2f}/{q(hs[:, 30], 1):.2f} us, "
                  f"ends p50/p99/max {q(hs[:, 31], .5):.2f}/{q(hs[:, 31], .99):.2f}/{q(hs[:, 31], 1):.2f} us, median cycles {int(np.median(hs[:, 11] - hs[:, 0]))}")
            for n_, x, y in TERM[:13]:
                ok = (hs[:, x] > 0) & (hs[:, y] > 0)
                if ok.any():
                    print(f"    helper {n_:28s} {int(np.median(hs[ok, y] - hs[ok, x])):7d}")
    env.close()
