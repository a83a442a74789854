"""c5 (128x128 pin_spatial, 8 192 environments, 3.1 GB written per launch) has been seen at two speeds with one library
(14.8 M or 17.0 M env-steps/s).  This records, for a series of placements of the three big cell tensors, their base
addresses and the step time: default allocations (fresh process state), then one arena with the tensors at chosen offsets.
python tools/c5_modes.py [steps]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rl-environment-for-component-placement_amd"))
import torch
from pcbenv import named_config
from pcbenv.batched_env import BatchedPlacementEnv
cfg = named_config("c5"); B = 8192
K = int(sys.argv[1]) if len(sys.argv) > 1 else 48
BIG = ("grid", "action_mask", "pin_grid")


def run(tag, allocator=None):
    env = BatchedPlacementEnv(cfg, B, queue_depth=2, auto_reset=True, allocator=allocator)
    env.generate_instances(); env.reset()
    acts = torch.empty((B, 3), dtype=torch.int32, device="cuda")
    for t in range(8):
        env.rollout_step(t, out=acts)
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for t in range(K):
            env.rollout_step(100 + rep * K + t, out=acts)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e6)
    addr = {k: env.traj[k].data_ptr() for k in BIG}
    print(f"{tag:34s} us/step {ts[0]:7.1f} {ts[1]:7.1f} {ts[2]:7.1f}  -> {B / min(ts):5.2f} M env-steps/s | " +
          " ".join(f"{k}@{a:#x} (GiB {a >> 30}, mod 1GiB {((a & ((1 << 30) - 1)) >> 20)} MiB, mod 2MiB {(a & ((1 << 21) - 1)) >> 10} KiB)" for k, a in addr.items()), flush=True)
    env.close()
    del env
    torch.cuda.empty_cache()


run("default allocations #1")
run("default allocations #2")
pad = torch.empty(3 << 30, dtype=torch.uint8, device="cuda")  # shifts where the caching allocator puts the next segments
run("default, 3 GiB allocated before")
del pad; torch.cuda.empty_cache()
arena = torch.zeros((6 << 30), dtype=torch.uint8, device="cuda")
for label, gap in (("arena, tensors back to back", 0), ("arena, +4 KiB between tensors", 4096), ("arena, +64 KiB", 65536), ("arena, +1 MiB", 1 << 20),
                   ("arena, +2 MiB + 256 B", (2 << 20) + 256), ("arena, +33 MiB", 33 << 20)):
    off = [0]

    def alloc(name, shape, dtype, off=off, gap=gap):
        n = 1
        for v in shape:
            n *= v
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        if name not in BIG:
            return torch.zeros(shape, dtype=dtype, device="cuda")
        start = (off[0] + 255) & ~255
        off[0] = start + nbytes + gap
        v = arena[start:start + nbytes].view(dtype).view(shape)
        v.zero_()
        return v
    run(label, alloc)
