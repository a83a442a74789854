import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "rl-environment-for-component-placement_amd")
import numpy as np, torch
from golden_util import load_case
from pcbenv.batched_env import BatchedPlacementEnv
meta, cfg, eps = load_case(sys.argv[1] if len(sys.argv) > 1 else "spatial_c4_beam")
env = BatchedPlacementEnv(cfg, len(eps), queue_depth=1)
env.load_instances([e.instance for e in eps])
env.reset()
acts = np.zeros((len(eps), 3), np.int32)
for i, e in enumerate(eps): acts[i] = e.actions[0]
o, r, d, info = env.step(torch.from_numpy(acts))
got = o["pin_grid"][0].cpu().numpy().astype(int); want = eps[0].obs["pin_grid"][1].astype(int)
bad = np.argwhere(got != want)
print("action", acts[0], "mismatches", len(bad))
for b in bad[:8]: print(tuple(b), "flat byte", (b[0]*64+b[1])*9+b[2], "got", got[tuple(b)], "want", want[tuple(b)])
x0 = max(bad[0][0]-1, 0)
for x in range(x0, x0+3):
    print("row", x, "want classes:", [(y, int(np.argmax(want[x, y]))+1 if want[x, y].any() else 0) for y in range(36, 48)])
    print("row", x, "got  classes:", [(y, int(np.argmax(got[x, y]))+1 if got[x, y].any() else 0) for y in range(36, 48)])
