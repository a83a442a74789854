"""debug: c3 x B lock-step, the sequence of test_full_size_batches; which environments disagree at the terminal step"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rl-environment-for-component-placement_amd"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pcbenv import named_config
from pcbenv.batched_env import BatchedPlacementEnv
from oracle import oracle as orc
B = 4096
cfg = named_config("c3")
P = {"terminal_teams": 0}
envs = {"a": BatchedPlacementEnv(cfg, B, queue_depth=2, run_seed=11),
        "b": BatchedPlacementEnv(cfg, B, queue_depth=2, run_seed=11, auto_reset=True, incremental_obs=True),
        "a_plain": BatchedPlacementEnv(cfg, B, queue_depth=2, run_seed=11, options=P),
        "b_plain": BatchedPlacementEnv(cfg, B, queue_depth=2, run_seed=11, auto_reset=True, incremental_obs=True, options=P),
        "b_noinc": BatchedPlacementEnv(cfg, B, queue_depth=2, run_seed=11, auto_reset=True)}
packed = None
for e in envs.values():
    packed = e.generate_instances(); e.reset()
ob = orc.OracleBatch(cfg, B)
ob.reset_packed(packed[0])
cursor = 0
for t in range(34):
    acts = {}
    pos = {}
    for name, e in envs.items():
        pos[name] = np.zeros((B, 2), np.uint32)
        if name.startswith("a"):
            acts[name] = e.sample_actions(t); e.step(acts[name])
        else:
            acts[name] = e.rollout_step(t)[4]
    rr, dd, _ = ob.step(acts["a_plain"].cpu().numpy(), 16)
    for name, e in envs.items():
        if not torch.equal(acts[name], acts["a_plain"]):
            bad = (acts[name] != acts["a_plain"]).any(1).nonzero().flatten().cpu().numpy()
            print(t, name, "ACTIONS differ in", len(bad), bad[:8], acts[name][bad[:2]].tolist(), acts["a_plain"][bad[:2]].tolist(), flush=True)
        r = e.reward.cpu().numpy(); d = e.done.cpu().numpy()
        bad = np.flatnonzero((r.view(np.uint64) != rr.view(np.uint64)) | (d != dd))
        if len(bad):
            print(t, name, "vs ORACLE reward/done mismatch in", len(bad), bad[:8], r[bad[:4]], rr[bad[:4]], d[bad[:4]], dd[bad[:4]], "marks", pos[name][bad[:4]].tolist(), flush=True)
    if dd.any():
        cursor += 1
        ob.reset_packed(packed[cursor % 2], dd.astype(np.uint8), 16)
    for name, e in envs.items():
        if not e.auto_reset: e.reset_done()
    for name, e in envs.items():
        for k in e.obs:
            badi = ob.first_mismatch(k, e.obs[k].cpu().numpy(), 16)
            if badi >= 0: print(t, name, "obs", k, "first mismatching env", badi, flush=True)
print("done")
