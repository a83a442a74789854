"""debug: small batch vs oracle with and without reward helpers"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rl-environment-for-component-placement_amd"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pcbenv import named_config
from pcbenv.batched_env import BatchedPlacementEnv
from oracle import oracle as orc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = named_config("c3")
for auto in (False, True):
    for teams in (None, 0):
        env = BatchedPlacementEnv(cfg, B, queue_depth=2, run_seed=11, auto_reset=auto, options=None if teams is None else {"terminal_teams": teams})
        packed = env.generate_instances(); env.reset()
        ob = orc.OracleBatch(cfg, B); ob.reset_packed(packed[0]); cursor = 0
        for t in range(34):
            a = env.sample_actions(t); env.step(a)
            rr, dd, ii = ob.step(a.cpu().numpy(), 4)
            r = env.reward.cpu().numpy(); d = env.done.cpu().numpy(); inf = env.info_raw.cpu().numpy()
            bad = np.flatnonzero((r.view(np.uint64) != rr.view(np.uint64)) | (d != dd))
            if len(bad):
                print("auto", auto, "teams", teams, "t", t, "bad envs", bad[:8], "got", r[bad[:4]], "want", rr[bad[:4]], "done", d[bad[:4]], dd[bad[:4]], "info", inf[bad[:2]].tolist(), ii[bad[:2]].tolist(), flush=True)
            if dd.any():
                cursor += 1
                ob.reset_packed(packed[cursor % 2], dd.astype(np.uint8), 4)
            if not auto: env.reset_done()
            for k in env.obs:
                bi = ob.first_mismatch(k, env.obs[k].cpu().numpy(), 4)
                if bi >= 0: print("auto", auto, "teams", teams, "t", t, "obs", k, "env", bi, flush=True)
        env.close()
print("done")
