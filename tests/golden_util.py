"""Loader for tests/golden/*.npz (format: tests/golden/make_golden.py)."""
import glob
import json
import os
from types import SimpleNamespace

import numpy as np

from pcbenv import EnvConfig, Instance

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BINARY_KEYS = ("grid", "action_mask", "pin_grid", "component_grid")
MAKE = {"square": EnvConfig.square, "rect": EnvConfig.rect, "pin": EnvConfig.pin, "spatial": EnvConfig.spatial}


def case_names():
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in names if n not in ("norm2", "setorder", "adapter_views")]


def load_case(name):
    """-> (meta, cfg, episodes); an episode has .seed .ep .instance .actions .reward .done .info .obs
    (obs[key] is a float64 stack [T+1, ...], index 0 = after reset)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    cfg = MAKE[meta["kind"]](*meta["args"])
    eps = []
    for seed in meta["seeds"]:
        for ep in range(meta["episodes"]):
            pre = f"s{seed}_e{ep}_"
            if meta["kind"] == "square":
                inst = None
            elif meta["kind"] == "rect":
                inst = Instance(z[pre + "comp_h"].astype(np.int64), z[pre + "comp_w"].astype(np.int64))
            else:
                inst = Instance(*(z[pre + k].astype(np.int64) if k != "num_nets" else int(z[pre + k]) for k in (
                    "comp_h", "comp_w", "num_nets", "pin_rel_x", "pin_rel_y", "pin_net", "pin_comp", "pin_id")))
            obs = {}
            for k in z.files:
                if not k.startswith(pre + "obs_"):
                    continue
                key = k[len(pre) + 4:]
                if key.endswith("_bits"):
                    key = key[:-5]
                    shape = tuple(z[pre + "obs_" + key + "_shape"])
                    obs[key] = np.unpackbits(z[k])[:int(np.prod(shape))].reshape(shape).astype(np.float64)
                elif not key.endswith("_shape"):
                    obs[key] = z[k]
            eps.append(SimpleNamespace(seed=seed, ep=ep, instance=inst, actions=z[pre + "actions"].astype(np.int64),
                                       reward=z[pre + "reward"], done=z[pre + "done"].astype(bool),
                                       info=z[pre + "info"], obs=obs))
    return meta, cfg, eps


def pad_component_grid(ref, max_components):
    """The reference's component_grid has only len(components) rows; batched layouts pad with zeros."""
    if ref.shape[0] == max_components:
        return ref
    out = np.zeros((max_components,) + ref.shape[1:], ref.dtype)
    out[:ref.shape[0]] = ref
    return out
