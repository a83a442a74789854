#!/usr/bin/env python3
"""Record golden episodes from the UNMODIFIED reference environments.

Run in the build container only (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py

It imports `/root/reference/environment/dummy_env_*.py` with the stand-in `gym`
package of oracle/refshim first on sys.path (gym is used by those files only for
the `gym.Env` base class and `gym.spaces` declarations), seeds the reference's two
global RNG streams (`np.random.seed(s); random.seed(s)`), and replays action
lists drawn from a *separate* `random.Random`, so the instance streams stay
action-independent.  Per case it stores, in `tests/golden/<case>.npz`:

* the constructor arguments and seed (JSON),
* per episode the instance tables the reference generated (component h/w; per
  pin rel_x, rel_y, net, component, pin_id in `env.pins` order),
* per step the action, reward (float64), done, info values,
* the full observation after reset and after every step (0/1 arrays bit-packed).

Also written: `spaces.json` (the gym spaces every reference constructor declares), `model_config_spatial.json` (the
shipped hyper-parameters of the spatial policy, a data file of the reference), `adapter_views.npz` (`env.components`
with all pin coordinates after whole episodes), `norm2.npz` (np.linalg.norm of length-2 vectors in this container's
NumPy/OpenBLAS -- SURVEY.md trap T1) and `setorder.npz` (CPython iteration order of
`set(points) - visited` -- trap T2).  The files are data only; no reference source
text is stored.
"""
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REPO, "oracle", "refshim"))
sys.path.insert(0, "/root/reference")

from environment.dummy_env_rectangular import DummyPlacementEnv as RefRect  # noqa: E402
from environment.dummy_env_rectangular_pin import DummyPlacementEnv as RefPin  # noqa: E402
from environment.dummy_env_rectangular_pin_spatial import DummyPlacementEnv as RefSpatial  # noqa: E402
from environment.dummy_env_square import DummyPlacementEnv as RefSquare  # noqa: E402

REF = {"square": RefSquare, "rect": RefRect, "pin": RefPin, "spatial": RefSpatial}
BINARY_KEYS = ("grid", "action_mask", "pin_grid", "component_grid")

C3 = (64, 64, 9, 9, 2, 6, 2, 6, 16, 16, 8, 8, 6, 6)
C5 = (128, 128, 9, 9, 2, 8, 2, 8, 32, 32, 16, 16, 8, 8)
SMALL = (10, 10, 3, 4, 2, 4, 2, 4, 6, 1, 2, 4, 5, 2)
MID = (12, 12, 5, 5, 2, 5, 2, 5, 8, 6, 3, 5, 7, 2)

# name, kind, ctor args, seeds, episodes per seed, probability of a random (mostly invalid) action
CASES = [
    ("square_c1", "square", (8, 8, 3), [0, 1, 2], 2, 0.05),
    ("square_11x10_n2", "square", (11, 10, 2), [0, 1], 2, 0.05),
    ("square_5x5_n1", "square", (5, 5, 1), [0], 2, 0.0),
    ("rect_c2", "rect", (32, 32, 2, 6, 2, 6, 8, 8), [0, 1, 2], 2, 0.03),
    ("rect_6x6", "rect", (6, 6, 2, 4, 2, 4, 4, 2), [0, 1, 2, 3, 4, 5], 3, 0.05),
]
for rt in ("centroid", "beam", "both"):
    CASES += [
        (f"pin_small_{rt}", "pin", SMALL + (rt, 2, 0.5), list(range(8)), 3, 0.05),
        (f"spatial_small_{rt}", "spatial", SMALL + (rt, 2, 0.5), list(range(8)), 3, 0.05),
        (f"pin_c3_{rt}", "pin", C3 + (rt, 2, 0.5), [0, 1], 2, 0.01),
        (f"spatial_c4_{rt}", "spatial", C3 + (rt, 2, 0.5), [0, 1], 2, 0.01),
    ]
CASES += [
    ("spatial_c5_both_k3", "spatial", C5 + ("both", 3, 0.5), [0], 1, 0.0),
    ("spatial_c5_centroid", "spatial", C5 + ("centroid", 2, 0.5), [1], 1, 0.0),
    ("spatial_mid_both_k4", "spatial", MID + ("both", 4, 0.25), list(range(6)), 3, 0.03),
    ("pin_mid_beam_k1", "pin", MID + ("beam", 1, 0.25), list(range(6)), 3, 0.03),
]


def tables(env, kind):
    out = {"comp_h": np.array([c.h for c in env.components], np.int16),
           "comp_w": np.array([c.w for c in env.components], np.int16)}
    if kind in ("pin", "spatial"):
        pins = env.pins
        out.update(num_nets=np.array(len(env.net_pins), np.int16),
                   pin_rel_x=np.array([p.relative_x for p in pins], np.int16),
                   pin_rel_y=np.array([p.relative_y for p in pins], np.int16),
                   pin_net=np.array([p.net_id for p in pins], np.int16),
                   pin_comp=np.array([p.component_id for p in pins], np.int16),
                   pin_id=np.array([p.pin_id for p in pins], np.int16))
        # the reference's net_pins[n] must be the contiguous net-major runs of env.pins
        flat = [p for n in range(len(env.net_pins)) for p in env.net_pins[n]]
        assert all(a is b for a, b in zip(flat, pins)) and len(flat) == len(pins)
    return out


def record_case(name, kind, args, seeds, episodes, p_random):
    data = {"meta": np.array(json.dumps({"name": name, "kind": kind, "args": list(args), "seeds": list(seeds),
                                         "episodes": episodes}))}
    for seed in seeds:
        np.random.seed(seed)
        random.seed(seed)
        env = REF[kind](*args)
        arng = random.Random(seed + 12345)
        for ep in range(episodes):
            pre = f"s{seed}_e{ep}_"
            obs_list = [env.reset()]
            if kind != "square":
                for k, v in tables(env, kind).items():
                    data[pre + k] = v
            actions, rewards, dones, infos = [], [], [], []
            done = False
            while not done:
                m = env.action_mask
                valid = np.argwhere(m == 1)
                if arng.random() < p_random or len(valid) == 0:
                    act = tuple(arng.randrange(0, d + 2) for d in m.shape)  # may be out of range
                else:
                    act = tuple(int(v) for v in valid[arng.randrange(len(valid))])
                obs, r, done, info = env.step(act)
                obs_list.append(obs)
                actions.append(act if kind != "square" else (0,) + act)
                rewards.append(r)
                dones.append(done)
                infos.append([info.get("wirelength", np.nan), info.get("num_intersections", np.nan)])
            if kind != "square":  # one more step after the terminal one (reference keeps no done latch)
                act = (0, 0, 0)
                obs, r, d, info = env.step(act)
                obs_list.append(obs)
                actions.append(act)
                rewards.append(r)
                dones.append(d)
                infos.append([info.get("wirelength", np.nan), info.get("num_intersections", np.nan)])
            data[pre + "actions"] = np.array(actions, np.int16)
            data[pre + "reward"] = np.array(rewards, np.float64)
            data[pre + "done"] = np.array(dones, np.uint8)
            data[pre + "info"] = np.array(infos, np.float64)
            for k in obs_list[0]:
                stack = np.stack([np.asarray(o[k], np.float64) for o in obs_list])
                if k in BINARY_KEYS:
                    assert np.isin(stack, (0.0, 1.0)).all()
                    data[pre + "obs_" + k + "_shape"] = np.array(stack.shape, np.int32)
                    data[pre + "obs_" + k + "_bits"] = np.packbits(stack.astype(np.uint8).ravel())
                else:
                    data[pre + "obs_" + k] = stack
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **data)
    return os.path.getsize(path)


def record_norm2():
    rng = np.random.RandomState(7)
    rows = []
    for _ in range(4000):  # pin-to-centroid style: integer point minus k-th fractions
        n = rng.randint(3, 9)
        pts = rng.randint(0, 128, size=(n, 2))
        c = np.mean(pts, axis=0)
        for p in pts:
            d = np.array(p) - np.array(c)
            rows.append((d[0], d[1], np.linalg.norm(d)))
    for _ in range(2000):  # integer differences
        d = rng.randint(-128, 129, size=2).astype(np.float64)
        rows.append((d[0], d[1], np.linalg.norm(d)))
    np.savez_compressed(os.path.join(HERE, "norm2.npz"), rows=np.array(rows, np.float64))


def record_setorder():
    rng = random.Random(11)
    pts_all, masks, orders, hashes = [], [], [], []
    for _ in range(3000):
        n = rng.randrange(1, 16)
        side = rng.choice([6, 10, 64, 128])
        pts = []
        while len(pts) < n:
            p = (rng.randrange(side), rng.randrange(side))
            if p not in pts:
                pts.append(p)
        k = rng.randrange(0, n + 1)
        vis_idx = rng.sample(range(n), k)
        to_visit = set(pts)
        visited = set()
        for i in vis_idx:  # built like beam_search does: visited | {neighbor}
            visited = visited | {pts[i]}
        order = list(to_visit - visited)
        row = np.full((15, 2), -1, np.int16)
        row[:n] = pts
        orow = np.full(15, -1, np.int16)
        orow[:len(order)] = [pts.index(p) for p in order]
        pts_all.append(row)
        masks.append(sum(1 << i for i in vis_idx))
        orders.append(orow)
    for _ in range(500):
        x, y = rng.randrange(0, 200), rng.randrange(0, 200)
        hashes.append((x, y, hash((x, y)) & 0xFFFFFFFFFFFFFFFF))
    np.savez_compressed(os.path.join(HERE, "setorder.npz"), points=np.array(pts_all), visited_mask=np.array(masks, np.int64),
                        order=np.array(orders), tuple_hash=np.array(hashes, np.uint64))


def _describe_space(sp):
    """JSON summary of one of the reference's space objects (the stand-in gym classes keep the ctor arguments)."""
    from gym import spaces as gs
    if isinstance(sp, gs.Discrete):
        return {"type": "Discrete", "n": int(sp.n)}
    if isinstance(sp, gs.Tuple):
        return {"type": "Tuple", "spaces": [_describe_space(x) for x in sp.spaces]}
    if isinstance(sp, gs.Box):
        return {"type": "Box", "low": float(np.min(sp.low)), "high": float(np.max(sp.high)),
                "shape": [int(v) for v in sp.shape], "dtype": str(np.dtype(sp.dtype))}
    if isinstance(sp, gs.Dict):
        return {"type": "Dict", "spaces": {k: _describe_space(v) for k, v in sp.spaces.items()}}
    raise TypeError(sp)


def record_spaces():
    """action_space / observation_space exactly as each reference constructor declares them, for every case above
    (`tests/golden/spaces.json`; compared with pcbenv/spaces.py by tests/test_spaces.py)."""
    out = {}
    for name, kind, args, seeds, _eps, _p in CASES:
        np.random.seed(seeds[0])
        random.seed(seeds[0])
        env = REF[kind](*args)
        out[name] = {"kind": kind, "args": list(args), "action_space": _describe_space(env.action_space),
                     "observation_space": _describe_space(env.observation_space)}
    with open(os.path.join(HERE, "spaces.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def record_adapter_views():
    """What callers read off the environment OBJECT rather than the observation: `env.components` (with every pin's
    coordinates, what utils/agent/utils.py:238 pickles) and `env.action_mask` after whole episodes, for the small
    pin / spatial cases (`tests/golden/adapter_views.npz`; replayed on the HIP path through SingleEnvAdapter)."""
    data = {}
    meta = []
    for name, kind, args, seeds, episodes, _p in CASES:
        if name not in ("pin_small_centroid", "spatial_small_centroid", "pin_mid_beam_k1"):
            continue
        meta.append({"name": name, "kind": kind, "args": list(args), "seeds": list(seeds[:4]), "episodes": 2})
        for seed in seeds[:4]:
            np.random.seed(seed)
            random.seed(seed)
            env = REF[kind](*args)
            arng = random.Random(seed + 777)
            for ep in range(2):
                pre = f"{name}_s{seed}_e{ep}_"
                env.reset()
                for k, v in tables(env, kind).items():
                    data[pre + k] = v
                actions, done = [], False
                while not done:
                    valid = np.argwhere(env.action_mask == 1)
                    act = tuple(int(v) for v in valid[arng.randrange(len(valid))])
                    _, _, done, _ = env.step(act)
                    actions.append(act)
                data[pre + "actions"] = np.array(actions, np.int16)
                data[pre + "comp_state"] = np.array([[c.h, c.w, c.area, c.comp_id, int(c.placed), c.position[0], c.position[1]]
                                                     for c in env.components], np.int16)
                data[pre + "pin_state"] = np.array([[c.comp_id, p.relative_x, p.relative_y, p.absolute_x, p.absolute_y,
                                                     p.pin_id, p.component_id, p.net_id]
                                                    for c in env.components for p in c.pins], np.int16).reshape(-1, 8)
                data[pre + "action_mask_sum"] = np.array(env.action_mask.sum(), np.float64)
    data["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "adapter_views.npz"), **data)


def record_model_config():
    """The hyper-parameters the reference ships for the spatial policy (`agent/config/rectangle_pin_spatial_model.json`,
    a data file): `tests/golden/model_config_spatial.json`, compared with pcbenv/policy.py's defaults by
    tests/test_policy_cpu.py."""
    with open("/root/reference/agent/config/rectangle_pin_spatial_model.json") as f:
        ref = json.load(f)
    with open(os.path.join(HERE, "model_config_spatial.json"), "w") as f:
        json.dump({"env_config": ref["env_config"], "custom_model_config": ref["model"]["custom_model_config"]}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "model_config":  # only the (tiny) model-config fixture
        record_model_config()
        raise SystemExit(0)
    total = 0
    for case in CASES:
        sz = record_case(*case)
        total += sz
        print(f"{case[0]:28s} {sz / 1024:8.1f} KiB")
    record_norm2()
    record_setorder()
    record_spaces()
    record_adapter_views()
    record_model_config()
    print(f"total {total / 1024:.1f} KiB")
