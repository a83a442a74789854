"""gym spaces of the boundary: `action_space` / `observation_space` of every environment kind and of the
two flattening wrappers against what the reference constructors declare (tests/golden/spaces.json, read off the
imported reference by tests/golden/make_golden.py:record_spaces), plus `contains` on the golden observations."""
import json
import os

import numpy as np
import pytest

from golden_util import GOLDEN_DIR, MAKE, case_names, load_case
from pcbenv.spaces import (Box, Dict, Discrete, Tuple, action_space_for, describe, flat_action_space,
                           flat_mask_observation_space, flatten_space, observation_space_for)

SPACES = json.load(open(os.path.join(GOLDEN_DIR, "spaces.json")))


@pytest.mark.parametrize("name", sorted(SPACES))
def test_spaces_equal_the_reference_constructors(name):
    rec = SPACES[name]
    cfg = MAKE[rec["kind"]](*rec["args"])
    assert describe(action_space_for(cfg)) == rec["action_space"]
    got, want = describe(observation_space_for(cfg)), rec["observation_space"]
    assert got["type"] == want["type"] == "Dict"
    assert set(got["spaces"]) == set(want["spaces"])
    for k in want["spaces"]:  # every key: bounds, shape, dtype
        assert got["spaces"][k] == want["spaces"][k], (name, k)


@pytest.mark.parametrize("name", sorted(SPACES))
def test_wrapper_spaces(name):
    """env_wrappers.py:28-31 (`flatten_space` of the action_mask Box) and :76-78 (`Discrete(prod(n))`)."""
    rec = SPACES[name]
    cfg = MAKE[rec["kind"]](*rec["args"])
    a = flat_action_space(action_space_for(cfg))
    assert isinstance(a, Discrete) and a.n == int(np.prod([s["n"] for s in rec["action_space"]["spaces"]]))
    o = flat_mask_observation_space(observation_space_for(cfg))
    m, ref = o["action_mask"], rec["observation_space"]["spaces"]["action_mask"]
    assert m.shape == (int(np.prod(ref["shape"])),) and str(m.dtype) == ref["dtype"]
    assert float(m.low.min()) == ref["low"] and float(m.high.max()) == ref["high"]
    for k in o.keys():
        if k != "action_mask":
            assert o[k] == observation_space_for(cfg)[k]
    assert list(o.keys()) == list(observation_space_for(cfg).keys())  # key order survives the wrapper


def test_space_semantics():
    d = Discrete(4)
    assert 0 in d and 3 in d and 4 not in d and -1 not in d and 1.0 not in d and np.int64(2) in d
    t = Tuple([Discrete(4), Discrete(8), Discrete(8)])
    assert (3, 7, 0) in t and [0, 0, 0] in t and (4, 0, 0) not in t and (0, 0) not in t and len(t) == 3 and t[1].n == 8
    b = Box(0, 1, (2, 3), np.float64)
    assert np.zeros((2, 3)) in b and np.ones((2, 3), np.float32) in b and np.full((2, 3), 2.0) not in b
    assert np.zeros((3, 2)) not in b
    assert np.zeros((2, 3), np.float64) not in Box(0, 1, (2, 3), np.int32)  # float64 does not cast safely to int32
    assert flatten_space(b).shape == (6,) and flatten_space(b).dtype == np.float64
    dd = Dict({"b": b, "a": d})
    assert list(dd.keys()) == ["a", "b"]  # gym 0.22 sorts a plain dict
    assert {"a": 1, "b": np.zeros((2, 3))} in dd and {"a": 1} not in dd
    rs = np.random.RandomState(0)
    for sp in (d, t, b, dd):
        assert sp.sample(rs) in sp


@pytest.mark.parametrize("name", [n for n in case_names() if "c5" not in n])
def test_golden_observations_lie_in_the_declared_spaces(name):
    """Every recorded reference observation is inside the declared space -- except where the reference's own
    declaration excludes its own observation (spatial all_pins_cat_feature is declared int32, Q7)."""
    meta, cfg, eps = load_case(name)
    space = observation_space_for(cfg)
    dtype = np.float32 if meta["kind"] == "square" else np.float64
    for e in eps[:3]:
        for k, stack in e.obs.items():
            if k == "component_grid" and stack.shape[1] != cfg.max_num_components:
                continue  # the reference allocates only len(components) rows (documented difference)
            ok = all(space[k].contains(np.asarray(o, dtype)) for o in stack)
            if meta["kind"] == "spatial" and k == "all_pins_cat_feature":
                assert not ok
            else:
                assert ok, (name, k)
