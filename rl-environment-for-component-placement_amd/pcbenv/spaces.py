"""`action_space` / `observation_space` of the four environments and of the two flattening wrappers.

The reference declares gym-0.22 spaces in every constructor
(`environment/dummy_env_square.py:53-63`, `dummy_env_rectangular.py:193-224`,
`dummy_env_rectangular_pin.py:482-545`, `dummy_env_rectangular_pin_spatial.py:465-548`) and its wrappers rebuild them
(`utils/environment/env_wrappers.py:28-31` flattens the `action_mask` Box, `:76-78` turns the Tuple of Discretes
into `Discrete(prod(n))`); RLlib's `register_env(create_env)` reads both attributes off whatever `create_env`
returns.  gym is not installable here, so the descriptors are light classes of our own with the part of the gym-0.22
surface those callers use: `Discrete.n`, `Tuple.spaces`, `Box.low/high/shape/dtype`, `Dict.spaces` (a plain dict is
stored with its keys sorted, as gym 0.22 does), `contains` (`x in space`) and `sample`.  Bounds, shapes and dtypes
are the reference's, including its quirk that the spatial `all_pins_cat_feature` space is declared int32 while the
observation is float64 -- so, like the reference's, that space does not contain its own observation.
tests/golden/spaces.json holds the values read off the reference constructors (tests/golden/make_golden.py).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict as _Dict, Sequence

import numpy as np

from .config import EnvConfig, KIND_PIN, KIND_RECT, KIND_SPATIAL, KIND_SQUARE


class Space:
    shape: tuple = ()
    dtype = None

    def contains(self, x) -> bool:
        raise NotImplementedError

    def __contains__(self, x) -> bool:
        return self.contains(x)


class Discrete(Space):
    def __init__(self, n: int):
        assert n >= 0
        self.n = int(n)
        self.shape, self.dtype = (), np.dtype(np.int64)

    def contains(self, x) -> bool:
        if isinstance(x, (int, np.integer)):
            v = int(x)
        elif isinstance(x, np.ndarray) and x.shape == () and np.issubdtype(x.dtype, np.integer):
            v = int(x)
        else:
            return False
        return 0 <= v < self.n

    def sample(self, rng=np.random):
        return int(rng.randint(self.n))

    def __eq__(self, other):
        return isinstance(other, Discrete) and other.n == self.n

    def __repr__(self):
        return f"Discrete({self.n})"


class Tuple(Space):
    def __init__(self, spaces: Sequence[Space]):
        self.spaces = tuple(spaces)

    def contains(self, x) -> bool:
        if isinstance(x, (list, np.ndarray)):
            x = tuple(x)
        return isinstance(x, tuple) and len(x) == len(self.spaces) and all(s.contains(v) for s, v in zip(self.spaces, x))

    def sample(self, rng=np.random):
        return tuple(s.sample(rng) for s in self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def __len__(self):
        return len(self.spaces)

    def __eq__(self, other):
        return isinstance(other, Tuple) and other.spaces == self.spaces

    def __repr__(self):
        return "Tuple(" + ", ".join(map(repr, self.spaces)) + ")"


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(int(v) for v in shape)
        self.low = np.full(self.shape, low, dtype=self.dtype) if np.isscalar(low) else np.asarray(low, self.dtype).reshape(self.shape)
        self.high = np.full(self.shape, high, dtype=self.dtype) if np.isscalar(high) else np.asarray(high, self.dtype).reshape(self.shape)

    def contains(self, x) -> bool:  # gym 0.22 Box.contains
        if not isinstance(x, np.ndarray):
            try:
                x = np.asarray(x, dtype=self.dtype)
            except (ValueError, TypeError):
                return False
        return bool(np.can_cast(x.dtype, self.dtype) and x.shape == self.shape
                    and np.all(x >= self.low) and np.all(x <= self.high))

    def sample(self, rng=np.random):
        u = rng.uniform(self.low.astype(np.float64), self.high.astype(np.float64), size=self.shape)
        return (np.floor(u) if self.dtype.kind in "iu" else u).astype(self.dtype)

    def __eq__(self, other):
        return (isinstance(other, Box) and other.shape == self.shape and other.dtype == self.dtype
                and np.array_equal(other.low, self.low) and np.array_equal(other.high, self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


class Dict(Space):
    def __init__(self, spaces: _Dict[str, Space]):
        if isinstance(spaces, dict) and not isinstance(spaces, OrderedDict):
            spaces = OrderedDict(sorted(spaces.items()))  # gym 0.22 sorts the keys of a plain dict
        self.spaces = OrderedDict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]

    def __iter__(self):
        return iter(self.spaces)

    def keys(self):
        return self.spaces.keys()

    def items(self):
        return self.spaces.items()

    def contains(self, x) -> bool:
        return (isinstance(x, dict) and len(x) == len(self.spaces)
                and all(k in self.spaces and self.spaces[k].contains(v) for k, v in x.items()))

    def sample(self, rng=np.random):
        return OrderedDict((k, s.sample(rng)) for k, s in self.spaces.items())

    def __eq__(self, other):
        return isinstance(other, Dict) and list(other.spaces.items()) == list(self.spaces.items())

    def __repr__(self):
        return "Dict(" + ", ".join(f"{k}: {v!r}" for k, v in self.spaces.items()) + ")"


def flatten_space(space: Space) -> Space:
    """`gym.spaces.utils.flatten_space` for the one case the reference uses it on, a Box: same bounds and dtype,
    shape `(prod(shape),)` (`utils/environment/env_wrappers.py:28-31`)."""
    if not isinstance(space, Box):
        raise NotImplementedError("flatten_space: only Box spaces occur on this path")
    return Box(space.low.reshape(-1), space.high.reshape(-1), dtype=space.dtype)


def action_space_for(cfg: EnvConfig) -> Tuple:
    if cfg.kind == KIND_SQUARE:  # dummy_env_square.py:53-55
        return Tuple([Discrete(cfg.height), Discrete(cfg.width)])
    return Tuple([Discrete(cfg.num_orientations), Discrete(cfg.height), Discrete(cfg.width)])


def observation_space_for(cfg: EnvConfig) -> Dict:
    H, W, k = cfg.height, cfg.width, cfg.kind
    f64 = np.float64
    if k == KIND_SQUARE:  # dummy_env_square.py:56-63
        return Dict({"grid": Box(0, 1, (H, W), np.float32), "action_mask": Box(0, 1, (H, W), np.float32)})
    C = cfg.max_num_components
    if k == KIND_RECT:  # dummy_env_rectangular.py:203-224
        return Dict({"grid": Box(0, 1, (H, W), f64), "action_mask": Box(0, 1, (2, H, W), f64),
                     "all_components_feature": Box(-1, max(H, W), (C, 5), f64),
                     "component_mask": Box(0, 1, (C,), f64), "placement_mask": Box(0, 1, (C,), f64)})
    mp, N = cfg.max_num_pins_per_component, cfg.max_num_nets
    if k == KIND_PIN:  # dummy_env_rectangular_pin.py:504-545
        return Dict({"grid": Box(0, 1, (H, W), f64), "action_mask": Box(0, 1, (4, H, W), f64),
                     "all_components_feature": Box(-1, max(H, W), (C, 5), f64),
                     "all_pins_num_feature": Box(-1, max(H, W), (C, mp, 4), f64),
                     "all_pins_cat_feature": Box(-1, N, (C, mp, 1), f64),
                     "placement_mask": Box(0, 3, (C,), f64)})
    assert k == KIND_SPATIAL  # dummy_env_rectangular_pin_spatial.py:488-548
    return Dict({"grid": Box(0, 1, (H, W), f64), "pin_grid": Box(0, 100, (H, W, N + 1), f64),
                 "component_grid": Box(0, 1, (C, cfg.max_component_h, cfg.max_component_w, N + 1), f64),
                 "action_mask": Box(0, 1, (4, H, W), f64),
                 "all_components_feature": Box(-1, max(cfg.max_num_pins_per_net * N, H * W), (C, 5 + mp), f64),
                 "all_pins_num_feature": Box(-1, max(H, W), (C * mp + 1, 4), f64),
                 "all_pins_cat_feature": Box(-1, max(C, N), (C * mp + 1, 2), np.int32),
                 "placement_mask": Box(0, 3, (C,), f64)})


def flat_action_space(action_space: Tuple) -> Discrete:
    """`FlatteningActionWrapper*.__init__` (`env_wrappers.py:76-78`, :180-182): Discrete(prod of the factor sizes)."""
    return Discrete(int(np.prod([s.n for s in action_space.spaces])))


def flat_mask_observation_space(observation_space: Dict) -> Dict:
    """`FlatteningActionMaskObservationWrapper*.__init__` (`env_wrappers.py:28-31`): `action_mask` flattened."""
    return Dict(OrderedDict((k, flatten_space(v) if k == "action_mask" else v) for k, v in observation_space.spaces.items()))


def describe(space: Space):
    """JSON-able summary used by the golden fixture: bounds are the scalar extremes (the reference's Boxes are uniform)."""
    if isinstance(space, Discrete):
        return {"type": "Discrete", "n": space.n}
    if isinstance(space, Tuple):
        return {"type": "Tuple", "spaces": [describe(s) for s in space.spaces]}
    if isinstance(space, Box):
        return {"type": "Box", "low": float(np.min(space.low)), "high": float(np.max(space.high)),
                "shape": list(space.shape), "dtype": str(np.dtype(space.dtype))}
    if isinstance(space, Dict):
        return {"type": "Dict", "spaces": {k: describe(v) for k, v in space.spaces.items()}}
    raise TypeError(space)
