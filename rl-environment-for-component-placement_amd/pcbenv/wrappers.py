"""Flat Discrete action / flat action-mask adapters.

The reference wraps its environments (`utils/environment/env_wrappers.py`,
`environment/wrapper_square.py`) so that RLlib sees one `Discrete(O*H*W)` action
and a 1-D mask:

* action  `a -> (a // (H*W), (a % (H*W)) // W, a % W)`   (`env_wrappers.py:80-98`;
  square `divmod(a, W)` :184-199) -- on the device this is `PCBENV_ACTION_FLAT`,
  decoded inside the step kernel, so the wrapper only forwards the 1-D tensor;
* mask    C-order ravel of `[O, H, W]`                   (`env_wrappers.py:35-51`)
  -- a zero-copy `view` of the contiguous uint8 tensor.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

from .spaces import flat_action_space, flat_mask_observation_space


def flat_to_tuple(action: int, num_orientations: int, height: int, width: int) -> Tuple[int, ...]:
    """Index math of FlatteningActionWrapperRect.action / ...Square.action."""
    if num_orientations == 1:
        return tuple(int(v) for v in divmod(int(action), width))
    o, rem = divmod(int(action), height * width)
    x, y = divmod(rem, width)
    return int(o), int(x), int(y)


def tuple_to_flat(action, num_orientations: int, height: int, width: int) -> int:
    if num_orientations == 1:
        x, y = action[-2], action[-1]
        return int(x) * width + int(y)
    o, x, y = action
    return (int(o) * height + int(x)) * width + int(y)


class _Wrapper:
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        return getattr(self.env, name)


class FlatteningActionMaskObservationWrapper(_Wrapper):
    """`action_mask` flattened to `[..., O*H*W]` (a view, no copy); `observation_space` likewise
    (`env_wrappers.py:28-31`)."""

    def __init__(self, env):
        super().__init__(env)
        self.observation_space = flat_mask_observation_space(env.observation_space)
        self.action_space = env.action_space

    def _obs(self, obs):
        out = dict(obs)
        m = obs["action_mask"]
        batched = m.dim() if hasattr(m, "dim") else m.ndim
        lead = 1 if getattr(self.env, "is_batched", False) else 0
        out["action_mask"] = m.reshape(m.shape[:lead] + (-1,)) if batched else m
        return out

    def reset(self, *a, **k):
        return self._obs(self.env.reset(*a, **k))

    def step(self, action):
        res = self.env.step(action)
        return (self._obs(res[0]),) + tuple(res[1:])


class FlatteningActionWrapper(_Wrapper):
    """Accepts flat actions.  Batched: an int tensor `[B]` goes to the kernel as PCBENV_ACTION_FLAT.
    Single env: a Python int is decoded with the reference's divmod chain.  `action_space` is
    `Discrete(prod(factor sizes))` (`env_wrappers.py:76-78`)."""

    def __init__(self, env):
        super().__init__(env)
        self.factor_sizes = [sp.n for sp in env.action_space.spaces]
        self.action_space = flat_action_space(env.action_space)
        self.observation_space = env.observation_space

    def action(self, action):
        cfg = self.env.cfg
        return flat_to_tuple(action, cfg.num_orientations, cfg.height, cfg.width)

    def step(self, action):
        if getattr(self.env, "is_batched", False):
            return self.env.step(action)  # 1-D tensor -> flat format
        return self.env.step(self.action(int(action)))

    def validate_action(self, action) -> bool:
        return self.env.validate_action(*self.action(int(action)))

    @property
    def num_actions(self) -> int:
        cfg = self.env.cfg
        return int(np.prod([cfg.num_orientations, cfg.height, cfg.width]))
