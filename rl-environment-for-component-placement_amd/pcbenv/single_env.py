"""SingleEnvAdapter: one environment behind the reference's gym-0.22 surface.

`reset() -> obs dict`, `step(action_tuple) -> (obs, reward: float, done: bool, info: dict)`,
attributes `action_space`, `observation_space` (pcbenv/spaces.py: the reference constructors'
declarations), `action_mask`, `grid`, `height`, `width`, `components`, `actions` -- so loops
written for the reference (`agent/random/random_policy_*.py` `simulate()`, RLlib's
`create_env`) run unchanged against the device path.  It is
a B = 1 `BatchedPlacementEnv`; observations come back as NumPy arrays in the
reference's dtypes (float64; float32 for the square env) and are fresh copies,
like the reference's.  Instances come from the reference-exact stream
`InstanceStream(cfg, seed)` unless given explicitly.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch

from .batched_env import BatchedPlacementEnv
from .config import EnvConfig, KIND_PIN, KIND_SPATIAL, KIND_SQUARE
from .instances import Instance, InstanceStream
from .spaces import action_space_for, observation_space_for


class _PinView:
    __slots__ = ("_comp", "_k")

    def __init__(self, comp, k):
        self._comp, self._k = comp, k

    def _get(self, i):
        return self._comp._adapter._component_state(self._comp.comp_id)[2][self._k][i]

    relative_x = property(lambda s: s._get(0))
    relative_y = property(lambda s: s._get(1))
    absolute_x = property(lambda s: s._get(2))
    absolute_y = property(lambda s: s._get(3))
    pin_id = property(lambda s: s._get(4))
    net_id = property(lambda s: s._get(5))
    component_id = property(lambda s: s._comp.comp_id)


class _ComponentView:
    """Live stand-in for the reference's `Component` (attribute names of `dummy_env_rectangular_pin.py:108-148`)."""
    __slots__ = ("_adapter", "comp_id", "h", "w", "area", "pins")

    def __init__(self, adapter, c):
        ins = adapter.instance
        self._adapter, self.comp_id = adapter, c
        self.h, self.w = int(ins.comp_h[c]), int(ins.comp_w[c])
        self.area = self.h * self.w
        self.pins = [_PinView(self, k) for k in range(int(np.count_nonzero(ins.pin_comp == c)))] if ins.num_pins else []

    placed = property(lambda s: s._adapter._component_state(s.comp_id)[0])
    position = property(lambda s: s._adapter._component_state(s.comp_id)[1])


class SingleEnvAdapter:
    is_batched = False

    def __init__(self, cfg: EnvConfig, seed: int = 0, device="cuda:0"):
        self.cfg = cfg
        self.height, self.width = cfg.height, cfg.width
        self._env = BatchedPlacementEnv(cfg, 1, device=device, queue_depth=1)
        self._stream = InstanceStream(cfg, seed)
        self.instance: Optional[Instance] = None
        self._dtype = np.float32 if cfg.kind == KIND_SQUARE else np.float64
        self.action_space = action_space_for(cfg)
        self.observation_space = observation_space_for(cfg)
        self.actions = []      # every action passed to step() this episode (the spatial reference keeps this list, S:1574)
        self._placed = []      # (o, x, y) of the valid placements, in component order
        self._components, self._row_owner = [], {}

    def _obs(self) -> Dict[str, np.ndarray]:
        return {k: v[0].cpu().numpy().astype(self._dtype) for k, v in self._env.obs.items()}

    def reset(self, verbose: bool = False, *args, instance: Optional[Instance] = None, **kwargs) -> Dict[str, np.ndarray]:
        """`reset(verbose=False, *args, **kwargs)` like the reference (S:1487; `verbose` is its first positional and is
        ignored here as there is nothing to print); `instance=` (keyword only) replaces the next instance of the stream."""
        if isinstance(verbose, Instance):
            raise TypeError("pass the instance by keyword: reset(instance=...) -- the first positional argument is the reference's `verbose`")
        if self.cfg.kind != KIND_SQUARE:
            self.instance = instance if instance is not None else self._stream.next()
            self._env.load_instances([self.instance])
        self._env.reset()
        self.actions, self._placed = [], []
        self._components, self._row_owner = [], {}
        if self.instance is not None:
            ins = self.instance
            if self.cfg.kind == KIND_PIN:  # the last pin in self.pins order owns the feature row [component, pin_id]
                for q in range(ins.num_pins):
                    self._row_owner[(int(ins.pin_comp[q]), int(ins.pin_id[q]))] = q
            self._components = [_ComponentView(self, c) for c in range(ins.num_components)]
        return self._obs()

    def step(self, action: Sequence[int], verbose: bool = False):
        a = torch.tensor([list(action)], dtype=torch.int32)
        valid = self.validate_action(*action)
        self._env.step(a)
        self.actions.append(tuple(int(v) for v in action))
        if valid and self.cfg.kind != KIND_SQUARE:
            self._placed.append(tuple(int(v) for v in action))
        reward = float(self._env.reward[0].item())
        done = bool(self._env.done[0].item())
        info = {}
        if self.cfg.kind in (KIND_PIN, KIND_SPATIAL):
            raw = self._env.info_raw[0].cpu().numpy()
            if not np.isnan(raw[0]):
                info = {"wirelength": float(raw[0]), "num_intersections": float(raw[1])}
        return self._obs(), reward, done, info

    def validate_action(self, *action) -> bool:
        m = self.action_mask
        try:
            if any(int(v) < 0 for v in action):
                return False
            return bool(m[tuple(int(v) for v in action)] == 1)
        except IndexError:
            return False

    @property
    def action_mask(self) -> np.ndarray:
        return self._env.obs["action_mask"][0].cpu().numpy().astype(self._dtype)

    @property
    def grid(self) -> np.ndarray:
        return self._env.obs["grid"][0].cpu().numpy().astype(self._dtype)

    @property
    def components(self):
        """The reference's `env.components`: one object per component with the reference's attribute names
        (`Component`: h, w, area, comp_id, placed, position, pins; `Pin`: relative_x/y, absolute_x/y, pin_id,
        component_id, net_id).  Like the reference's list it is created by `reset()` and its objects are LIVE: a
        caller that keeps the list from right after `reset()` (`utils/agent/utils.py:238` does) sees the placements
        when the episode is over.  Every attribute read goes to the device tensors."""
        return self._components

    def _component_state(self, c: int):
        """(placed, (x, y), [(rel_x, rel_y, abs_x, abs_y, pin_id, net_id) ...]) of component c, now.  Sizes and positions
        come from the feature tensors on the device.  Pin coordinates are read back from `all_pins_num_feature`
        wherever a pin owns a row (spatial: row = global pin id; pin env: row [component, pin_id] unless a later pin
        of the component overwrote it, quirk Q1); the overwritten ones are recomputed from the instance and the
        recorded placement with `place_component`'s rotation (S:149-190)."""
        ins = self.instance
        feat = self._env.obs["all_components_feature"][0, c].cpu().numpy()
        x, y = int(feat[2]), int(feat[3])
        h, w = int(ins.comp_h[c]), int(ins.comp_w[c])
        pin_kind, spatial = self.cfg.kind in (KIND_PIN, KIND_SPATIAL), self.cfg.kind == KIND_SPATIAL
        pins = []
        if pin_kind:
            pins_num = self._env.obs["all_pins_num_feature"][0].cpu().numpy()
            for q in np.flatnonzero(ins.pin_comp == c):
                pid = int(ins.pin_id[q])
                if spatial or self._row_owner[(c, pid)] == q:
                    rx, ry, ax, ay = (int(v) for v in (pins_num[pid] if spatial else pins_num[c, pid]))
                else:
                    rx, ry, ax, ay = int(ins.pin_rel_x[q]), int(ins.pin_rel_y[q]), -1, -1
                    if c < len(self._placed):
                        o, px, py = self._placed[c]
                        if o == 1:
                            rx, ry = ry, h - rx - 1
                        elif o == 2:
                            rx, ry = h - rx - 1, w - ry - 1
                        elif o == 3:
                            rx, ry = w - ry - 1, rx
                        ax, ay = px + rx, py + ry
                pins.append((rx, ry, ax, ay, pid, int(ins.pin_net[q])))
        return x >= 0, (x, y), pins

    def close(self):
        self._env.close()
