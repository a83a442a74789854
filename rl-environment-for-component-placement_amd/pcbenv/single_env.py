"""SingleEnvAdapter: one environment behind the reference's gym-0.22 surface.

`reset() -> obs dict`, `step(action_tuple) -> (obs, reward: float, done: bool, info: dict)`,
attributes `action_mask`, `grid`, `height`, `width`, `components`-free instance
access -- so loops written for the reference (`agent/random/random_policy_*.py`
`simulate()`, RLlib's `create_env`) run unchanged against the device path.  It is
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


class SingleEnvAdapter:
    is_batched = False

    def __init__(self, cfg: EnvConfig, seed: int = 0, device="cuda:0"):
        self.cfg = cfg
        self.height, self.width = cfg.height, cfg.width
        self._env = BatchedPlacementEnv(cfg, 1, device=device, queue_depth=1)
        self._stream = InstanceStream(cfg, seed)
        self.instance: Optional[Instance] = None
        self._dtype = np.float32 if cfg.kind == KIND_SQUARE else np.float64

    def _obs(self) -> Dict[str, np.ndarray]:
        return {k: v[0].cpu().numpy().astype(self._dtype) for k, v in self._env.obs.items()}

    def reset(self, instance: Optional[Instance] = None, verbose: bool = False) -> Dict[str, np.ndarray]:
        if self.cfg.kind != KIND_SQUARE:
            self.instance = instance if instance is not None else self._stream.next()
            self._env.load_instances([self.instance])
        self._env.reset()
        return self._obs()

    def step(self, action: Sequence[int], verbose: bool = False):
        a = torch.tensor([list(action)], dtype=torch.int32)
        self._env.step(a)
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
        """Read-only view with the reference's attribute names (`Component`: h, w, area, comp_id, placed, position,
        pins; `Pin`: relative_x/y, absolute_x/y, pin_id, component_id, net_id) built from the instance and the current
        feature tensors -- what `utils/agent/utils.py:238` reads to save a rollout."""
        from types import SimpleNamespace
        if self.instance is None:
            return []
        feat = self._env.obs["all_components_feature"][0].cpu().numpy()
        ins, out = self.instance, []
        spatial = self.cfg.kind == KIND_SPATIAL
        pins_num = self._env.obs["all_pins_num_feature"][0].cpu().numpy() if self.cfg.kind in (KIND_PIN, KIND_SPATIAL) else None
        for c in range(ins.num_components):
            x, y = int(feat[c, 2]), int(feat[c, 3])
            pins = []
            if spatial:  # rows are indexed by the global pin id, so current (rotated) coordinates can be read back
                for q in np.flatnonzero(ins.pin_comp == c):
                    r = pins_num[int(ins.pin_id[q])]
                    pins.append(SimpleNamespace(relative_x=int(r[0]), relative_y=int(r[1]), absolute_x=int(r[2]),
                                                absolute_y=int(r[3]), pin_id=int(ins.pin_id[q]), component_id=c,
                                                net_id=int(ins.pin_net[q])))
            out.append(SimpleNamespace(h=int(ins.comp_h[c]), w=int(ins.comp_w[c]), area=int(ins.comp_h[c] * ins.comp_w[c]),
                                       comp_id=c, placed=x >= 0, position=(x, y), pins=pins))
        return out

    def close(self):
        self._env.close()
