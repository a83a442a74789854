"""On-disk formats for instances and episodes.

* `save_instances / load_instances`: a compact binary of the instance tables -- exactly the wire records of
  `include/pcbenv.h` behind a small header (magic, version, the constructor arguments as JSON, count, stride).
  A file written here can be fed to `pcbenv_load_instances` byte for byte, and re-imported as a test fixture.
* `episode_to_reference_objects`: the content of the reference's `components.pkl` / `actions.pkl`
  (`utils/visualization/csv_utils.py:11-25`, read by `web_app/visualization_grid.py:13-72`): given the
  reference's own `Component` / `Pin` classes (the caller imports them inside the reference tree; this package
  never imports the reference) it builds the component list in the state after the episode and the action list,
  so an episode stepped on the GPU can be pickled for the reference's renderer.
"""
from __future__ import annotations

import json
import struct
from dataclasses import asdict
from typing import List, Sequence, Tuple

import numpy as np

from .config import EnvConfig
from .instances import Instance, instance_stride, pack_instances, unpack_instances

MAGIC = b"PCBI"
VERSION = 1


def save_instances(path: str, cfg: EnvConfig, instances) -> None:
    """instances: list of Instance or packed uint8 [n, instance_stride(cfg)]."""
    packed = instances if isinstance(instances, np.ndarray) else pack_instances(cfg, instances)
    packed = np.ascontiguousarray(packed, np.uint8)
    assert packed.ndim == 2 and packed.shape[1] == instance_stride(cfg)
    meta = json.dumps(asdict(cfg)).encode()
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<IIQQ", VERSION, len(meta), packed.shape[0], packed.shape[1]))
        f.write(meta)
        f.write(packed.tobytes())


def load_instances(path: str) -> Tuple[EnvConfig, np.ndarray]:
    with open(path, "rb") as f:
        if f.read(4) != MAGIC:
            raise ValueError("not a pcbenv instance file")
        version, mlen, n, stride = struct.unpack("<IIQQ", f.read(24))
        if version != VERSION:
            raise ValueError(f"unsupported version {version}")
        cfg = EnvConfig(**json.loads(f.read(mlen).decode()))
        if stride != instance_stride(cfg):
            raise ValueError("record stride does not match the stored configuration")
        data = np.frombuffer(f.read(n * stride), np.uint8)
        if data.size != n * stride:
            raise ValueError("truncated instance file")
    return cfg, data.reshape(n, stride).copy()


def episode_to_reference_objects(instance: Instance, actions: Sequence[Sequence[int]], Component, Pin):
    """-> (components, actions) as the reference's `save_to_file` pickles them.  `Component` / `Pin` are the
    classes of environment/dummy_env_rectangular_pin.py (or the spatial twin); placement replays the reference's
    own `place_component` (rotation of the pins included) for every valid action in order."""
    pins_by_comp: List[list] = [[] for _ in range(instance.num_components)]
    for rx, ry, net, comp, pid in zip(instance.pin_rel_x, instance.pin_rel_y, instance.pin_net, instance.pin_comp,
                                     instance.pin_id):
        pins_by_comp[int(comp)].append(Pin(int(rx), int(ry), int(pid), int(comp), int(net)))
    comps = [Component(int(h), int(w), i, pins_by_comp[i]) for i, (h, w) in enumerate(zip(instance.comp_h, instance.comp_w))]
    acts = [tuple(int(v) for v in a) for a in actions]
    for comp, (o, x, y) in zip(comps, acts):
        comp.place_component(o, x, y)
    return comps, acts
