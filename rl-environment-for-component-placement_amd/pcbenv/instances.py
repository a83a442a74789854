"""Host-side instance generator: the random placement problems `reset()` draws.

Restates, in array form, the generator of the reference environments

* rect   : `environment/dummy_env_rectangular.py:253-273`   (`generate_instances`)
* pin    : `environment/dummy_env_rectangular_pin.py:1006-1265, 1476-1498`
* spatial: `environment/dummy_env_rectangular_pin_spatial.py:931-1212, 1408-1443`
           (`generate_instances`, `generate_components`, `sample_num_nets`,
           `sample_total_num_pins`, `allocate_pins_to_nets`,
           `sample_truncated_multinomial` :250-287,
           `allocate_pins_to_components[_for_net]`, `place_pins_on_component`)

The reference draws from the *global* legacy NumPy stream (`np.random.randint /
normal / multinomial`) and the *global* Python `random` stream
(`random.choice`).  One :class:`InstanceStream` owns a private
`np.random.RandomState(seed)` and `random.Random(seed)` and calls the *same
library methods in the same order* (SURVEY.md Appendix A), so stream `seed`
produces exactly the instances the reference produces after
`np.random.seed(seed); random.seed(seed)`; successive `next()` calls continue
both streams like successive `reset()` calls do.

Instances are action-independent, so they are generated ahead of time on the
host, packed into 8-byte records (:func:`pack_instances`, layout in
`include/pcbenv.h`) and queued in HBM; the `reset` kernel consumes them.
"""
from __future__ import annotations

import random
from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np

from .config import EnvConfig, KIND_PIN, KIND_RECT, KIND_SPATIAL, KIND_SQUARE


@dataclass
class Instance:
    """One placement problem (everything `reset()` randomises).

    Pins are stored in the order of the reference's `self.pins` list after
    `allocate_pins_to_components` (net-major: all pins of net 0, then net 1 ...;
    `..._spatial.py:1117-1119`).  `net_pins[n]` is therefore the contiguous run
    of pins with `pin_net == n`, and `component.pins` is the sub-sequence with
    `pin_comp == c` (`..._spatial.py:983-988`).
    """

    comp_h: np.ndarray  # (num_components,) int
    comp_w: np.ndarray
    num_nets: int = 0
    pin_rel_x: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    pin_rel_y: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    pin_net: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    pin_comp: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    pin_id: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))

    @property
    def num_components(self) -> int:
        return int(len(self.comp_h))

    @property
    def num_pins(self) -> int:
        return int(len(self.pin_net))


def _truncated_multinomial(rs, n: int, m: int, p: np.ndarray, k: int) -> List[int]:
    """`sample_truncated_multinomial` (`..._spatial.py:250-287`): m sequential
    single-trial multinomial draws, bins that reached k are zeroed and the
    probabilities renormalised before every draw."""
    if k < 1 or k > m:
        raise ValueError("Invalid value of k.")
    if not np.isclose(np.sum(p), 1):
        raise ValueError("Probabilities should sum to 1.")
    counts = np.zeros(n, dtype=int)
    for _ in range(m):
        q = p * (counts < k)
        q /= np.sum(q)
        counts += rs.multinomial(1, q)
    return counts.tolist()


class InstanceStream:
    """The sequence of instances one reference env produces over its resets."""

    def __init__(self, cfg: EnvConfig, seed: int):
        self.cfg = cfg
        self.seed = int(seed)
        self.rs = np.random.RandomState(self.seed)
        self.pr = random.Random(self.seed)

    # -- draw order: SURVEY.md Appendix A steps 1-2 ---------------------------
    def _components(self):
        c, rs = self.cfg, self.rs
        n = rs.randint(c.min_num_components, c.max_num_components + 1)
        hs, ws = [], []
        for _ in range(n):
            hs.append(int(rs.randint(c.min_component_h, c.max_component_h + 1)))
            ws.append(int(rs.randint(c.min_component_w, c.max_component_w + 1)))
        return np.asarray(hs, np.int64), np.asarray(ws, np.int64)

    def next(self) -> Instance:
        c, rs = self.cfg, self.rs
        if c.kind == KIND_SQUARE:
            return Instance(np.zeros(0, np.int64), np.zeros(0, np.int64))
        comp_h, comp_w = self._components()
        if c.kind == KIND_RECT:
            return Instance(comp_h, comp_w)

        ncomp = len(comp_h)
        areas = (comp_h * comp_w).tolist()
        total_area = int(sum(areas))
        # steps 3-4
        num_nets = min(int(rs.randint(c.min_num_nets, c.max_num_nets + 1)), int(total_area / 2))
        total_pins = min(
            int(rs.randint(c.min_num_pins_per_net * num_nets, c.max_num_pins_per_net * num_nets + 1)),
            total_area,
        )
        # step 5: softmax of normal samples (drawn even when unused)
        z = rs.normal(1 / num_nets, 1 / (c.net_distribution + 1), num_nets)
        p = np.exp(z) / np.sum(np.exp(z))
        # steps 6-7: creation ids -> nets
        lo = c.min_num_pins_per_net
        if lo * num_nets > total_pins:
            raise IndexError("reference raises IndexError: fewer pins than min_pins_per_net * num_nets")
        net_ids: List[List[int]] = [list(range(i * lo, (i + 1) * lo)) for i in range(num_nets)]
        cursor = lo * num_nets
        rem = total_pins - cursor
        if c.max_num_pins_per_net > lo and rem > 0:
            alloc = _truncated_multinomial(rs, num_nets, rem, p, min(c.max_num_pins_per_net - lo, rem))
            for i in range(num_nets):
                net_ids[i].extend(range(cursor, cursor + alloc[i]))
                cursor += alloc[i]
        # step 8
        if c.kind == KIND_SPATIAL:
            kcomp = min(int((c.pin_spread / 10) * ncomp) + 1, ncomp)
        else:
            kcomp = min(max(int(((c.pin_spread + 1) / 10) * ncomp), 1), ncomp)
        avail = {i: areas[i] for i in range(ncomp)}
        # step 9
        net_comp: List[List[int]] = []
        net_local: List[List[int]] = []  # pin env only: index within the multinomial batch (Q1)
        for net in range(num_nets):
            unassigned = len(net_ids[net])
            avail = dict(sorted(avail.items(), key=lambda kv: kv[1], reverse=True))
            order = list(avail.keys())
            k = kcomp - 1
            space = 0
            while space < unassigned:
                k += 1
                space = sum(avail[cid] for cid in order[:k])
            first = order[:k]
            comps_of_net: List[int] = []
            local_of_net: List[int] = []
            while unassigned > 0:
                tot = sum(avail[cid] for cid in first)
                cnt = rs.multinomial(unassigned, np.array([avail[cid] / tot for cid in first]))
                for cid, n in zip(first, cnt):
                    n = int(n)
                    if avail[cid] < n:
                        n = avail[cid]
                    avail[cid] -= n
                    comps_of_net.extend([cid] * n)
                    local_of_net.extend(range(n))
                    unassigned -= n
            net_comp.append(comps_of_net)
            net_local.append(local_of_net)
        # step 10: self.pins = concat(net_pins); per component random.choice of cells
        pin_net = np.concatenate([np.full(len(net_ids[n]), n, np.int64) for n in range(num_nets)])
        pin_comp = np.concatenate([np.asarray(net_comp[n], np.int64) for n in range(num_nets)])
        if c.kind == KIND_SPATIAL:
            pin_id = np.concatenate([np.asarray(net_ids[n], np.int64) for n in range(num_nets)])
        else:
            pin_id = np.concatenate([np.asarray(net_local[n], np.int64) for n in range(num_nets)])
        rel_x = np.full(len(pin_net), -1, np.int64)
        rel_y = np.full(len(pin_net), -1, np.int64)
        for cid in range(ncomp):
            w = int(comp_w[cid])
            cells = list(range(int(comp_h[cid]) * w))  # row-major (x, y) = divmod(cell, w)
            for j in np.flatnonzero(pin_comp == cid):
                cell = self.pr.choice(cells)
                cells.remove(cell)
                rel_x[j], rel_y[j] = divmod(cell, w)
        return Instance(comp_h, comp_w, num_nets, rel_x, rel_y, pin_net, pin_comp, pin_id)


def env_seed(run_seed: int, global_env_index: int) -> int:
    """Per-env stream seed: independent of how envs are sharded."""
    return 1_000_003 * int(run_seed) + int(global_env_index)


# ---------------------------------------------------------------------------
# packing: include/pcbenv.h `pcbenv_instance` wire format
# ---------------------------------------------------------------------------
INSTANCE_HEADER_BYTES = 16
RECORD_BYTES = 8


def instance_stride(cfg: EnvConfig) -> int:
    return INSTANCE_HEADER_BYTES + RECORD_BYTES * (cfg.max_num_components + cfg.max_total_pins)


def pack_instances(cfg: EnvConfig, instances: Sequence[Instance]) -> np.ndarray:
    """-> uint8 [n, instance_stride(cfg)] in the layout `include/pcbenv.h` documents."""
    C, P = cfg.max_num_components, cfg.max_total_pins
    out = np.zeros((len(instances), instance_stride(cfg)), np.uint8)
    hdr = out[:, :INSTANCE_HEADER_BYTES].view(np.int32)
    comps = out[:, INSTANCE_HEADER_BYTES:INSTANCE_HEADER_BYTES + RECORD_BYTES * C].reshape(len(instances), C, RECORD_BYTES)
    pins = out[:, INSTANCE_HEADER_BYTES + RECORD_BYTES * C:].reshape(len(instances), P, RECORD_BYTES)
    for i, ins in enumerate(instances):
        nc, npn = ins.num_components, ins.num_pins
        if nc > C or npn > P:
            raise ValueError(f"instance {i} exceeds config maxima ({nc}>{C} or {npn}>{P})")
        hdr[i, 0], hdr[i, 1], hdr[i, 2] = nc, ins.num_nets, npn
        comps[i, :nc, 0] = ins.comp_h
        comps[i, :nc, 1] = ins.comp_w
        if npn:
            if np.any(np.diff(ins.pin_net) < 0):
                raise ValueError("pins must be net-major (non-decreasing net id)")
            pins[i, :npn, 0] = ins.pin_rel_x
            pins[i, :npn, 1] = ins.pin_rel_y
            pins[i, :npn, 2] = ins.pin_net
            pins[i, :npn, 3] = ins.pin_comp
            pins[i, :npn, 4] = ins.pin_id & 0xFF
            pins[i, :npn, 5] = ins.pin_id >> 8
    return out


def unpack_instances(cfg: EnvConfig, packed: np.ndarray) -> List[Instance]:
    """Inverse of :func:`pack_instances`."""
    C, P = cfg.max_num_components, cfg.max_total_pins
    out = []
    for rec in np.ascontiguousarray(packed, np.uint8):
        nc, nn, npn = (int(v) for v in rec[:12].view(np.int32))
        comps = rec[INSTANCE_HEADER_BYTES:INSTANCE_HEADER_BYTES + RECORD_BYTES * C].reshape(C, RECORD_BYTES)
        pins = rec[INSTANCE_HEADER_BYTES + RECORD_BYTES * C:].reshape(P, RECORD_BYTES)[:npn].astype(np.int64)
        if cfg.kind == KIND_RECT:
            out.append(Instance(comps[:nc, 0].astype(np.int64), comps[:nc, 1].astype(np.int64)))
        else:
            out.append(Instance(comps[:nc, 0].astype(np.int64), comps[:nc, 1].astype(np.int64), nn, pins[:, 0],
                                pins[:, 1], pins[:, 2], pins[:, 3], pins[:, 4] | (pins[:, 5] << 8)))
    return out


class NativeInstanceStreams:
    """n independent instance streams advanced by libpcbenv.so's native generator (csrc/instance_gen.cpp),
    `threads` host threads.  Same tables as n `InstanceStream`s (see the caveat on exp() in that file);
    ~100x faster, which is what keeps a deep instance queue fed."""

    def __init__(self, cfg: EnvConfig, seeds: Sequence[int], threads: int = 8):
        import ctypes as C
        from . import _lib
        self.cfg, self.threads = cfg, int(threads)
        self._L = _lib.load()
        self._ccfg = _lib.make_config(cfg, max(1, len(seeds)))
        self._handles = (C.c_void_p * len(seeds))()
        for i, s in enumerate(seeds):
            h = C.c_void_p()
            _lib.check(self._L.pcbenv_instgen_create(C.byref(self._ccfg), int(s), C.byref(h)))
            self._handles[i] = h
        self.stride = instance_stride(cfg)

    def next_packed(self) -> np.ndarray:
        from . import _lib
        out = np.zeros((len(self._handles), self.stride), np.uint8)
        _lib.check(self._L.pcbenv_instgen_next_batch(self._handles, len(self._handles), out.ctypes.data, self.threads))
        return out

    def next(self) -> List[Instance]:
        return unpack_instances(self.cfg, self.next_packed())

    def __del__(self):
        for h in getattr(self, "_handles", ()):
            if h:
                self._L.pcbenv_instgen_destroy(h)
