"""ctypes wrapper over oracle/libpcbenv_oracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg import
this module; the product package never does.  `OracleEnv` steps ONE environment
and returns observations in the reference's shapes and float64 values, so the
golden fixtures recorded from the reference compare with `==`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpcbenv_oracle.so")

KIND_SQUARE, KIND_RECT, KIND_PIN, KIND_SPATIAL = 0, 1, 2, 3
REWARD = {"beam": 0, "centroid": 1, "both": 2}


class OrcConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "kind", "height", "width", "min_component_w", "max_component_w", "min_component_h",
        "max_component_h", "max_num_components", "min_num_components", "net_distribution", "pin_spread",
        "min_num_nets", "max_num_nets", "max_num_pins_per_net", "min_num_pins_per_net", "reward_type",
        "reward_beam_width", "component_n")] + [("weight_wirelength", C.c_double),
                                                ("weight_num_intersections", C.c_double)]


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("pcbenv_oracle.c", "oracle_batch.c", "Makefile")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libpcbenv_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(OrcConfig)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_reset.argtypes = [C.c_void_p, C.c_int, ip, ip, C.c_int, C.c_int, ip, ip, ip, ip, ip]
        L.orc_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, dp, ip, dp, ip]
        L.orc_obs.restype = dp
        L.orc_obs.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        L.orc_current_component.argtypes = [C.c_void_p]
        L.orc_max_wirelength.restype = C.c_double
        L.orc_max_wirelength.argtypes = [C.c_void_p]
        L.orc_max_num_intersections.restype = C.c_double
        L.orc_max_num_intersections.argtypes = [C.c_void_p]
        L.orc_norm2.restype = C.c_double
        L.orc_norm2.argtypes = [C.c_double, C.c_double]
        L.orc_euclidean_distance.restype = C.c_double
        L.orc_euclidean_distance.argtypes = [C.c_double] * 4
        L.orc_is_intersect.argtypes = [dp, dp]
        L.orc_find_num_intersection.argtypes = [dp, ip, C.c_int]
        L.orc_find_wirelength.restype = C.c_double
        L.orc_find_wirelength.argtypes = [dp, ip, C.c_int]
        L.orc_get_centroid.argtypes = [ip, C.c_int, dp, dp]
        L.orc_pin_outlier.argtypes = [ip, C.c_int]
        L.orc_tuple_hash2.restype = C.c_uint64
        L.orc_tuple_hash2.argtypes = [C.c_int64, C.c_int64]
        L.orc_set_difference_order.argtypes = [ip, C.c_int, C.c_uint32, ip]
        L.orc_beam_search.argtypes = [C.c_int, C.c_int, ip, C.c_int, C.c_int, ip]
        L.orc_route.argtypes = [ip, ip, C.c_int, C.c_int, C.c_int, dp, ip]
        L.orc_find_reward_pts.argtypes = [C.POINTER(OrcConfig), C.c_int, ip, ip, C.c_int, dp]
        L.orc_upper_bound_wirelength.restype = C.c_double
        L.orc_upper_bound_wirelength.argtypes = [C.POINTER(OrcConfig)]
        L.orc_upper_bound_intersections.restype = C.c_double
        L.orc_upper_bound_intersections.argtypes = [C.POINTER(OrcConfig)]
        L.orc_intersections_norm.restype = C.c_double
        L.orc_intersections_norm.argtypes = [C.POINTER(OrcConfig)]
        L.orc_batch_create.restype = C.c_void_p
        L.orc_batch_create.argtypes = [C.POINTER(OrcConfig), C.c_int]
        L.orc_batch_destroy.argtypes = [C.c_void_p]
        L.orc_batch_env.restype = C.c_void_p
        L.orc_batch_env.argtypes = [C.c_void_p, C.c_int]
        L.orc_batch_reset_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int]
        L.orc_batch_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_batch_first_mismatch.restype = C.c_int64
        L.orc_batch_first_mismatch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int]
        _lib = L
    return _lib


def make_config(cfg) -> OrcConfig:
    """cfg: any object with the EnvConfig field names (duck-typed; no product import needed)."""
    c = OrcConfig()
    for name, _ in OrcConfig._fields_:
        if name == "reward_type":
            c.reward_type = REWARD.get(getattr(cfg, "reward_type", "both"), 2)
        else:
            setattr(c, name, getattr(cfg, name))
    return c


def _ia(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.intc))


def _ip(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


_OBS = {"grid": 0, "action_mask": 1, "all_components_feature": 2, "placement_mask": 3, "component_mask": 4,
        "all_pins_num_feature": 5, "all_pins_cat_feature": 6, "pin_grid": 7, "component_grid": 8}


def obs_keys(kind: int) -> Tuple[str, ...]:
    return {KIND_SQUARE: ("grid", "action_mask"),
            KIND_RECT: ("grid", "action_mask", "all_components_feature", "component_mask", "placement_mask"),
            KIND_PIN: ("grid", "action_mask", "all_components_feature", "placement_mask",
                       "all_pins_num_feature", "all_pins_cat_feature"),
            KIND_SPATIAL: ("grid", "pin_grid", "component_grid", "action_mask", "all_components_feature",
                           "placement_mask", "all_pins_num_feature", "all_pins_cat_feature")}[kind]


def obs_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    H, W, k = cfg.height, cfg.width, cfg.kind
    if k == KIND_SQUARE:
        return {"grid": (H, W), "action_mask": (H, W)}
    Cc = cfg.max_num_components
    if k == KIND_RECT:
        return {"grid": (H, W), "action_mask": (2, H, W), "all_components_feature": (Cc, 5),
                "component_mask": (Cc,), "placement_mask": (Cc,)}
    mp, N = cfg.max_component_h * cfg.max_component_w, cfg.max_num_nets
    if k == KIND_PIN:
        return {"grid": (H, W), "action_mask": (4, H, W), "all_components_feature": (Cc, 5),
                "placement_mask": (Cc,), "all_pins_num_feature": (Cc, mp, 4), "all_pins_cat_feature": (Cc, mp, 1)}
    return {"grid": (H, W), "pin_grid": (H, W, N + 1),
            "component_grid": (Cc, cfg.max_component_h, cfg.max_component_w, N + 1),
            "action_mask": (4, H, W), "all_components_feature": (Cc, 5 + mp), "placement_mask": (Cc,),
            "all_pins_num_feature": (Cc * mp + 1, 4), "all_pins_cat_feature": (Cc * mp + 1, 2)}


class OracleEnv:
    """One environment, reference semantics (reset takes the instance tables)."""

    def __init__(self, cfg, handle: Optional[int] = None, owner=None):
        self.cfg = cfg
        self._L = lib()
        self._c = make_config(cfg)
        self._own = handle is None
        self._owner = owner
        self._h = self._L.orc_create(C.byref(self._c)) if handle is None else handle
        self.shapes = obs_shapes(cfg)
        self.keys = obs_keys(cfg.kind)

    def __del__(self):
        if getattr(self, "_own", False) and getattr(self, "_h", None):
            self._L.orc_destroy(self._h)
            self._h = None

    def obs(self) -> Dict[str, np.ndarray]:
        out = {}
        for k in self.keys:
            n = C.c_int64()
            p = self._L.orc_obs(self._h, _OBS[k], C.byref(n))
            a = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
            out[k] = a.reshape(self.shapes[k])
        return out

    def reset(self, instance=None) -> Dict[str, np.ndarray]:
        if self.cfg.kind == KIND_SQUARE:
            z = _ia([])
            rc = self._L.orc_reset(self._h, 0, _ip(z), _ip(z), 0, 0, _ip(z), _ip(z), _ip(z), _ip(z), _ip(z))
        else:
            a = [_ia(getattr(instance, f)) for f in ("comp_h", "comp_w", "pin_rel_x", "pin_rel_y", "pin_net",
                                                      "pin_comp", "pin_id")]
            rc = self._L.orc_reset(self._h, len(a[0]), _ip(a[0]), _ip(a[1]), int(instance.num_nets), len(a[2]),
                                   _ip(a[2]), _ip(a[3]), _ip(a[4]), _ip(a[5]), _ip(a[6]))
        if rc != 0:
            raise RuntimeError(f"orc_reset failed ({rc})")
        return self.obs()

    def step_raw(self, action: Sequence[int]):
        if self.cfg.kind == KIND_SQUARE:
            o, x, y = 0, int(action[-2]), int(action[-1])
        else:
            o, x, y = (int(v) for v in action)
        r, d, has = C.c_double(), C.c_int(), C.c_int()
        info = (C.c_double * 2)()
        rc = self._L.orc_step(self._h, o, x, y, C.byref(r), C.byref(d), info, C.byref(has))
        if rc != 0:
            raise RuntimeError(f"orc_step failed ({rc})")
        inf = {"wirelength": info[0], "num_intersections": info[1]} if has.value else {}
        return r.value, bool(d.value), inf

    def step(self, action: Sequence[int]):
        r, d, inf = self.step_raw(action)
        return self.obs(), r, d, inf

    @property
    def current_component(self) -> int:
        return self._L.orc_current_component(self._h)


class OracleBatch:
    """n independent OracleEnvs stepped with OpenMP (cpu_baseline + batch parity)."""

    def __init__(self, cfg, n: int):
        self.cfg, self.n = cfg, n
        self._L = lib()
        self._c = make_config(cfg)
        self._h = self._L.orc_batch_create(C.byref(self._c), n)
        self.max_threads = self._L.orc_max_threads()

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_batch_destroy(self._h)
            self._h = None

    def env(self, i: int) -> OracleEnv:
        return OracleEnv(self.cfg, handle=self._L.orc_batch_env(self._h, i), owner=self)

    def reset_packed(self, packed: np.ndarray, mask: Optional[np.ndarray] = None, threads: int = 1):
        packed = np.ascontiguousarray(packed, np.uint8)
        assert packed.shape[0] == self.n
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        rc = self._L.orc_batch_reset_packed(self._h, packed.ctypes.data, packed.shape[1],
                                            self.cfg.max_num_components,
                                            None if m is None else m.ctypes.data, threads)
        if rc != 0:
            raise RuntimeError("orc_batch_reset_packed failed")

    def first_mismatch(self, key: str, dev_copy: np.ndarray, threads: int = 1) -> int:
        """First environment whose observation `key` differs from `dev_copy` (host copy of the device tensor, uint8
        cells or float64 features, leading dim = environments); -1 if the whole batch agrees."""
        a = np.ascontiguousarray(dev_copy)
        assert a.shape[0] == self.n and a.dtype in (np.uint8, np.float64), (a.shape, a.dtype)
        rc = self._L.orc_batch_first_mismatch(self._h, _OBS[key], a.ctypes.data, a.dtype.itemsize,
                                              int(a.size // self.n), threads)
        if rc == -2:
            raise ValueError(f"{key}: element count per environment differs from the oracle's")
        return int(rc)

    def step(self, actions: np.ndarray, threads: int = 1):
        a = np.ascontiguousarray(actions, np.int32).reshape(self.n, 3)
        reward = np.zeros(self.n, np.float64)
        done = np.zeros(self.n, np.uint8)
        info = np.zeros((self.n, 2), np.float64)
        rc = self._L.orc_batch_step(self._h, a.ctypes.data, reward.ctypes.data, done.ctypes.data,
                                    info.ctypes.data, threads)
        if rc != 0:
            raise RuntimeError("orc_batch_step failed")
        return reward, done, info


# ---- stand-alone reward helpers for the reference's hand-built KATs -----------------------------
def route(nets: Sequence[Sequence[Tuple[int, int]]], method: str, beam_width: int = 2):
    L = lib()
    pts = _ia([c for net in nets for p in net for c in p])
    off = _ia(np.concatenate([[0], np.cumsum([len(n) for n in nets])]))
    seg = np.zeros((len(pts) // 2 + 1, 4), np.float64)
    soff = _ia(np.zeros(len(nets) + 1))
    n = L.orc_route(_ip(pts), _ip(off), len(nets), REWARD[method], beam_width, _dp(seg), _ip(soff))
    if n < 0:
        raise RuntimeError("orc_route failed")
    return [[((seg[i, 0], seg[i, 1]), (seg[i, 2], seg[i, 3])) for i in range(soff[k], soff[k + 1])]
            for k in range(len(nets))]


def _flatten_route(route_):
    seg = np.ascontiguousarray([[a[0], a[1], b[0], b[1]] for net in route_ for (a, b) in net], np.float64).reshape(-1, 4)
    off = _ia(np.concatenate([[0], np.cumsum([len(n) for n in route_])]))
    return seg, off


def find_num_intersection(route_) -> int:
    seg, off = _flatten_route(route_)
    return lib().orc_find_num_intersection(_dp(seg), _ip(off), len(route_))


def find_wirelength(route_) -> float:
    seg, off = _flatten_route(route_)
    return lib().orc_find_wirelength(_dp(seg), _ip(off), len(route_))


def is_intersect(l1, l2) -> bool:
    a = np.asarray([l1[0][0], l1[0][1], l1[1][0], l1[1][1]], np.float64)
    b = np.asarray([l2[0][0], l2[0][1], l2[1][0], l2[1][1]], np.float64)
    return bool(lib().orc_is_intersect(_dp(a), _dp(b)))


def beam_search(start, points, beam_width: int):
    pts = _ia([c for p in points for c in p])
    out = _ia(np.zeros(len(points) + 1))
    n = lib().orc_beam_search(int(start[0]), int(start[1]), _ip(pts), len(points), beam_width, _ip(out))
    if n < 0:
        raise RuntimeError("orc_beam_search failed")
    return [tuple(start) if i < 0 else tuple(points[i]) for i in out[:n]]


def set_difference_order(points, visited_mask: int):
    pts = _ia([c for p in points for c in p])
    out = _ia(np.zeros(len(points) + 1))
    n = lib().orc_set_difference_order(_ip(pts), len(points), visited_mask, _ip(out))
    return [tuple(points[i]) for i in out[:n]]


def find_reward(cfg, nets, placed_all: bool = True):
    L = lib()
    c = make_config(cfg)
    pts = _ia([v for net in nets for p in net for v in p])
    off = _ia(np.concatenate([[0], np.cumsum([len(n) for n in nets])]))
    out = np.zeros(3, np.float64)
    if L.orc_find_reward_pts(C.byref(c), int(placed_all), _ip(pts), _ip(off), len(nets), _dp(out)) != 0:
        raise RuntimeError("orc_find_reward_pts failed")
    return float(out[0]), float(out[1]), float(out[2])
