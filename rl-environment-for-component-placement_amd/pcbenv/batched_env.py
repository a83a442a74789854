"""BatchedPlacementEnv: thousands of independent placement environments on one MI355X.

The host-side mirror of the reference `gym.Env` classes
(`environment/dummy_env_{square,rectangular,rectangular_pin,rectangular_pin_spatial}.py`):
same observation keys, same action encoding, `reset()` / `step()` with a leading
batch dimension.  Observations are torch tensors that live on the device and are
updated in place by every call (the reference returns fresh copies -- `.clone()`
where copy semantics are needed).  All compute happens in libpcbenv.so's HIP
kernels on the current torch stream; nothing here touches observation data.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .config import EnvConfig, KIND_PIN, KIND_RECT, KIND_SPATIAL, KIND_SQUARE
from .instances import Instance, InstanceStream, env_seed, pack_instances


FEATURE_KEYS = ("all_components_feature", "placement_mask", "component_mask", "all_pins_num_feature", "all_pins_cat_feature")
COMPACT_DTYPES = {"all_components_feature": torch.int16, "placement_mask": torch.uint8, "component_mask": torch.uint8,
                  "all_pins_num_feature": torch.int8, "all_pins_cat_feature": torch.int8}


def expand_compact_features(cfg: EnvConfig, compact: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Compact feature tensors (pcbenv_compact_features: int16 / int8 / uint8) -> the reference's float64 tensors, bit
    for bit: every element is a small integer except all_components_feature[..., 4], carried as h * w and divided by
    H * W here in float64 -- the one IEEE division the reference does (`area / grid_area`, S:203-239)."""
    out = {}
    for k, t in compact.items():
        f = t.to(torch.float64)
        if k == "all_components_feature":
            f[..., 4] = f[..., 4] / float(cfg.height * cfg.width)
        out[k] = f
    return out


def obs_spec(cfg: EnvConfig) -> Dict[str, tuple]:
    """key -> (shape without batch dim, torch dtype); keys and shapes are the reference's."""
    H, W, k = cfg.height, cfg.width, cfg.kind
    u8, f64 = torch.uint8, torch.float64
    if k == KIND_SQUARE:
        return {"grid": ((H, W), u8), "action_mask": ((H, W), u8)}
    Cc = cfg.max_num_components
    if k == KIND_RECT:
        return {"grid": ((H, W), u8), "action_mask": ((2, H, W), u8), "all_components_feature": ((Cc, 5), f64),
                "component_mask": ((Cc,), f64), "placement_mask": ((Cc,), f64)}
    mp, N = cfg.max_num_pins_per_component, cfg.max_num_nets
    if k == KIND_PIN:
        return {"grid": ((H, W), u8), "action_mask": ((4, H, W), u8), "all_components_feature": ((Cc, 5), f64),
                "placement_mask": ((Cc,), f64), "all_pins_num_feature": ((Cc, mp, 4), f64),
                "all_pins_cat_feature": ((Cc, mp, 1), f64)}
    return {"grid": ((H, W), u8), "pin_grid": ((H, W, N + 1), u8),
            "component_grid": ((Cc, cfg.max_component_h, cfg.max_component_w, N + 1), u8),
            "action_mask": ((4, H, W), u8), "all_components_feature": ((Cc, 5 + mp), f64),
            "placement_mask": ((Cc,), f64), "all_pins_num_feature": ((Cc * mp + 1, 4), f64),
            "all_pins_cat_feature": ((Cc * mp + 1, 2), f64)}


class _ExternalBlock:
    """__cuda_array_interface__ view over library-owned device memory (no ownership)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _as_tensor(ptr: int, nbytes: int, device) -> torch.Tensor:
    return torch.as_tensor(_ExternalBlock(ptr, nbytes), device=device)


class BatchedPlacementEnv:
    is_batched = True

    def __init__(self, cfg: EnvConfig, num_envs: int, device="cuda:0", queue_depth: int = 1,
                 run_seed: int = 0, first_env_index: int = 0, incremental_obs: bool = False,
                 auto_reset: bool = False, threads_per_env: int = 0,
                 mask_marginals: bool = False, num_slots: int = 1, options: Optional[Dict[str, int]] = None,
                 compact_features: bool = False, allocator=None):
        """num_slots > 1: trajectory layout -- every output tensor is allocated `[num_slots, B, ...]` (`self.traj`,
        `self.traj_reward`, ...), `select_slot(s)` chooses the slot the next reset / step writes and `self.obs`,
        `self.reward`, `self.done`, `self.info_raw` are views of that slot (pcbenv_bind_buffers_slots).
        options: tuning knobs of the handle, `pcbenv_set_option` (include/pcbenv.h): "stream_threshold_bytes",
        "terminal_teams", "gen_grid", "gen_lanes" -- none changes a result.
        compact_features (trajectory layout only): the feature tensors are kept as int16 / int8 / uint8
        (`pcbenv_bind_compact_features`: 8x fewer feature bytes per step) under the same keys of `traj` / `obs`;
        `obs_f64()` / `expand_compact_features` give the reference's float64 tensors, bit for bit.
        allocator: `f(name, shape, dtype) -> zero-filled device tensor` for the observation tensors (default torch.zeros);
        lets a caller place them (tools/c5_modes.py studies where the big cell tensors should sit)."""
        cfg.validate()
        self.cfg, self.num_envs, self.queue_depth = cfg, int(num_envs), int(queue_depth)
        self.run_seed, self.first_env_index = int(run_seed), int(first_env_index)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BatchedPlacementEnv needs a GPU device (there is no CPU fallback)")
        self._L = _lib.load()
        # per-environment spaces (the reference constructors' declarations); every tensor adds a leading batch dim
        from .spaces import action_space_for, observation_space_for
        self.action_space = self.single_action_space = action_space_for(cfg)
        self.observation_space = self.single_observation_space = observation_space_for(cfg)
        self.auto_reset = bool(auto_reset)
        self._ccfg = _lib.make_config(cfg, num_envs, queue_depth,
                                      (_lib.FLAG_INCREMENTAL_OBS if incremental_obs else 0)
                                      | (_lib.FLAG_AUTO_RESET if auto_reset else 0))
        self._ccfg.threads_per_env = int(threads_per_env)  # 0 = auto; 64 / 256 threads (1 / 4 waves) per environment
        h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self._L.pcbenv_create(C.byref(self._ccfg), dev_index, C.byref(h)))
        self._h = h
        for name, value in (options or {}).items():
            self.set_option(name, value)
        B = self.num_envs
        S = self.num_slots = int(num_slots)
        self.compact_features = bool(compact_features)
        if self.compact_features and S < 2:
            raise ValueError("compact_features needs the trajectory layout (num_slots > 1)")
        alloc = allocator or (lambda name, shape, dtype: torch.zeros(shape, dtype=dtype, device=self.device))
        with torch.cuda.device(self.device):
            self.traj: Dict[str, torch.Tensor] = {
                k: alloc(k, (S, B) + shape, COMPACT_DTYPES[k] if self.compact_features and k in FEATURE_KEYS else dt)
                for k, (shape, dt) in obs_spec(cfg).items()}
            self.traj_reward = torch.zeros((S, B), dtype=torch.float64, device=self.device)
            self.traj_done = torch.zeros((S, B), dtype=torch.uint8, device=self.device)
            self.traj_info = torch.full((S, B, 2), float("nan"), dtype=torch.float64, device=self.device)
            self._actions = torch.zeros((B, 3), dtype=torch.int32, device=self.device)
            O = cfg.num_orientations
            # marginals of action_mask for factorised policies (not reference observation keys)
            self.traj_marginals = {"orientation": torch.zeros((S, B, O), dtype=torch.uint8, device=self.device),
                                   "rows": torch.zeros((S, B, O, cfg.height), dtype=torch.uint8, device=self.device)} if mask_marginals else {}
        bufs = _lib.PcbenvBuffers()
        for name in _lib.BUFFER_FIELDS:
            t = self.traj.get(name)
            if name == "reward":
                t = self.traj_reward
            elif name == "done":
                t = self.traj_done
            elif name == "info":
                t = self.traj_info if cfg.kind in (KIND_PIN, KIND_SPATIAL) else None
            elif name == "mask_orientation":
                t = self.traj_marginals.get("orientation")
            elif name == "mask_rows":
                t = self.traj_marginals.get("rows")
            if self.compact_features and name in FEATURE_KEYS:
                t = None  # not produced in float64: the compact twin is bound below
            setattr(bufs, name, t.data_ptr() if t is not None else None)
        _lib.check(self._L.pcbenv_bind_buffers_slots(self._h, C.byref(bufs), S), self._h)
        if self.compact_features:
            cb = _lib.PcbenvCompactFeatures()
            for name in _lib.COMPACT_FIELDS:
                t = self.traj.get(name)
                setattr(cb, name, t.data_ptr() if t is not None else None)
            _lib.check(self._L.pcbenv_bind_compact_features(self._h, C.byref(cb)), self._h)
        self.slot = -1
        self.select_slot(0)
        self._last_done = self.done  # `done` of the latest step (it may live in another slot than the selected one)
        self._streams: Optional[List[InstanceStream]] = None
        self._native = None
        self.device_instances = False  # True once the on-device generator owns the queue (enable_device_instances)
        torch.cuda.synchronize(self.device)

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            self._L.pcbenv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int):
        code = {"stream_threshold_bytes": _lib.OPT_STREAM_THRESHOLD_BYTES, "terminal_teams": _lib.OPT_TERMINAL_TEAMS,
                "gen_grid": _lib.OPT_GEN_GRID, "gen_lanes": _lib.OPT_GEN_LANES}[name]
        _lib.check(self._L.pcbenv_set_option(self._h, code, int(value)), self._h)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def select_slot(self, slot: int):
        """The slot of the `[num_slots, B, ...]` tensors the next reset / step writes; `obs`, `reward`, `done`,
        `info_raw`, `mask_marginals` become views of it."""
        slot = int(slot) % self.num_slots
        if slot == self.slot:
            return
        _lib.check(self._L.pcbenv_select_slot(self._h, slot), self._h)
        self.slot = slot
        self.obs: Dict[str, torch.Tensor] = {k: v[slot] for k, v in self.traj.items()}
        self.reward, self.done, self.info_raw = self.traj_reward[slot], self.traj_done[slot], self.traj_info[slot]
        self.mask_marginals = {k: v[slot] for k, v in self.traj_marginals.items()}

    def obs_f64(self, slot: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """The observation of `slot` (default: the selected one) with the feature tensors in the reference's float64
        (cell tensors stay uint8 views) -- what `obs` is without compact_features."""
        obs = self.obs if slot is None else {k: v[int(slot) % self.num_slots] for k, v in self.traj.items()}
        if not self.compact_features:
            return dict(obs)
        out = {k: v for k, v in obs.items() if k not in FEATURE_KEYS}
        out.update(expand_compact_features(self.cfg, {k: v for k, v in obs.items() if k in FEATURE_KEYS}))
        return out

    # -- instances --------------------------------------------------------------------------
    def load_instances(self, instances: Sequence[Instance], slot: int = 0, env_ids: Optional[Sequence[int]] = None):
        """Queue one instance per environment (or per listed env id) into `slot`."""
        if self.cfg.kind == KIND_SQUARE:
            return
        packed = np.ascontiguousarray(pack_instances(self.cfg, instances))
        self.load_packed(packed, slot, env_ids)

    def _host_queue_only(self, what: str):
        if self.device_instances:
            raise RuntimeError(f"{what}: the on-device generator owns the instance queue (enable_device_instances); "
                               "host-side records can no longer be queued")

    def load_packed(self, packed: np.ndarray, slot: int = 0, env_ids: Optional[Sequence[int]] = None):
        self._host_queue_only("load_instances")
        ids = None if env_ids is None else np.ascontiguousarray(env_ids, np.int32)
        _lib.check(self._L.pcbenv_load_instances(
            self._h, None if ids is None else ids.ctypes.data, packed.shape[0], slot, packed.ctypes.data,
            self._stream()), self._h)

    def generate_instances(self, native: bool = True, verify: int = 0, threads: int = 8):
        """Fill every queue slot from per-environment reference RNG streams (seed = f(run_seed, global env index)):
        slot s holds each environment's next reset instance, exactly what the reference env seeded with that
        stream seed would draw at that `reset()`.  `native=True` uses libpcbenv.so's generator
        (csrc/instance_gen.cpp, ~100x faster); `native=False` the NumPy/`random`-calling `InstanceStream`;
        `verify=n` cross-checks the first n environments of every slot between the two.
        Returns the packed records per slot (uint8 [B, instance_stride])."""
        if self.cfg.kind == KIND_SQUARE:
            return []
        self._host_queue_only("generate_instances")
        mode = ("native", bool(verify)) if native else ("numpy", False)
        if getattr(self, "_gen_mode", mode) != mode:
            raise RuntimeError("generate_instances: keep the same generator (native / verify) for the lifetime of the "
                               "environment -- every call continues the per-environment RNG streams")
        self._gen_mode = mode
        seeds = [env_seed(self.run_seed, self.first_env_index + i) for i in range(self.num_envs)]
        if native and self._native is None:
            from .instances import NativeInstanceStreams
            self._native = NativeInstanceStreams(self.cfg, seeds, threads)
        if (not native or verify) and self._streams is None:
            n = self.num_envs if not native else min(verify, self.num_envs)
            self._streams = [InstanceStream(self.cfg, seeds[i]) for i in range(n)]
        out = []
        for s in range(self.queue_depth):
            if native:
                packed = self._native.next_packed()
                if verify:
                    ref = pack_instances(self.cfg, [st.next() for st in self._streams])
                    if not np.array_equal(packed[:len(ref)], ref):
                        raise RuntimeError("native instance generator disagrees with InstanceStream")
            else:
                packed = np.ascontiguousarray(pack_instances(self.cfg, [st.next() for st in self._streams]))
            self.load_packed(packed, slot=s)
            out.append(packed)
        return out

    def enable_device_instances(self):
        """Fresh instances at every reset, generated on the device (csrc/pcb_geninst.h): environment i draws stream
        `env_seed(run_seed, first_env_index + i)`, the same records `generate_instances()` would queue, episode after
        episode, without the host in the loop.  The library owns the queue from here on."""
        if self.cfg.kind == KIND_SQUARE:
            return
        seeds = np.asarray([env_seed(self.run_seed, self.first_env_index + i) for i in range(self.num_envs)], np.int64)
        if seeds.min() < 0 or seeds.max() >= 2 ** 32:
            raise ValueError("stream seeds must fit 32 bits (np.random.seed)")
        s32 = np.ascontiguousarray(seeds, np.uint32)
        _lib.check(self._L.pcbenv_instgen_device_enable(self._h, s32.ctypes.data, self._stream()), self._h)
        self.device_instances = True

    def device_instance_errors(self) -> int:
        err = C.c_uint32()
        _lib.check(self._L.pcbenv_instgen_device_status(self._h, C.byref(err), self._stream()), self._h)
        return int(err.value)

    def queued_instances(self, slot: int) -> np.ndarray:
        """Packed records of one queue slot, uint8 [B, instance_stride] (synchronises)."""
        from .instances import instance_stride
        out = np.zeros((self.num_envs, instance_stride(self.cfg)), np.uint8)
        _lib.check(self._L.pcbenv_get_instances(self._h, int(slot), out.ctypes.data, self._stream()), self._h)
        return out

    def refill_slot(self, slot: int, native: bool = True):
        """Overwrite one queue slot with every environment's next instance (call when no environment can be
        about to read that slot, e.g. between rollouts; copies are ordered on the current stream)."""
        if self.cfg.kind == KIND_SQUARE:
            return None
        self._host_queue_only("refill_slot")
        if native:
            if self._native is None:
                raise RuntimeError("call generate_instances(native=True) first")
            packed = self._native.next_packed()
        else:
            packed = np.ascontiguousarray(pack_instances(self.cfg, [st.next() for st in self._streams]))
        self.load_packed(packed, slot=slot)
        return packed

    # -- gym-style API ----------------------------------------------------------------------
    def reset(self, mask: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        _lib.check(self._L.pcbenv_reset(self._h, None if m is None else m.data_ptr(), self._stream()), self._h)
        return self.obs

    def reset_done(self) -> Dict[str, torch.Tensor]:
        """Explicit auto-reset: environments whose last step returned done take their next instance."""
        _lib.check(self._L.pcbenv_reset(self._h, self._last_done.data_ptr(), self._stream()), self._h)
        return self.obs

    def step(self, actions: torch.Tensor):
        """actions: int tensor [B, 3] = (orientation, x, y) ([B, 2] = (x, y) for the square env) or flat [B]
        (`a = o*H*W + x*W + y`, utils/environment/env_wrappers.py:80-98)."""
        a = actions.to(device=self.device, dtype=torch.int32)
        if a.dim() == 1:
            fmt = _lib.ACTION_FLAT
            a = a.contiguous()
        else:
            fmt = _lib.ACTION_TUPLE
            if a.shape[1] == 2:
                self._actions[:, 1:] = a
                a = self._actions
            a = a.contiguous()
        assert a.shape[0] == self.num_envs
        _lib.check(self._L.pcbenv_step(self._h, a.data_ptr(), fmt, self._stream()), self._h)
        self._last_done = self.done
        return self.obs, self.reward, self.done, self.info

    @property
    def info(self) -> Dict[str, torch.Tensor]:
        """`wirelength` / `num_intersections` per environment; NaN where the reference's info dict is `{}`."""
        if self.cfg.kind not in (KIND_PIN, KIND_SPATIAL):
            return {}
        return {"wirelength": self.info_raw[:, 0], "num_intersections": self.info_raw[:, 1]}

    def sample_actions(self, step_index: int, flat: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Uniform draw over the legal actions of every environment, on device (random-policy counterpart)."""
        if out is None:
            out = torch.empty((self.num_envs,) if flat else (self.num_envs, 3), dtype=torch.int32, device=self.device)
        _lib.check(self._L.pcbenv_sample_actions(
            self._h, out.data_ptr(), _lib.ACTION_FLAT if flat else _lib.ACTION_TUPLE, self.run_seed,
            self.first_env_index, int(step_index), self._stream()), self._h)
        return out

    def rollout_step(self, step_index: int, flat: bool = False, out: Optional[torch.Tensor] = None):
        """`sample_actions` + `step` in one kernel launch (the body of the reference's random-policy
        `simulate()` loop, agent/random/random_policy_square.py:38-56); `out` receives the actions taken."""
        if out is None:
            out = torch.empty((self.num_envs,) if flat else (self.num_envs, 3), dtype=torch.int32, device=self.device)
        _lib.check(self._L.pcbenv_step_sampled(
            self._h, out.data_ptr(), _lib.ACTION_FLAT if flat else _lib.ACTION_TUPLE, self.run_seed,
            self.first_env_index, int(step_index), self._stream()), self._h)
        self._last_done = self.done
        return self.obs, self.reward, self.done, self.info, out

    def rollout_steps(self, step_index0: int, num_steps: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`num_steps` sample+step transitions in one persistent kernel launch (use with auto_reset=True): step t
        writes slot `(self.slot + t) % num_slots`; returns the actions taken, int32 [num_steps, B, 3].  The selected
        slot is left where it was -- `select_slot(self.slot + num_steps)` continues behind the rollout."""
        if out is None:
            out = torch.empty((num_steps, self.num_envs, 3), dtype=torch.int32, device=self.device)
        _lib.check(self._L.pcbenv_rollout_sampled(
            self._h, out.data_ptr(), _lib.ACTION_TUPLE, int(num_steps), self.run_seed, self.first_env_index,
            int(step_index0), self._stream()), self._h)
        return out

    def queue_cursors(self):
        """(min, max) over the environments of the number of resets performed so far (synchronises)."""
        lo, hi = C.c_uint32(), C.c_uint32()
        _lib.check(self._L.pcbenv_queue_cursors(self._h, C.byref(lo), C.byref(hi), self._stream()), self._h)
        return lo.value, hi.value

    # -- checkpoint / resume ------------------------------------------------------------------
    def state_dict(self) -> dict:
        """Library state + observation tensors (host copies).  The reference never serialises env state
        (SURVEY.md §5); this is what a resumable rollout needs.  With the on-device generator the blob also holds its
        streams, counters and queued records (a resumed run draws the same instances); a host-fed queue is reloaded
        separately."""
        n = self._L.pcbenv_state_bytes(self._h)
        buf = np.empty(n, np.uint8)
        _lib.check(self._L.pcbenv_get_state(self._h, buf.ctypes.data, self._stream()), self._h)
        stride = C.c_int64()
        self._L.pcbenv_mask_bits(self._h, C.byref(stride))
        nb = stride.value * self.num_envs  # the state blocks; the generator's section (if any) follows them
        d = {"state": buf[:nb], "generator": buf[nb:], "reward": self.reward.cpu(), "done": self.done.cpu(), "info": self.info_raw.cpu(),
             "last_done": self._last_done.cpu(), "slot": self.slot, "device_instances": self.device_instances}
        d.update({"obs/" + k: v.cpu() for k, v in self.obs.items()})
        return d

    def load_state_dict(self, d: dict):
        buf = np.ascontiguousarray(np.concatenate([np.asarray(d["state"], np.uint8).ravel(), np.asarray(d.get("generator", ()), np.uint8).ravel()]))
        if bool(d.get("device_instances", False)) != self.device_instances:
            raise ValueError("the checkpoint was taken with" + ("" if d.get("device_instances") else "out") + " the on-device generator: "
                             "call enable_device_instances() on this environment " + ("first" if d.get("device_instances") else "only after restoring"))
        if buf.size != self._L.pcbenv_state_bytes(self._h):
            raise ValueError("state size does not match this environment")
        if "slot" in d:
            self.select_slot(int(d["slot"]))
        _lib.check(self._L.pcbenv_set_state(self._h, buf.ctypes.data, self._stream()), self._h)
        self.reward.copy_(d["reward"]); self.done.copy_(d["done"]); self.info_raw.copy_(d["info"])
        for k, v in self.obs.items():
            v.copy_(d["obs/" + k])
        # `done` of the latest step: restored into the selected slot's tensor (where reset_done() will look)
        self._last_done = self.done
        if "last_done" in d:
            self._last_done.copy_(d["last_done"])

    def mask_bits(self) -> torch.Tensor:
        """Bit-packed legal-action mask, int64 [B, 2, H, ceil(W/64)] (a copy; bit y of word [b, o, x, y // 64])."""
        stride = C.c_int64()
        ptr = self._L.pcbenv_mask_bits(self._h, C.byref(stride))
        H, WW = self.cfg.height, (self.cfg.width + 63) // 64
        nbytes = stride.value * self.num_envs
        raw = _as_tensor(ptr, nbytes, self.device)  # library-owned block, wrapped without taking ownership
        return raw.view(self.num_envs, stride.value)[:, :2 * H * WW * 8].contiguous().view(torch.int64).view(self.num_envs, 2, H, WW)

    # reference-style attribute access
    @property
    def action_mask(self) -> torch.Tensor:
        return self.obs["action_mask"]

    @property
    def grid(self) -> torch.Tensor:
        return self.obs["grid"]
