"""The C-ABI library loads and exports every symbol include/pcbenv.h declares; the
parameter validation (no device needed) behaves like the reference constructors.
No compute call is made: this runs in the CPU-only container."""
import ctypes as C
import os
import re

import pytest

from pcbenv import EnvConfig, named_config, _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(REPO, "include", "pcbenv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcbenv_[a-z_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    L = _lib.load()
    names = _declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(L, n), f"libpcbenv.so does not export {n}"
    assert set(names) == set(_lib.EXPORTS), "pcbenv/_lib.py binds a different set than the header declares"
    assert L.pcbenv_abi_version() == _lib.ABI_VERSION == 3


def test_sizes_match_host_packing():
    L = _lib.load()
    from pcbenv import instance_stride
    for name in ("c2", "c3", "c4", "c5"):
        cfg = named_config(name)
        c = _lib.make_config(cfg, 4)
        assert L.pcbenv_instance_stride(C.byref(c)) == instance_stride(cfg)
        assert L.pcbenv_max_total_pins(C.byref(c)) == cfg.max_total_pins
    assert named_config("c3").max_total_pins == 48 and named_config("c5").max_total_pins == 128


def _create(cfg_struct):
    L = _lib.load()
    h = C.c_void_p()
    rc = L.pcbenv_create(C.byref(cfg_struct), 0, C.byref(h))
    msg = L.pcbenv_last_error(None).decode()
    if rc == 0:
        L.pcbenv_destroy(h)
    return rc, msg


def test_create_rejects_what_the_reference_rejects():
    """ValueError cases of the reference constructors -> PCBENV_EINVAL before any device call."""
    good = named_config("c3")
    bad = []
    for field, value in (("max_component_w", 65), ("min_component_w", 0), ("max_num_components", 0),
                         ("min_num_pins_per_net", 1), ("reward_beam_width", 0), ("height", -1)):
        c = _lib.make_config(good, 8)
        setattr(c, field, value)
        bad.append((field, c))
    c = _lib.make_config(good, 8); c.reward_type = 7; bad.append(("reward_type", c))
    c = _lib.make_config(named_config("c4"), 8); c.reward_beam_width = 1; bad.append(("spatial beam < 2", c))
    c = _lib.make_config(named_config("c1"), 1); c.component_n = 9; bad.append(("square n > grid", c))
    for what, c in bad:
        rc, msg = _create(c)
        assert rc == _lib.PCBENV_EINVAL, (what, rc, msg)
        assert msg
    # beyond the HIP path's limits (valid for the reference): PCBENV_ELIMIT
    c = _lib.make_config(EnvConfig.rect(200, 200, 2, 6, 2, 6, 8, 8), 8)
    assert _create(c)[0] == _lib.PCBENV_ELIMIT
    c = _lib.make_config(good, 8); c.queue_depth = 0
    assert _create(c)[0] == _lib.PCBENV_ELIMIT


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rc, msg = _create(_lib.make_config(named_config("c3"), 8))
    assert rc == _lib.PCBENV_EHIP and msg  # no silent CPU fallback
    from pcbenv.batched_env import BatchedPlacementEnv
    with pytest.raises(RuntimeError):
        BatchedPlacementEnv(named_config("c3"), 8, device="cpu")


def test_null_handles_are_errors_not_crashes():
    L = _lib.load()
    assert L.pcbenv_reset(None, None, None) == _lib.PCBENV_EINVAL
    assert L.pcbenv_step(None, None, 0, None) == _lib.PCBENV_EINVAL
    assert L.pcbenv_bind_buffers(None, None) == _lib.PCBENV_EINVAL
    assert L.pcbenv_sample_actions(None, None, 0, 0, 0, 0, None) == _lib.PCBENV_EINVAL
    L.pcbenv_destroy(None)
