"""The native instance generator (csrc/instance_gen.cpp: MT19937 + NumPy legacy randint / normal / multinomial +
CPython random.choice restated in C++) against (a) the tables the reference generated (golden files) and
(b) `InstanceStream`, which calls NumPy / `random` themselves.  CPU only."""
import numpy as np
import pytest

from golden_util import case_names, load_case
from pcbenv import EnvConfig, InstanceStream, named_config, pack_instances
from pcbenv.instances import NativeInstanceStreams, unpack_instances

CONFIGS = [named_config("c2"), named_config("c3"), named_config("c4"), named_config("c5"),
           EnvConfig.spatial(10, 10, 3, 4, 2, 4, 2, 4, 6, 1, 2, 4, 5, 2, "both", 2, 0.5),
           EnvConfig.pin(10, 10, 1, 1, 2, 4, 2, 4, 6, 1, 2, 4, 5, 2, "both", 2, 0.5),
           EnvConfig.pin(30, 30, 5, 2, 2, 5, 2, 5, 6, 1, 2, 4, 5, 2, "beam", 2, 0.5),
           EnvConfig.spatial(20, 20, 0, 0, 1, 3, 1, 5, 9, 3, 1, 5, 4, 2, "beam", 2, 0.5),
           EnvConfig.spatial(24, 24, 5, 5, 2, 4, 2, 4, 12, 6, 2, 3, 16, 9, "both", 4, 0.5),
           EnvConfig.rect(6, 6, 2, 4, 2, 4, 4, 2)]


@pytest.mark.parametrize("idx", range(len(CONFIGS)))
def test_native_streams_equal_numpy_streams(idx):
    cfg = CONFIGS[idx]
    seeds = list(range(1000, 1000 + (60 if cfg.height > 64 else 150)))
    nat = NativeInstanceStreams(cfg, seeds, threads=4)
    py = [InstanceStream(cfg, s) for s in seeds]
    for ep in range(3):  # successive resets continue both RNG streams (incl. the cached second gaussian)
        a = nat.next_packed()
        b = pack_instances(cfg, [st.next() for st in py])
        assert np.array_equal(a, b), (idx, ep, np.flatnonzero((a != b).any(axis=1))[:5])


@pytest.mark.parametrize("name", [n for n in case_names() if not n.startswith("square")])
def test_native_streams_reproduce_reference_tables(name):
    meta, cfg, eps = load_case(name)
    seeds = sorted({e.seed for e in eps})
    nat = NativeInstanceStreams(cfg, seeds, threads=1)
    per_seed = {s: [] for s in seeds}
    for _ in range(meta["episodes"]):
        for s, ins in zip(seeds, nat.next()):
            per_seed[s].append(ins)
    for e in eps:
        got, want = per_seed[e.seed][e.ep], e.instance
        for f in ("comp_h", "comp_w", "pin_rel_x", "pin_rel_y", "pin_net", "pin_comp", "pin_id"):
            assert np.array_equal(getattr(got, f), getattr(want, f)), (name, e.seed, e.ep, f)
        assert got.num_nets == want.num_nets


def test_unpack_is_inverse_of_pack():
    cfg = named_config("c4")
    ins = [InstanceStream(cfg, s).next() for s in range(5)]
    back = unpack_instances(cfg, pack_instances(cfg, ins))
    for a, b in zip(ins, back):
        for f in ("comp_h", "comp_w", "pin_rel_x", "pin_rel_y", "pin_net", "pin_comp", "pin_id"):
            assert np.array_equal(getattr(a, f), getattr(b, f))
