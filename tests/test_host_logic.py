"""Host-side logic (no GPU): constructor validation like the reference's, the instance generator against the
tables the reference generated (golden), packing, flat-action index math, the env factory."""
import json
import os

import numpy as np
import pytest

from golden_util import case_names, load_case
from pcbenv import (EnvConfig, InstanceStream, config_from_env_config, env_seed, flat_to_tuple, instance_stride,
                    named_config, pack_instances, tuple_to_flat)
from pcbenv.config import KIND_PIN, KIND_SPATIAL, KIND_SQUARE


@pytest.mark.parametrize("name", [n for n in case_names() if not n.startswith("square")])
def test_instance_stream_reproduces_reference_tables(name):
    """Every golden episode's instance tables == InstanceStream(cfg, seed) continued over resets (Appendix A)."""
    meta, cfg, eps = load_case(name)
    streams = {}
    for e in eps:
        st = streams.setdefault(e.seed, InstanceStream(cfg, e.seed))
        got, want = st.next(), e.instance
        for f in ("comp_h", "comp_w", "pin_rel_x", "pin_rel_y", "pin_net", "pin_comp", "pin_id"):
            assert np.array_equal(getattr(got, f), getattr(want, f)), (name, e.seed, e.ep, f)
        assert got.num_nets == want.num_nets


def test_constructor_validation_mirrors_reference():
    # dummy_env_rectangular.py:239-251
    with pytest.raises(ValueError): EnvConfig.rect(6, 6, 2, 7, 2, 4, 4, 1)
    with pytest.raises(ValueError): EnvConfig.rect(6, 6, 0, 4, 2, 4, 4, 1)
    with pytest.raises(ValueError): EnvConfig.rect(6, 6, 2, 4, 2, 4, 0, 1)
    with pytest.raises(ValueError): EnvConfig.rect(6, 6, 2, 4, 2, 4, 37, 1)
    # dummy_env_square.py:67-72
    with pytest.raises(ValueError): EnvConfig.square(4, 4, 5)
    with pytest.raises(ValueError): EnvConfig.square(-1, 4, 1)
    # dummy_env_rectangular_pin.py:598-641
    ok = (10, 10, 1, 1, 2, 4, 2, 4, 4, 2, 4, 4, 2)
    EnvConfig.pin(*ok)
    with pytest.raises(ValueError): EnvConfig.pin(*ok, 3)                         # min pins > max pins
    with pytest.raises(ValueError): EnvConfig.pin(*ok, 1)                         # min pins < 2
    with pytest.raises(ValueError): EnvConfig.pin(*ok, 2, "nearest")              # reward type
    with pytest.raises(ValueError): EnvConfig.pin(*ok, 2, "beam", 0)              # beam width
    with pytest.raises(ValueError): EnvConfig.pin(10, 10, 1, 1, 1, 4, 1, 4, 4, 1, 4, 4, 2)  # 2*4 > 1*1*1
    # quirk Q6: pin compares w with width, spatial/rect compare w with height
    EnvConfig.pin(5, 10, 1, 1, 2, 10, 2, 5, 4, 2, 4, 4, 2)
    with pytest.raises(ValueError): EnvConfig.spatial(5, 10, 1, 1, 2, 10, 2, 5, 4, 2, 4, 4, 2)
    # dummy_env_rectangular_pin_spatial.py:593-607
    with pytest.raises(ValueError): EnvConfig.spatial(*ok, 2, "both", 1)
    with pytest.raises(ValueError): EnvConfig.spatial(*ok, 2, "both", 3)          # beam > max pins per net
    with pytest.raises(ValueError): EnvConfig.spatial(*ok, 2, "both", 2, 1)       # weight must be a float
    with pytest.raises(ValueError): EnvConfig.spatial(*ok, 2, "both", 2, -0.5)
    # the complexity knobs are clipped to [0, 9]
    c = EnvConfig.pin(10, 10, 15, -3, 2, 4, 2, 4, 4, 2, 4, 4, 2)
    assert (c.net_distribution, c.pin_spread) == (9, 0)


def test_reward_constants_match_survey_probes():
    """values probed from the reference."""
    assert named_config("c4").max_wirelength == 16.970562748477143
    assert named_config("c4").max_num_intersections == 1008.0
    assert named_config("c3").max_wirelength == 2172.2320318050743
    c3, c4 = named_config("c3"), named_config("c4")
    worst = lambda c: -c.weight_wirelength * (c.max_wirelength / c.wirelength_norm) - c.weight_num_intersections * (c.max_num_intersections / c.intersections_norm)
    assert worst(c4) == -10.56629126073624 and worst(c3) == -18.98528137423857


def test_flat_action_index_math():
    """utils/environment/env_wrappers.py:80-98 (rect) and :184-199 (square)."""
    for (O, H, W) in ((4, 64, 64), (2, 6, 7), (4, 10, 10)):
        for a in (0, 1, W, H * W - 1, H * W, O * H * W - 1, 2 * H * W + 3 * W + 5):
            if a >= O * H * W:
                continue
            o, rem = divmod(a, H * W)
            x, y = divmod(rem, W)
            assert flat_to_tuple(a, O, H, W) == (o, x, y)
            assert tuple_to_flat((o, x, y), O, H, W) == a
    assert flat_to_tuple(23, 1, 5, 7) == divmod(23, 7)
    assert tuple_to_flat((3, 2), 1, 5, 7) == 23


def test_pack_instances_layout_and_limits():
    cfg = named_config("c3")
    ins = [InstanceStream(cfg, env_seed(0, i)).next() for i in range(3)]
    packed = pack_instances(cfg, ins)
    assert packed.shape == (3, instance_stride(cfg)) == (3, 16 + 8 * (16 + 48))
    hdr = packed[:, :16].view(np.int32)
    assert list(hdr[:, 0]) == [16] * 3 and list(hdr[:, 2]) == [48] * 3
    pins = packed[0, 16 + 8 * 16:].reshape(48, 8)
    assert np.array_equal(pins[:, 2], ins[0].pin_net) and np.array_equal(pins[:, 0], ins[0].pin_rel_x)
    assert np.array_equal(pins[:, 4].astype(int) | (pins[:, 5].astype(int) << 8), ins[0].pin_id)
    bad = ins[0]
    bad.pin_net = bad.pin_net[::-1].copy()
    with pytest.raises(ValueError):
        pack_instances(cfg, [bad])


def test_factory_reads_reference_json_shapes():
    """utils/agent/utils.py:317-418: type strings, positional order, Q5 (weight_num_intersections never forwarded)."""
    ec = {"type": "rectangle_spatial_pin", "height": 10, "width": 10, "net_distribution": 9, "pin_spread": 9,
          "min_component_w": 2, "max_component_w": 2, "min_component_h": 2, "max_component_h": 2,
          "max_num_components": 5, "min_num_components": 5, "min_num_nets": 3, "max_num_nets": 3,
          "max_num_pins_per_net": 6, "min_num_pins_per_net": 6, "reward_type": "centroid", "reward_beam_width": 2,
          "weight_wirelength": 0.75, "weight_num_intersections": 0.25}
    c = config_from_env_config(ec)
    assert c.kind == KIND_SPATIAL and c.weight_wirelength == 0.75 and c.weight_num_intersections == 0.5
    assert config_from_env_config(dict(ec, type="rectangle_pin_attn_all")).kind == KIND_PIN
    assert config_from_env_config({"type": "square", "height": 8, "width": 8, "component_n": 3}).kind == KIND_SQUARE
    with pytest.raises(KeyError):
        config_from_env_config(dict(ec, type="hexagon"))


def test_env_seed_is_shard_independent():
    assert env_seed(0, 5) == 5 and env_seed(2, 7) == 2 * 1_000_003 + 7
    # rank r of N with B envs each owns global indices r*B .. r*B+B-1: same streams whatever N is
    B = 4
    all8 = [env_seed(0, i) for i in range(8)]
    assert [env_seed(0, 1 * B + i) for i in range(B)] == all8[4:8]


def test_instance_file_roundtrip(tmp_path):
    from pcbenv import io
    cfg = named_config("c4")
    packed = pack_instances(cfg, [InstanceStream(cfg, s).next() for s in range(7)])
    p = str(tmp_path / "inst.pcbi")
    io.save_instances(p, cfg, packed)
    cfg2, back = io.load_instances(p)
    assert cfg2 == cfg and np.array_equal(back, packed)
    with open(p, "r+b") as f:
        f.write(b"XXXX")
    with pytest.raises(ValueError):
        io.load_instances(p)


@pytest.mark.skipif(not os.path.isdir("/root/reference/environment"), reason="reference tree not present (build container only)")
def test_episode_export_builds_reference_components():
    """utils/visualization/csv_utils.py:11-25 pickles `components` + `actions`; the export rebuilds those objects
    with the reference's own classes and must land every pin where the oracle's features say it is."""
    import sys
    sys.dont_write_bytecode = True
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(repo, "oracle", "refshim"), "/root/reference"]
    try:
        from environment.dummy_env_rectangular_pin_spatial import Component, Pin
    finally:
        del sys.path[:2]
    from oracle import oracle as orc
    from pcbenv import io
    meta, cfg, eps = load_case("spatial_small_centroid")
    e = next(ep for ep in eps if ep.done[len(ep.actions) - 2] and not np.isnan(ep.info[len(ep.actions) - 2, 0]) and len(ep.actions) > 3)
    valid_actions = e.actions[:e.instance.num_components]
    comps, acts = io.episode_to_reference_objects(e.instance, valid_actions, Component, Pin)
    env = orc.OracleEnv(cfg)
    env.reset(instance=e.instance)
    for a in valid_actions:
        obs, _, _, _ = env.step(a)
    feats = obs["all_pins_num_feature"]
    for c in comps:
        assert tuple(c.position) == (obs["all_components_feature"][c.comp_id, 2], obs["all_components_feature"][c.comp_id, 3])
        for pin in c.pins:
            assert list(feats[pin.pin_id]) == [pin.relative_x, pin.relative_y, pin.absolute_x, pin.absolute_y]


def test_expand_compact_features_is_the_reference_arithmetic():
    """pcbenv_compact_features -> float64: every element an exact small integer, column 4 of all_components_feature the
    ONE division the reference does (area / grid_area, S:203-239) on the same operands."""
    import torch
    from pcbenv.batched_env import COMPACT_DTYPES, FEATURE_KEYS, expand_compact_features, obs_spec
    cfg = named_config("c4")
    spec = obs_spec(cfg)
    comp = torch.zeros((2, 3) + spec["all_components_feature"][0], dtype=torch.int16)
    comp[..., 0] = 5; comp[..., 1] = 3; comp[..., 2] = -1; comp[..., 3] = 17; comp[..., 4] = 15; comp[..., 5:] = -1
    comp[0, 0, 0, 5] = 47
    num = torch.full((2, 3) + spec["all_pins_num_feature"][0], -1, dtype=torch.int8)
    out = expand_compact_features(cfg, {"all_components_feature": comp, "all_pins_num_feature": num})
    assert out["all_components_feature"].dtype == torch.float64 and out["all_pins_num_feature"].dtype == torch.float64
    assert float(out["all_components_feature"][1, 2, 3, 4]) == 15 / (64 * 64) and float(out["all_components_feature"][0, 0, 0, 5]) == 47.0
    assert float(out["all_components_feature"][0, 0, 0, 2]) == -1.0 and bool((out["all_pins_num_feature"] == -1.0).all())
    assert set(COMPACT_DTYPES) == set(FEATURE_KEYS)
