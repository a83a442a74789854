"""Host-side (CPU) checks of the policy restatement and the masking helpers -- no environment, no GPU."""
import json
import os

import torch

from pcbenv import EnvConfig
from pcbenv.policy import REFERENCE_CUSTOM_MODEL_CONFIG, Attention, SpatialPolicy, reference_parameter_count

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
from pcbenv.rollout import masked_logits, sample_masked_categorical


def _fake_obs(cfg, B):
    H, W, K, C = cfg.height, cfg.width, cfg.max_num_nets + 1, cfg.max_num_components
    g = torch.Generator().manual_seed(0)
    mask = (torch.rand((B, 4, H, W), generator=g) < 0.2).to(torch.uint8)
    mask[:, 0, 0, 0] = 1
    return {"grid": (torch.rand((B, H, W), generator=g) < 0.3).to(torch.uint8),
            "pin_grid": (torch.rand((B, H, W, K), generator=g) < 0.1).to(torch.uint8),
            "component_grid": (torch.rand((B, C, cfg.max_component_h, cfg.max_component_w, K), generator=g) < 0.3).to(torch.uint8),
            "placement_mask": torch.randint(0, 4, (B, C), generator=g).double(),
            "action_mask": mask}


def test_policy_masks_logits_like_the_reference_models():
    """`logits += max(log(action_mask), float32.min)` (agent/models/square_model.py:137-139)."""
    cfg = EnvConfig.spatial(10, 10, 9, 9, 2, 2, 2, 2, 5, 5, 3, 3, 6, 6, "centroid", 2, 0.75)
    pol = SpatialPolicy(cfg).eval()
    obs = _fake_obs(cfg, 6)
    logits, value = pol(obs)
    assert logits.shape == (6, 4 * 10 * 10) and value.shape == (6,)
    flat = obs["action_mask"].reshape(6, -1).bool()
    assert bool((logits[~flat] < -1e30).all()) and bool(torch.isfinite(logits[flat]).all())
    a = torch.distributions.Categorical(logits=logits).sample()
    assert bool(flat[torch.arange(6), a].all())           # never an illegal action
    loss = logits[flat].sum() + value.sum()
    loss.backward()                                        # gradients flow through every block
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in pol.parameters())


def test_masked_categorical_helpers():
    logits = torch.zeros(4, 12)
    mask = torch.zeros(4, 12, dtype=torch.uint8)
    mask[:, 5] = 1
    mask[0, 7] = 1
    ml = masked_logits(logits, mask)
    assert bool((ml[mask == 0] < -1e30).all()) and bool((ml[mask == 1] == 0).all())
    g = torch.Generator().manual_seed(1)
    for _ in range(20):
        a = sample_masked_categorical(logits, mask, g)
        assert bool(mask[torch.arange(4), a.long()].all())


def _shipped():
    """agent/config/rectangle_pin_spatial_model.json as recorded by tests/golden/make_golden.py (a data file)."""
    with open(os.path.join(GOLDEN, "model_config_spatial.json")) as f:
        ref = json.load(f)
    e = ref["env_config"]
    cfg = EnvConfig.spatial(e["height"], e["width"], e["net_distribution"], e["pin_spread"], e["min_component_w"], e["max_component_w"],
                            e["min_component_h"], e["max_component_h"], e["max_num_components"], e["min_num_components"],
                            e["min_num_nets"], e["max_num_nets"], e["max_num_pins_per_net"], e["min_num_pins_per_net"],
                            e["reward_type"], e["reward_beam_width"], e["weight_wirelength"])
    return cfg, ref["custom_model_config"]


def test_defaults_are_the_shipped_hyper_parameters():
    _, mc = _shipped()
    for k, v in REFERENCE_CUSTOM_MODEL_CONFIG.items():
        assert mc[k] == v, k


def test_parameter_count_and_layer_shapes_equal_the_reference_graph():
    """rectangle_pin_spatial_model.py:60-272 on the shipped 10x10 config, closed form: ConvBlocks(2 x [3x3 valid, 3 filters])
    on grid (1 ch) and pin_grid (4 ch): 10 -> 8 -> 6; per component ConvBlocks(1 x [3x3 same, 3 filters]) on (2, 2, 4);
    Attention(16) on [5, 3*2*2 + 4]; Dense(400) / Dense(1) on 2*3*6*6 + 5*16 = 296 features."""
    cfg, mc = _shipped()
    want = reference_parameter_count(cfg, mc)
    # the same numbers written out by hand from the Keras layer definitions
    assert want == {"grid": (9 * 1 * 3 + 3 + 6) + (9 * 3 * 3 + 3 + 6), "pin_grid": (9 * 4 * 3 + 3 + 6) + (9 * 3 * 3 + 3 + 6),
                    "components": 5 * (9 * 4 * 3 + 3 + 6), "attention": 3 * (16 * 16 + 16), "logits": 296 * 400 + 400,
                    "value": 297, "encoding_dim": 296}
    pol = SpatialPolicy(cfg, mc)
    count = lambda m: sum(p.numel() for p in m.parameters())
    assert count(pol.grid_net) == want["grid"] and count(pol.pin_net) == want["pin_grid"]
    assert count(pol.comp_net) == want["components"] and count(pol.attn) == want["attention"]
    assert count(pol.logits) == want["logits"] and count(pol.value) == want["value"]
    assert count(pol) == sum(v for k, v in want.items() if k != "encoding_dim") == 120831
    enc, mid = pol.eval().encode(_fake_obs(cfg, 3))
    assert tuple(mid["processed_grid"].shape) == (3, 3, 6, 6)         # Keras (None, 6, 6, 3)
    assert tuple(mid["processed_pin_grid"].shape) == (3, 3, 6, 6)
    assert tuple(mid["components_encodings"].shape) == (3, 5, 16)     # 3 * 2 * 2 conv features + 4 one-hot classes
    assert tuple(mid["component_attn_output"].shape) == (3, 5, 16)
    assert tuple(enc.shape) == (3, want["encoding_dim"])
    # c4 (64x64, 16 components up to 6x6, 8 nets): the dense logits layer dominates
    from pcbenv import named_config
    c4 = named_config("c4")
    w4 = reference_parameter_count(c4, REFERENCE_CUSTOM_MODEL_CONFIG)
    assert w4["encoding_dim"] == 2 * 3 * 60 * 60 + 16 * 16 and w4["logits"] == w4["encoding_dim"] * 16384 + 16384


def test_attention_is_the_reference_block():
    """model_building_blocks.py:168-177: softmax(Q K^T) with no 1/sqrt(d), ReLU on weights @ V."""
    torch.manual_seed(0)
    att = Attention(6, 4)
    x = torch.randn(2, 5, 6)
    q, k, v = att.q(x), att.k(x), att.v(x)
    want = torch.relu(torch.softmax(q @ k.transpose(1, 2), dim=-1) @ v)
    assert torch.equal(att(x), want) and bool((att(x) >= 0).all())
