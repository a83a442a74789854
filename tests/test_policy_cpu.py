"""Host-side (CPU) checks of the policy restatement and the masking helpers -- no environment, no GPU."""
import torch

from pcbenv import EnvConfig
from pcbenv.policy import SpatialPolicy
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
