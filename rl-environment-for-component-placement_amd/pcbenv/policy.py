"""PyTorch restatement of the policy network the spatial environment feeds
(`agent/models/rectangle_pin_spatial_model.py:14-272`, `model_building_blocks.py:11-179`,
`square_model.py:60-140`): ConvBlocks (valid conv + batch norm + ReLU) on `grid` and `pin_grid`, a ConvBlock per
component on `component_grid`, the one-hot `placement_mask`, self-attention over components, dense logits over
the flat action space and a dense value head; logits are masked the reference's way,
`logits += max(log(action_mask), float32.min)`.  It consumes the device-resident uint8 observation tensors
directly (cast to float on the fly).  Hyper-parameters default to `agent/config/rectangle_pin_spatial_model.json`.
The reference trains it with RLlib 2.2.0 (not in the tree): numerical parity with that is **unpinned**; this
module exists so that a PPO loop can be closed around the path on the GPU (see ppo.py).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .config import EnvConfig, KIND_SPATIAL


def conv_blocks(in_ch: int, blocks: int, filters: int, k: int, padding: int = 0) -> nn.Sequential:
    layers, c = [], in_ch
    for _ in range(blocks):
        layers += [nn.Conv2d(c, filters, k, padding=padding), nn.BatchNorm2d(filters), nn.ReLU()]
        c = filters
    return nn.Sequential(*layers)


class SelfAttention(nn.Module):
    """model_building_blocks.py:145-179: single-head dot-product self-attention with a hidden size."""

    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.q, self.k, self.v = nn.Linear(dim, hidden), nn.Linear(dim, hidden), nn.Linear(dim, hidden)

    def forward(self, x):
        q, k, v = self.q(x), self.k(x), self.v(x)
        w = torch.softmax(q @ k.transpose(1, 2) / (q.shape[-1] ** 0.5), dim=-1)
        return w @ v


class SpatialPolicy(nn.Module):
    def __init__(self, cfg: EnvConfig, num_conv_blocks: int = 2, num_conv_filters: int = 3, conv_kernel_size: int = 3,
                 comp_conv_filters: int = 3, comp_conv_kernel: int = 1, attn_hidden: int = 8):
        super().__init__()
        assert cfg.kind == KIND_SPATIAL
        self.cfg = cfg
        H, W, K, C = cfg.height, cfg.width, cfg.max_num_nets + 1, cfg.max_num_components
        self.grid_net = conv_blocks(1, num_conv_blocks, num_conv_filters, conv_kernel_size)
        self.pin_net = conv_blocks(K, num_conv_blocks, num_conv_filters, conv_kernel_size)
        # one ConvBlock per component (independent weights, as in the reference's python loop) = grouped conv
        self.comp_net = nn.Sequential(nn.Conv2d(C * K, C * comp_conv_filters, comp_conv_kernel, groups=C),
                                      nn.BatchNorm2d(C * comp_conv_filters), nn.ReLU())
        shrink = num_conv_blocks * (conv_kernel_size - 1)
        enc_grid = num_conv_filters * (H - shrink) * (W - shrink)
        ch, cw = cfg.max_component_h - (comp_conv_kernel - 1), cfg.max_component_w - (comp_conv_kernel - 1)
        self.comp_dim = comp_conv_filters * ch * cw + 4
        self.attn = SelfAttention(self.comp_dim, attn_hidden)
        enc = 2 * enc_grid + C * attn_hidden
        self.num_actions = cfg.num_orientations * H * W
        self.logits = nn.Linear(enc, self.num_actions)
        self.value = nn.Linear(enc, 1)

    def forward(self, obs):
        """obs: the BatchedPlacementEnv observation dict (device tensors) -> (masked logits [B, O*H*W], value [B])."""
        B = obs["grid"].shape[0]
        C, K = self.cfg.max_num_components, self.cfg.max_num_nets + 1
        g = self.grid_net(obs["grid"].float().unsqueeze(1)).flatten(1)
        p = self.pin_net(obs["pin_grid"].float().permute(0, 3, 1, 2)).flatten(1)
        cg = obs["component_grid"].float().permute(0, 1, 4, 2, 3).reshape(B, C * K, self.cfg.max_component_h, self.cfg.max_component_w)
        ce = self.comp_net(cg).reshape(B, C, -1)
        pm = torch.nn.functional.one_hot(obs["placement_mask"].long(), 4).float()
        comp = self.attn(torch.cat([ce, pm], dim=2)).flatten(1)
        enc = torch.cat([g, p, comp], dim=1)
        logits = self.logits(enc)
        mask = obs["action_mask"].reshape(B, -1).float()
        logits = logits + torch.clamp(torch.log(mask), min=torch.finfo(torch.float32).min)
        return logits, self.value(enc).squeeze(-1)
