"""PyTorch restatement of the policy network the spatial environment feeds
(`agent/models/rectangle_pin_spatial_model.py:14-272`, `model_building_blocks.py:11-179`,
`square_model.py:60-140`), layer for layer:

* `encode_grid` (square_model.py:91-116) and `encode_pin_grid` (rectangle_pin_spatial_model.py:144-166):
  `ConvBlocks` = `num_conv_blocks` x [Conv2D(valid) + BatchNormalization + activation (+ max pool)], flattened;
* `encode_component_grid` (:168-228): one `ConvBlocks` PER COMPONENT (independent weights: the reference builds
  them in a python loop -- a grouped convolution here) with `conv_padding_component_grid`, flattened, concatenated
  with the one-hot (4 classes) `placement_mask`, then `Attention` (model_building_blocks.py:145-179): three
  Dense(hidden) maps, `softmax(Q K^T)` with NO 1/sqrt(d) scaling, `weights @ V`, and a ReLU on the output;
* `Dense(action_space.n)` logits and `Dense(1)` value on the concatenated encoding (:100-142);
* `logits += max(log(action_mask), float32.min)` (:265-270).

Hyper-parameters default to the shipped `agent/config/rectangle_pin_spatial_model.json`
(`REFERENCE_CUSTOM_MODEL_CONFIG`; tests/golden/model_config_spatial.json is that file's content, compared in
tests/test_policy_cpu.py together with the closed-form parameter count and every layer's output shape).
It consumes the device-resident uint8 observation tensors directly (cast to float on the fly).  The reference
trains it with RLlib 2.2.0 / TensorFlow (not importable here): NUMERICAL parity is **unpinned**; structure
(layers, shapes, parameter count, masking) is what is pinned.  Flatten order differs from Keras (NCHW vs NHWC):
a fixed permutation of the inputs of the following Dense layer, not a different function class.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from .config import EnvConfig, KIND_SPATIAL

# agent/config/rectangle_pin_spatial_model.json "model.custom_model_config" (the keys this graph reads)
REFERENCE_CUSTOM_MODEL_CONFIG = {
    "num_conv_blocks": 2, "num_conv_filters": 3, "conv_kernel_size": 3, "activation": "relu",
    "max_pool": False, "max_pool_kernel_size": 2,
    "num_conv_blocks_component_grid": 1, "num_conv_filters_component_grid": 3, "conv_kernel_size_component_grid": 3,
    "activation_component_grid": "relu", "max_pool_component_grid": False, "max_pool_kernel_size_component_grid": 3,
    "conv_padding_component_grid": "same", "component_attn_hidden_size": 16,
}


def _activation(name: str) -> nn.Module:
    if name != "relu":  # every shipped config uses relu
        raise ValueError(f"activation {name!r} is not used by any reference config")
    return nn.ReLU()


def _conv_out(n: int, k: int, padding: str) -> int:
    return n if padding == "same" else n - k + 1


class ConvBlocks(nn.Module):
    """model_building_blocks.py:11-143: `blocks` x [Conv2D + BatchNormalization + activation (+ max_pool2d VALID)].
    `groups` > 1 = that many independent copies side by side (the per-component blocks)."""

    def __init__(self, in_ch: int, blocks: int, filters: int, k: int, padding: str = "valid", activation: str = "relu",
                 max_pool: bool = False, max_pool_k: int = 4, groups: int = 1):
        super().__init__()
        layers, c = [], in_ch
        for _ in range(blocks):
            layers += [nn.Conv2d(groups * c, groups * filters, k, padding=(padding if padding == "same" else 0), groups=groups),
                       nn.BatchNorm2d(groups * filters, eps=1e-3, momentum=0.01),  # Keras defaults: epsilon 1e-3, momentum 0.99
                       _activation(activation)]
            if max_pool:
                layers.append(nn.MaxPool2d(max_pool_k, max_pool_k))
            c = filters
        self.net = nn.Sequential(*layers)
        self.blocks, self.k, self.padding, self.max_pool, self.max_pool_k, self.filters = blocks, k, padding, max_pool, max_pool_k, filters

    def out_hw(self, h: int, w: int):
        for _ in range(self.blocks):
            h, w = _conv_out(h, self.k, self.padding), _conv_out(w, self.k, self.padding)
            if self.max_pool:
                h, w = h // self.max_pool_k, w // self.max_pool_k
        return h, w

    def forward(self, x):
        return self.net(x)


class Attention(nn.Module):
    """model_building_blocks.py:145-179 -- unscaled dot-product self-attention, ReLU on the output."""

    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.q, self.k, self.v = nn.Linear(dim, hidden), nn.Linear(dim, hidden), nn.Linear(dim, hidden)

    def forward(self, x):
        q, k, v = self.q(x), self.k(x), self.v(x)
        w = torch.softmax(q @ k.transpose(1, 2), dim=-1)  # :173-174, no 1/sqrt(d)
        return torch.relu(w @ v)                           # :176-177


class SpatialPolicy(nn.Module):
    def __init__(self, cfg: EnvConfig, model_config: Optional[Dict] = None):
        super().__init__()
        assert cfg.kind == KIND_SPATIAL
        mc = dict(REFERENCE_CUSTOM_MODEL_CONFIG)
        mc.update(model_config or {})
        self.cfg, self.model_config = cfg, mc
        H, W, K, C = cfg.height, cfg.width, cfg.max_num_nets + 1, cfg.max_num_components
        common = dict(blocks=mc["num_conv_blocks"], filters=mc["num_conv_filters"], k=mc["conv_kernel_size"],
                      activation=mc["activation"], max_pool=mc["max_pool"], max_pool_k=mc["max_pool_kernel_size"])
        self.grid_net = ConvBlocks(1, **common)
        self.pin_net = ConvBlocks(K, **common)
        self.comp_net = ConvBlocks(K, blocks=mc["num_conv_blocks_component_grid"], filters=mc["num_conv_filters_component_grid"],
                                   k=mc["conv_kernel_size_component_grid"], padding=mc["conv_padding_component_grid"],
                                   activation=mc["activation_component_grid"], max_pool=mc["max_pool_component_grid"],
                                   max_pool_k=mc["max_pool_kernel_size_component_grid"], groups=C)
        gh, gw = self.grid_net.out_hw(H, W)
        self.enc_grid = mc["num_conv_filters"] * gh * gw
        ch, cw = self.comp_net.out_hw(cfg.max_component_h, cfg.max_component_w)
        self.comp_dim = mc["num_conv_filters_component_grid"] * ch * cw + 4
        self.attn = Attention(self.comp_dim, mc["component_attn_hidden_size"])
        self.enc_dim = 2 * self.enc_grid + C * mc["component_attn_hidden_size"]
        self.num_actions = cfg.num_orientations * H * W
        self.logits = nn.Linear(self.enc_dim, self.num_actions)
        self.value = nn.Linear(self.enc_dim, 1)

    def encode(self, obs):
        """-> (encoding [B, enc_dim], {name: intermediate tensor}) -- the intermediates are what the shape test checks."""
        B = obs["grid"].shape[0]
        C, K = self.cfg.max_num_components, self.cfg.max_num_nets + 1
        g = self.grid_net(obs["grid"].float().unsqueeze(1))
        p = self.pin_net(obs["pin_grid"].float().permute(0, 3, 1, 2))
        cg = obs["component_grid"].float().permute(0, 1, 4, 2, 3).reshape(B, C * K, self.cfg.max_component_h, self.cfg.max_component_w)
        ce = self.comp_net(cg).reshape(B, C, -1)
        pm = torch.nn.functional.one_hot(obs["placement_mask"].long(), 4).float()
        comp_in = torch.cat([ce, pm], dim=2)
        comp = self.attn(comp_in)
        enc = torch.cat([g.flatten(1), p.flatten(1), comp.flatten(1)], dim=1)
        return enc, {"processed_grid": g, "processed_pin_grid": p, "components_encodings": comp_in, "component_attn_output": comp}

    def forward(self, obs):
        """obs: the BatchedPlacementEnv observation dict (device tensors) -> (masked logits [B, O*H*W], value [B])."""
        enc, _ = self.encode(obs)
        logits = self.logits(enc)
        mask = obs["action_mask"].reshape(enc.shape[0], -1).float()
        logits = logits + torch.clamp(torch.log(mask), min=torch.finfo(torch.float32).min)
        return logits, self.value(enc).squeeze(-1)


def reference_parameter_count(cfg: EnvConfig, mc: Dict) -> Dict[str, int]:
    """Trainable parameters of the reference graph in closed form (Keras: Conv2D k*k*in*out + out, BatchNormalization
    gamma + beta, Dense in*out + out), per block of `RectanglePinSpatialModel` (rectangle_pin_spatial_model.py:60-272)."""
    H, W, K, C = cfg.height, cfg.width, cfg.max_num_nets + 1, cfg.max_num_components

    def blocks(in_ch, n, f, k):
        total, c = 0, in_ch
        for _ in range(n):
            total += k * k * c * f + f + 2 * f
            c = f
        return total

    def shrink(n, blocks_, k, padding, pool, pk):
        for _ in range(blocks_):
            n = _conv_out(n, k, padding)
            if pool:
                n //= pk
        return n
    nb, nf, ks = mc["num_conv_blocks"], mc["num_conv_filters"], mc["conv_kernel_size"]
    gh = shrink(H, nb, ks, "valid", mc["max_pool"], mc["max_pool_kernel_size"])
    gw = shrink(W, nb, ks, "valid", mc["max_pool"], mc["max_pool_kernel_size"])
    cb, cf, ck = mc["num_conv_blocks_component_grid"], mc["num_conv_filters_component_grid"], mc["conv_kernel_size_component_grid"]
    ch = shrink(cfg.max_component_h, cb, ck, mc["conv_padding_component_grid"], mc["max_pool_component_grid"], mc["max_pool_kernel_size_component_grid"])
    cw = shrink(cfg.max_component_w, cb, ck, mc["conv_padding_component_grid"], mc["max_pool_component_grid"], mc["max_pool_kernel_size_component_grid"])
    comp_dim, hid = cf * ch * cw + 4, mc["component_attn_hidden_size"]
    enc = 2 * nf * gh * gw + C * hid
    A = cfg.num_orientations * H * W
    return {"grid": blocks(1, nb, nf, ks), "pin_grid": blocks(K, nb, nf, ks), "components": C * blocks(K, cb, cf, ck),
            "attention": 3 * (comp_dim * hid + hid), "logits": enc * A + A, "value": enc + 1, "encoding_dim": enc}
