"""`init_env` / `create_env`: the reference's environment factory for this path.

Mirrors `utils/agent/utils.py:317-418` of the reference: the same `env_config`
dictionary (keys of `agent/config/*.json`), the same `type` strings, the same
positional argument order -- including quirk Q5: `weight_num_intersections` is
never forwarded, so it is always the constructor default 0.5 whatever the JSON
says (`utils/agent/utils.py:352-390`).
"""
from __future__ import annotations

from typing import Mapping

from .config import EnvConfig

SQUARE_TYPES = ("square",)
RECT_TYPES = ("rectangle", "rectangle_factorized")
PIN_TYPES = ("rectangle_pin", "rectangle_factorized_pin", "rectangle_pin_attn_component",
             "rectangle_pin_attn_all", "rectangle_pin_attn_all_no_grid", "rectangle_pin_all_attn_factorized")
SPATIAL_TYPES = ("rectangle_spatial_pin",)
# create_env flattens action + action_mask for these types only (`utils/agent/utils.py:404-416`)
FLATTENED_TYPES = ("square", "rectangle", "rectangle_pin", "rectangle_pin_attn_component",
                   "rectangle_pin_attn_all", "rectangle_pin_attn_all_no_grid", "rectangle_spatial_pin")

_PIN_KEYS = ("height", "width", "net_distribution", "pin_spread", "min_component_w", "max_component_w",
             "min_component_h", "max_component_h", "max_num_components", "min_num_components", "min_num_nets",
             "max_num_nets", "max_num_pins_per_net", "min_num_pins_per_net", "reward_type", "reward_beam_width",
             "weight_wirelength")


def config_from_env_config(env_config: Mapping) -> EnvConfig:
    t = env_config["type"]
    if t in SQUARE_TYPES:
        return EnvConfig.square(env_config["height"], env_config["width"], env_config["component_n"])
    if t in RECT_TYPES:
        return EnvConfig.rect(*(env_config[k] for k in (
            "height", "width", "min_component_w", "max_component_w", "min_component_h", "max_component_h",
            "max_num_components", "min_num_components")))
    if t in PIN_TYPES:
        return EnvConfig.pin(*(env_config[k] for k in _PIN_KEYS))
    if t in SPATIAL_TYPES:
        return EnvConfig.spatial(*(env_config[k] for k in _PIN_KEYS))
    raise KeyError(f"unknown environment type {t!r}")


def init_env(env_config: Mapping, num_envs: int = 1, **kw):
    """-> BatchedPlacementEnv (num_envs > 1) or SingleEnvAdapter (num_envs == 1, gym-style)."""
    from .batched_env import BatchedPlacementEnv
    from .single_env import SingleEnvAdapter
    cfg = config_from_env_config(env_config)
    if num_envs == 1:
        return SingleEnvAdapter(cfg, **kw)
    return BatchedPlacementEnv(cfg, num_envs, **kw)


def create_env(env_config: Mapping, num_envs: int = 1, **kw):
    """`init_env` + the reference's two flattening wrappers where `create_env` applies them."""
    from .wrappers import FlatteningActionMaskObservationWrapper, FlatteningActionWrapper
    env = init_env(env_config, num_envs, **kw)
    if env_config["type"] in FLATTENED_TYPES:
        env = FlatteningActionWrapper(FlatteningActionMaskObservationWrapper(env))
    return env
