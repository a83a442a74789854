"""pcbenv: MI355X-native batched PCB component-placement environments (host side)."""
from .config import (EnvConfig, KIND_PIN, KIND_RECT, KIND_SPATIAL, KIND_SQUARE, named_config)  # noqa: F401
from .instances import Instance, InstanceStream, env_seed, pack_instances, instance_stride  # noqa: F401
from .wrappers import flat_to_tuple, tuple_to_flat  # noqa: F401,E402
from .factory import config_from_env_config  # noqa: F401,E402


def __getattr__(name):
    # the device-backed classes import torch + libpcbenv.so lazily (host-only tools stay light)
    if name in ("BatchedPlacementEnv", "obs_spec"):
        from . import batched_env
        return getattr(batched_env, name)
    if name == "SingleEnvAdapter":
        from .single_env import SingleEnvAdapter
        return SingleEnvAdapter
    if name in ("init_env", "create_env"):
        from . import factory
        return getattr(factory, name)
    raise AttributeError(name)
