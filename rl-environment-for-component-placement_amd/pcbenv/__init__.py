"""pcbenv: MI355X-native batched PCB component-placement environments (host side)."""
from .config import (EnvConfig, KIND_PIN, KIND_RECT, KIND_SPATIAL, KIND_SQUARE, named_config)  # noqa: F401
from .instances import Instance, InstanceStream, env_seed, pack_instances, instance_stride  # noqa: F401
