"""Environment configuration: one POD mirrored by `pcbenv_config` in include/pcbenv.h.

Field names and validation follow the constructors of the four reference
environments:

* square : `environment/dummy_env_square.py:37-72`
* rect   : `environment/dummy_env_rectangular.py:152-251`
* pin    : `environment/dummy_env_rectangular_pin.py:396-563` + `validate_env_params` :565-641
* spatial: `environment/dummy_env_rectangular_pin_spatial.py:396-607`

Bad parameters raise `ValueError` exactly where the reference does; the C ABI
repeats the same checks and returns `PCBENV_EINVAL` (include/pcbenv.h).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

KIND_SQUARE, KIND_RECT, KIND_PIN, KIND_SPATIAL = 0, 1, 2, 3
KIND_NAMES = {"square": KIND_SQUARE, "rect": KIND_RECT, "rectangular": KIND_RECT,
              "pin": KIND_PIN, "rectangular_pin": KIND_PIN,
              "spatial": KIND_SPATIAL, "pin_spatial": KIND_SPATIAL, "rectangular_pin_spatial": KIND_SPATIAL}
REWARD_TYPES = {"beam": 0, "centroid": 1, "both": 2}

# device limits of the HIP path (include/pcbenv.h PCBENV_MAX_*)
MAX_SIDE = 128
MAX_COMPONENTS = 64
MAX_PINS = 256
MAX_NETS = 32
MAX_PINS_PER_NET = 16
MAX_PINS_PER_COMPONENT = 64
MAX_BEAM_WIDTH = 4


@dataclass
class EnvConfig:
    kind: int
    height: int
    width: int
    # rect / pin / spatial
    min_component_w: int = 1
    max_component_w: int = 1
    min_component_h: int = 1
    max_component_h: int = 1
    max_num_components: int = 1
    min_num_components: int = 1
    # pin / spatial
    net_distribution: int = 0
    pin_spread: int = 0
    min_num_nets: int = 0
    max_num_nets: int = 0
    max_num_pins_per_net: int = 0
    min_num_pins_per_net: int = 2
    reward_type: str = "both"
    reward_beam_width: int = 2
    weight_wirelength: float = 0.5
    weight_num_intersections: float = 0.5
    # square
    component_n: int = 1

    # ---- reference constructors (positional order preserved) ---------------
    @staticmethod
    def square(height: int, width: int, component_n: int) -> "EnvConfig":
        c = EnvConfig(KIND_SQUARE, height, width, component_n=component_n,
                      min_component_w=component_n, max_component_w=component_n,
                      min_component_h=component_n, max_component_h=component_n)
        c.validate()
        return c

    @staticmethod
    def rect(height, width, min_component_w, max_component_w, min_component_h, max_component_h,
             max_num_components, min_num_components) -> "EnvConfig":
        c = EnvConfig(KIND_RECT, height, width, min_component_w, max_component_w, min_component_h,
                      max_component_h, max_num_components, min_num_components)
        c.validate()
        return c

    @staticmethod
    def _pin_like(kind, height, width, net_distribution, pin_spread, min_component_w, max_component_w,
                  min_component_h, max_component_h, max_num_components, min_num_components,
                  min_num_nets, max_num_nets, max_num_pins_per_net, min_num_pins_per_net=2,
                  reward_type="both", reward_beam_width=2, weight_wirelength=0.5,
                  weight_num_intersections=0.5) -> "EnvConfig":
        c = EnvConfig(kind, height, width, min_component_w, max_component_w, min_component_h,
                      max_component_h, max_num_components, min_num_components,
                      net_distribution, pin_spread, min_num_nets, max_num_nets,
                      max_num_pins_per_net, min_num_pins_per_net, reward_type, reward_beam_width,
                      weight_wirelength, weight_num_intersections)
        c.validate()
        # the reference clips the two complexity knobs to [0, 9] after validation
        # (`..._pin.py:467-468`, `..._spatial.py:450-451`)
        c.net_distribution = max(0, min(9, c.net_distribution))
        c.pin_spread = max(0, min(9, c.pin_spread))
        return c

    @staticmethod
    def pin(*a, **k) -> "EnvConfig":
        return EnvConfig._pin_like(KIND_PIN, *a, **k)

    @staticmethod
    def spatial(*a, **k) -> "EnvConfig":
        return EnvConfig._pin_like(KIND_SPATIAL, *a, **k)

    # ---- derived sizes ------------------------------------------------------
    @property
    def area(self) -> int:
        return self.height * self.width

    @property
    def num_orientations(self) -> int:
        return {KIND_SQUARE: 1, KIND_RECT: 2}.get(self.kind, 4)

    @property
    def max_num_pins_per_component(self) -> int:  # `..._spatial.py:464`
        return self.max_component_h * self.max_component_w

    @property
    def max_total_pins(self) -> int:
        if self.kind not in (KIND_PIN, KIND_SPATIAL):
            return 0
        return min(self.max_num_pins_per_net * self.max_num_nets,
                   self.max_num_components * self.max_num_pins_per_component)

    def cell_tensor_bytes_per_step(self, incremental: bool = False) -> int:
        """uint8 cell tensors one environment rewrites per step (grid, action_mask, pin_grid): what pcbenv_create
        compares with the Infinity Cache size to choose the store policy."""
        planes = self.num_orientations if incremental else 1 + self.num_orientations + (
            self.max_num_nets + 1 if self.kind == KIND_SPATIAL else 0)
        return self.area * planes

    @property
    def reward_type_code(self) -> int:
        return REWARD_TYPES.get(self.reward_type, -1)

    # ---- constants of the reward (a15, `find_reward` :839-850) -------------
    @property
    def max_num_intersections(self) -> float:
        v = 0.5 * (self.max_num_pins_per_net ** 2) * self.max_num_nets * (self.max_num_nets - 1)
        return float(int(v)) if self.kind == KIND_PIN else v  # pin: int(...) `..._pin.py:822-830`

    @property
    def max_wirelength(self) -> float:
        dist = math.sqrt(float(self.height * self.height + self.width * self.width))
        total = 0.5 * dist * (self.max_num_nets * self.max_num_pins_per_net)
        return total / (self.height + self.width) if self.kind == KIND_SPATIAL else total  # Q3

    @property
    def wirelength_norm(self) -> float:
        return float(self.height + self.width)

    @property
    def intersections_norm(self) -> float:
        a = ((self.min_component_h + self.max_component_h) / 2.0) * \
            ((self.min_component_w + self.max_component_w) / 2.0) * \
            ((self.min_num_components + self.max_num_components) / 2.0)
        b = ((self.min_num_pins_per_net + self.max_num_pins_per_net) / 2.0) * \
            ((self.min_num_nets + self.max_num_nets) / 2.0)
        return min(a, b)

    # ---- validation ---------------------------------------------------------
    def validate(self) -> None:
        k = self.kind
        if k not in (KIND_SQUARE, KIND_RECT, KIND_PIN, KIND_SPATIAL):
            raise ValueError("unknown environment kind")
        if self.height < 0 or self.width < 0:
            raise ValueError("Grid size must not be negative.")
        if k == KIND_SQUARE:
            if self.component_n > self.height or self.component_n > self.width:
                raise ValueError("Component size must not exceed the grid size.")
        else:
            # Q6: rect/spatial compare w against height and h against width; pin compares like with like
            if k == KIND_PIN:
                too_big = self.max_component_w > self.width or self.max_component_h > self.height
            else:
                too_big = self.max_component_w > self.height or self.max_component_h > self.width
            if too_big:
                raise ValueError("Component size must not exceed the grid size.")
            if self.min_component_w < 1 or self.min_component_h < 1:
                raise ValueError("Component size must be at least 1.")
            if self.max_num_components < 1 or self.max_num_components > self.area:
                raise ValueError("Number of components must be in [1, grid area].")
        if k == KIND_PIN:
            if self.min_num_pins_per_net > self.max_num_pins_per_net:
                raise ValueError("min_num_pins_per_net must not exceed max_num_pins_per_net.")
            if self.min_num_pins_per_net < 2:
                raise ValueError("min_num_pins_per_net must be at least 2.")
            if (self.min_num_pins_per_net * self.min_num_nets
                    > self.min_component_w * self.min_component_h * self.min_num_components):
                raise ValueError("min_num_pins_per_net * min_num_nets exceeds the minimum total component area.")
            if not isinstance(self.reward_beam_width, int) or self.reward_beam_width < 1:
                raise ValueError("Beam width must be a positive integer.")
            if self.reward_type not in REWARD_TYPES:
                raise ValueError("Reward type must be 'beam', 'centroid' or 'both'.")
        if k == KIND_SPATIAL:
            if self.reward_type not in REWARD_TYPES:
                raise ValueError("Reward type must be 'beam', 'centroid' or 'both'.")
            if (not isinstance(self.reward_beam_width, int) or self.reward_beam_width < 2
                    or self.reward_beam_width > self.max_num_pins_per_net):
                raise ValueError("Beam width must be an integer in [2, max_num_pins_per_net].")
            if not isinstance(self.weight_wirelength, float):
                raise ValueError("weight_wirelength must be a float.")
            if self.weight_wirelength < 0:
                raise ValueError("weight_wirelength must not be negative.")

    def check_device_limits(self) -> None:
        """Limits of the HIP path (not of the reference); exceeded -> ValueError."""
        if not (1 <= self.height <= MAX_SIDE and 1 <= self.width <= MAX_SIDE):
            raise ValueError(f"HIP path supports grids up to {MAX_SIDE}x{MAX_SIDE}")
        if self.kind == KIND_SQUARE:
            return
        if max(self.max_component_h, self.max_component_w) > min(self.height, self.width):
            # the reference raises inside scipy.signal.convolve2d('valid') at run time for such
            # components (kernel larger than the grid in one dimension); rejected up front here
            raise ValueError("a component side exceeds the shorter grid side")
        if self.max_num_components > MAX_COMPONENTS:
            raise ValueError(f"HIP path supports at most {MAX_COMPONENTS} components")
        if self.kind in (KIND_PIN, KIND_SPATIAL):
            if self.max_total_pins > MAX_PINS or self.max_num_nets > MAX_NETS:
                raise ValueError(f"HIP path supports at most {MAX_PINS} pins / {MAX_NETS} nets")
            if self.max_num_pins_per_net > MAX_PINS_PER_NET:
                raise ValueError(f"HIP path supports at most {MAX_PINS_PER_NET} pins per net")
            if self.max_num_pins_per_component > MAX_PINS_PER_COMPONENT:
                raise ValueError(f"HIP path supports at most {MAX_PINS_PER_COMPONENT} pins per component")
            if self.reward_type in ("beam", "both") and self.reward_beam_width > MAX_BEAM_WIDTH:
                raise ValueError(f"HIP path supports beam widths up to {MAX_BEAM_WIDTH}")


# the five benchmark / parity parameterisations
def named_config(name: str, reward_type: str = "centroid") -> EnvConfig:
    if name == "c1":
        return EnvConfig.square(8, 8, 3)
    if name == "c2":
        return EnvConfig.rect(32, 32, 2, 6, 2, 6, 8, 8)
    if name == "c3":
        return EnvConfig.pin(64, 64, 9, 9, 2, 6, 2, 6, 16, 16, 8, 8, 6, 6, reward_type, 2, 0.5)
    if name == "c4":
        return EnvConfig.spatial(64, 64, 9, 9, 2, 6, 2, 6, 16, 16, 8, 8, 6, 6, reward_type, 2, 0.5)
    if name == "c5":
        return EnvConfig.spatial(128, 128, 9, 9, 2, 8, 2, 8, 32, 32, 16, 16, 8, 8, reward_type, 2, 0.5)
    raise KeyError(name)
