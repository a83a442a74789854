"""The drop-in boundary on the device: gym spaces, reference-style loops, `env.components`, and the
record validation of pcbenv_load_instances -- all through the C ABI."""
import json
import os
import random

import numpy as np
import pytest

from golden_util import GOLDEN_DIR, MAKE
from pcbenv import EnvConfig, Instance, create_env, named_config, pack_instances
from pcbenv.batched_env import BatchedPlacementEnv
from pcbenv.single_env import SingleEnvAdapter
from pcbenv.spaces import Dict, Discrete, Tuple

pytestmark = pytest.mark.gpu


def _simulate(env, policy, n_episodes):
    """The reference's `simulate()` loop, verbatim in shape (agent/random/random_policy_square.py:25-58)."""
    episode_returns = []
    for _ in range(n_episodes):
        episode_return = 0.0
        env.reset()
        while True:
            valid_actions = np.argwhere(env.action_mask == 1).tolist()
            action = policy(valid_actions)
            _, reward, done, _ = env.step(action, verbose=False)
            episode_return += reward
            if done:
                break
        episode_returns.append(episode_return)
    return episode_returns


def test_reference_simulate_loop_runs_on_the_adapter():
    rng = random.Random(0)
    env = SingleEnvAdapter(EnvConfig.square(8, 8, 3))                        # c1
    rets = _simulate(env, rng.choice, 6)
    assert all(1.0 <= r <= 4.0 and r == int(r) for r in rets)                # 1 per placement, at most 4 fit
    env.close()
    env = SingleEnvAdapter(EnvConfig.rect(12, 12, 2, 4, 2, 4, 6, 3), seed=5)
    rets = _simulate(env, rng.choice, 6)
    assert all(3.0 <= r <= 6.0 for r in rets)
    env.close()
    env = SingleEnvAdapter(EnvConfig.spatial(10, 10, 3, 4, 2, 4, 2, 4, 6, 1, 2, 4, 5, 2, "centroid", 2, 0.5), seed=2)
    rets = _simulate(env, rng.choice, 6)
    assert all(r < 0 for r in rets)                                          # sparse terminal reward -(...)
    env.close()


@pytest.mark.parametrize("kind", ["square", "rect", "pin", "spatial"])
def test_spaces_contain_what_the_device_returns(kind):
    cfg = {"square": EnvConfig.square(8, 8, 3), "rect": named_config("c2"), "pin": named_config("c3"),
           "spatial": named_config("c4")}[kind]
    env = SingleEnvAdapter(cfg, seed=1)
    assert isinstance(env.action_space, Tuple) and isinstance(env.observation_space, Dict)
    obs = env.reset()
    rs = np.random.RandomState(0)
    for _ in range(5):
        assert set(obs) == set(env.observation_space.keys())
        for k, v in obs.items():
            if kind == "spatial" and k == "all_pins_cat_feature":
                assert not env.observation_space[k].contains(v)              # the reference declares it int32 (Q7)
            else:
                assert env.observation_space[k].contains(v), k
        legal = np.argwhere(env.action_mask == 1)
        a = tuple(int(v) for v in legal[rs.randint(len(legal))])
        assert a in env.action_space
        obs, _, done, _ = env.step(a)
        if done:
            break
    env.close()


def test_create_env_result_has_the_wrapper_spaces():
    """What RLlib's register_env(create_env) reads: Discrete(O*H*W) actions, flat action_mask Box
    (utils/environment/env_wrappers.py:28-31, :76-78)."""
    ec = {"type": "rectangle_spatial_pin", "height": 10, "width": 10, "net_distribution": 3, "pin_spread": 4,
          "min_component_w": 2, "max_component_w": 4, "min_component_h": 2, "max_component_h": 4,
          "max_num_components": 6, "min_num_components": 1, "min_num_nets": 2, "max_num_nets": 4,
          "max_num_pins_per_net": 5, "min_num_pins_per_net": 2, "reward_type": "centroid", "reward_beam_width": 2,
          "weight_wirelength": 0.5}
    env = create_env(ec)
    assert isinstance(env.action_space, Discrete) and env.action_space.n == 4 * 10 * 10
    assert env.observation_space["action_mask"].shape == (400,) and env.observation_space["grid"].shape == (10, 10)
    obs = env.reset()
    assert obs["action_mask"].shape == (400,) and env.observation_space["action_mask"].contains(obs["action_mask"])
    done, steps = False, 0
    while not done:
        a = int(np.flatnonzero(obs["action_mask"] == 1)[0])
        assert a in env.action_space and env.validate_action(a)
        obs, r, done, info = env.step(a)
        steps += 1
    assert steps >= 1 and r < 0 and set(info) == {"wirelength", "num_intersections"}
    benv = create_env(ec, num_envs=16)  # the batched flavour carries the same per-environment spaces
    assert benv.action_space == env.action_space and benv.observation_space == env.observation_space
    benv.close()
    env.close()


def test_components_view_equals_the_reference_objects():
    """`env.components` after whole episodes, against tests/golden/adapter_views.npz (dumped from the reference's
    Component / Pin objects): sizes, placed, position and every pin's relative / absolute coordinates, ids and net --
    including the pin env's pins whose feature row was overwritten (Q1).  The list is taken right after reset(), as
    utils/agent/utils.py:238 does, and read when the episode is over."""
    z = np.load(os.path.join(GOLDEN_DIR, "adapter_views.npz"))
    checked_pins = 0
    for meta in json.loads(str(z["meta"])):
        cfg = MAKE[meta["kind"]](*meta["args"])
        env = SingleEnvAdapter(cfg)
        for seed in meta["seeds"]:
            for ep in range(meta["episodes"]):
                pre = f"{meta['name']}_s{seed}_e{ep}_"
                ins = Instance(*(z[pre + k].astype(np.int64) if k != "num_nets" else int(z[pre + k]) for k in (
                    "comp_h", "comp_w", "num_nets", "pin_rel_x", "pin_rel_y", "pin_net", "pin_comp", "pin_id")))
                env.reset(instance=ins)
                comps = env.components
                for a in z[pre + "actions"]:
                    env.step(tuple(int(v) for v in a))
                got_c = np.array([[c.h, c.w, c.area, c.comp_id, int(c.placed), c.position[0], c.position[1]] for c in comps])
                assert np.array_equal(got_c, z[pre + "comp_state"]), pre
                got_p = np.array([[c.comp_id, p.relative_x, p.relative_y, p.absolute_x, p.absolute_y, p.pin_id,
                                   p.component_id, p.net_id] for c in comps for p in c.pins]).reshape(-1, 8)
                assert np.array_equal(got_p, z[pre + "pin_state"]), pre
                assert float(env.action_mask.sum()) == float(z[pre + "action_mask_sum"])
                assert env.actions == [tuple(int(v) for v in a) for a in z[pre + "actions"]]
                checked_pins += len(got_p)
        env.close()
    assert checked_pins > 200


def _corrupt(cfg, edit):
    from pcbenv import InstanceStream, env_seed
    packed = pack_instances(cfg, [InstanceStream(cfg, env_seed(0, i)).next() for i in range(4)])
    edit(packed, 16 + 8 * cfg.max_num_components)
    return packed


@pytest.mark.parametrize("what", ["nets_beyond_config", "pin_id_beyond_mp", "spatial_id_beyond_pins", "spatial_duplicate_id",
                                  "component_beyond_count", "pin_outside_component", "not_net_major", "pins_on_rect"])
def test_load_instances_rejects_records_that_would_index_out_of_bounds(what):
    """Every index the kernels derive from an instance record is range-checked on the host first (a record read from
    a pcbenv/io.py file is untrusted input)."""
    cfg = {"pin_id_beyond_mp": named_config("c3"), "pins_on_rect": named_config("c2")}.get(what, named_config("c4"))
    mp = cfg.max_num_pins_per_component if what != "pins_on_rect" else 0

    def edit(p, pin0):
        hdr = p[2].view(np.int32)
        if what == "nets_beyond_config":
            hdr[1] = cfg.max_num_nets + 1
        elif what == "pin_id_beyond_mp":
            p[2, pin0 + 4] = mp & 0xFF; p[2, pin0 + 5] = mp >> 8
        elif what == "spatial_id_beyond_pins":
            p[2, pin0 + 4] = int(hdr[2]) & 0xFF
        elif what == "spatial_duplicate_id":
            p[2, pin0 + 4:pin0 + 6] = p[2, pin0 + 8 + 4:pin0 + 8 + 6]
        elif what == "component_beyond_count":
            p[2, pin0 + 3] = int(hdr[0])
        elif what == "pin_outside_component":
            p[2, pin0] = 200
        elif what == "not_net_major":
            p[2, pin0 + 2] = 7
        elif what == "pins_on_rect":
            hdr[2] = 1

    packed = _corrupt(cfg, edit)
    env = BatchedPlacementEnv(cfg, 4, queue_depth=1)
    with pytest.raises(ValueError):
        env.load_packed(packed)
    env.close()
