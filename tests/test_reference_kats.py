"""The reference's own KATs (kat_scenarios.py) against the CPU oracle -- and, on
the GPU box, against the HIP path through the C ABI (`-m gpu`)."""
import pytest

import kat_scenarios as K
from oracle import oracle as orc


def _oracle(cfg):
    return orc.OracleEnv(cfg)


def test_square_kats_oracle():
    K.square_scenarios(_oracle)


def test_rect_kats_oracle():
    K.rect_scenarios(_oracle)


def test_pin_kats_oracle():
    K.pin_scenarios(_oracle, "pin")
    K.pin_scenarios(_oracle, "spatial")


def test_reward_chain_kats_oracle():
    K.reward_scenarios(orc)


def _oracle_find_reward(rt):
    """The oracle's stand-alone find_reward on the KAT's pin positions (pinned by reward_scenarios above)."""
    from pcbenv import EnvConfig
    cfg = EnvConfig.pin(10, 10, 1, 1, 2, 4, 2, 4, 5, 2, 4, 4, 2, 2, rt, 2, 0.5)
    return orc.find_reward(cfg, [[(0, 2), (5, 3), (8, 1)], [(2, 0), (3, 4), (4, 1), (7, 5)]])


def test_reward_kats_through_reset_and_step_oracle():
    K.reward_env_scenarios(_oracle, _oracle_find_reward)


def _gpu(cfg):
    from pcbenv.single_env import SingleEnvAdapter
    return SingleEnvAdapter(cfg)


@pytest.mark.gpu
def test_square_kats_hip():
    K.square_scenarios(_gpu)


@pytest.mark.gpu
def test_rect_kats_hip():
    K.rect_scenarios(_gpu)


@pytest.mark.gpu
def test_pin_kats_hip():
    K.pin_scenarios(_gpu, "pin")
    K.pin_scenarios(_gpu, "spatial")


@pytest.mark.gpu
def test_reward_kats_through_reset_and_step_hip():
    """/root/reference/tests/pin_environment/test_env.py:199-391 (find_reward beam / centroid / both, worst case) on
    the HIP path: the hand-built instance through reset(instance) + step, bit-equal to the oracle's find_reward."""
    K.reward_env_scenarios(_gpu, _oracle_find_reward)


@pytest.mark.gpu
def test_single_env_adapter_surface():
    """gym-0.22 surface of the reference envs on the device path: reset()/step(tuple), float64 copies, info dict,
    validate_action, action_mask / grid / components attributes, flat-action wrappers (create_env)."""
    import numpy as np
    from pcbenv import create_env, named_config
    from pcbenv.single_env import SingleEnvAdapter
    env = SingleEnvAdapter(named_config("c4"), seed=3)
    obs = env.reset()
    assert obs["grid"].dtype == np.float64 and obs["action_mask"].shape == (4, 64, 64)
    assert env.validate_action(0, 0, 0) and not env.validate_action(0, 64, 0) and not env.validate_action(-1, 0, 0)
    legal = np.argwhere(env.action_mask == 1)
    obs2, r, d, info = env.step(tuple(int(v) for v in legal[len(legal) // 2]))
    assert r == 0.0 and d is False and info == {} and obs2["grid"].sum() > 0 and obs["grid"].sum() == 0  # fresh copies
    comps = env.components
    assert comps[0].placed and comps[0].position == tuple(int(v) for v in legal[len(legal) // 2][1:]) and not comps[1].placed
    assert all(p.absolute_x >= 0 for p in comps[0].pins) and all(p.absolute_x == -1 for p in comps[1].pins)
    obs3, r, d, info = env.step((0, 0, 0) if env.action_mask[0, 0, 0] == 0 else (0, 63, 63))  # an invalid action
    assert d is True and r == -10.56629126073624 and set(info) == {"wirelength", "num_intersections"}
    env.close()
    ec = {"type": "rectangle", "height": 6, "width": 6, "min_component_w": 2, "max_component_w": 4, "min_component_h": 2,
          "max_component_h": 4, "max_num_components": 4, "min_num_components": 2}
    wenv = create_env(ec)
    o = wenv.reset()
    assert o["action_mask"].shape == (2 * 6 * 6,) and wenv.num_actions == 72
    a = int(np.flatnonzero(o["action_mask"] == 1)[0])
    assert wenv.validate_action(a)
    o, r, d, _ = wenv.step(a)
    assert r == 1.0
