"""The reference's own known-answer tests for the path, restated against the
reset(instance)/step(action) surface shared by the CPU oracle (`OracleEnv`) and
the HIP path (`SingleEnvAdapter` over the C ABI).  Each scenario cites the
reference test it restates (paths under /root/reference/tests/).

The reference tests poke internals (`env.components = [...]`,
`env.update_action_mask(x, y)`); here the same situations are reached through the
public surface: hand-built instances via `reset(instance)`, mask edits via
`step()`.
"""
import math

import numpy as np

from pcbenv import EnvConfig, Instance


def inst(comps, pins=(), num_nets=0):
    """comps: [(h, w)], pins: [(rel_x, rel_y, net, comp, pin_id)] in self.pins (net-major) order."""
    p = np.asarray(pins, np.int64).reshape(-1, 5)
    return Instance(np.asarray([c[0] for c in comps], np.int64), np.asarray([c[1] for c in comps], np.int64),
                    num_nets, p[:, 0], p[:, 1], p[:, 2], p[:, 3], p[:, 4])


# ---- square_environment/test_env.py ---------------------------------------------------------------------
def square_scenarios(make):
    A = np.array
    # :6-26 test_compute_if_done, :29-43 test_update_grid, :224-249 step sequences, :252-258 reset sums
    env = make(EnvConfig.square(4, 4, 2))
    obs = env.reset()
    assert obs["grid"].sum() == 0 and obs["action_mask"].sum() == 3 * 3
    obs, r, d, _ = env.step((0, 0))
    assert r == 1.0 and not d
    assert (obs["grid"] == A([[1, 1, 0, 0], [1, 1, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]])).all()
    assert obs["action_mask"][1, 1] == 0          # :46-55 overlap
    assert obs["action_mask"][2, 2] == 1          # :71-80 correct
    obs, r, d, info = env.step((4, 4))            # :58-68 out of bounds -> invalid -> terminal, reward 0
    assert r == 0.0 and d and info == {}
    env = make(EnvConfig.square(4, 4, 2))
    env.reset()
    env.step((0, 0)); env.step((0, 2))
    obs, r, d, _ = env.step((2, 1))
    assert d                                      # :6-26: no room left
    env = make(EnvConfig.square(4, 4, 2))
    env.reset()
    for a in [(0, 0), (0, 2), (2, 0), (2, 2)]:
        obs, r, d, _ = env.step(a)
    assert d and obs["grid"].sum() == 16 and obs["action_mask"].sum() == 0
    obs, r, d, _ = env.step((0, 0))               # :239-249 an extra action after the end raises nothing
    assert r == 0.0 and d
    # :80-148 masks on 5x5, n = 2
    before = A([[1, 1, 1, 1, 0]] * 4 + [[0] * 5])
    cases5 = {(0, 0): A([[0, 0, 1, 1, 0], [0, 0, 1, 1, 0], [1, 1, 1, 1, 0], [1, 1, 1, 1, 0], [0] * 5]),
              (1, 0): A([[0, 0, 1, 1, 0], [0, 0, 1, 1, 0], [0, 0, 1, 1, 0], [1, 1, 1, 1, 0], [0] * 5]),
              (0, 2): A([[1, 0, 0, 0, 0], [1, 0, 0, 0, 0], [1, 1, 1, 1, 0], [1, 1, 1, 1, 0], [0] * 5])}
    for act, want in cases5.items():
        env = make(EnvConfig.square(5, 5, 2))
        assert (env.reset()["action_mask"] == before).all()
        assert (env.step(act)[0]["action_mask"] == want).all(), act
    # :151-221 masks on 11x10, n = 3
    before = A([[1] * 8 + [0, 0]] * 9 + [[0] * 10] * 2)
    want11 = {(1, 1): A([[0, 0, 0, 0, 1, 1, 1, 1, 0, 0]] * 4 + [[1] * 8 + [0, 0]] * 5 + [[0] * 10] * 2),
              (3, 3): A([[1] * 8 + [0, 0]] + [[1, 0, 0, 0, 0, 0, 1, 1, 0, 0]] * 5 + [[1] * 8 + [0, 0]] * 3 + [[0] * 10] * 2)}
    for act, want in want11.items():
        env = make(EnvConfig.square(11, 10, 3))
        assert (env.reset()["action_mask"] == before).all()
        assert (env.step(act)[0]["action_mask"] == want).all(), act


# ---- rectangular_environment/test_env.py, test_components.py -----------------------------------------------
def rect_scenarios(make):
    cfg = EnvConfig.rect(6, 6, 2, 4, 2, 4, 4, 1)
    two = inst([(1, 2), (3, 2)])
    env = make(cfg)
    obs = env.reset(instance=two)
    m = obs["action_mask"]                        # test_env.py:182-201 test_validate_action
    assert m[0, 0, 0] == 1 and m[0, 4, 5] == 0 and m[1, 2, 3] == 1 and m[1, 5, 4] == 0
    # test_components.py:27-50: [h, w, -1, -1, area ratio]
    assert list(obs["all_components_feature"][0]) == [1, 2, -1, -1, 2 / 36]
    assert list(obs["all_components_feature"][1]) == [3, 2, -1, -1, 6 / 36]
    assert not obs["all_components_feature"][2:].any()
    obs, r, d, info = env.step((0, 0, 0))         # :281-311 test_step
    assert obs["grid"][:1, :2].all() and obs["grid"].sum() == 2
    assert list(obs["placement_mask"]) == [1, 0, 0, 0] and list(obs["component_mask"]) == [1, 1, 0, 0]
    assert list(obs["all_components_feature"][0]) == [1, 2, 0, 0, 2 / 36]
    assert r == 1.0 and not d and info == {}
    # the mask now belongs to the 3x2 component: rows_cols_to_mask(o=0) -> 2 rows, 1 column (:257-278)
    assert not obs["action_mask"][0, 4:, :].any() and not obs["action_mask"][0, :, 5:].any()
    assert obs["action_mask"][0, 3, 4] == 1
    # :204-228 test_compute_action_mask: after (0,0,0),(0,2,3) a 2x2 cannot go at (2,3) but can at (4,0)
    env = make(cfg)
    env.reset(instance=inst([(1, 2), (3, 2), (2, 2)]))
    env.step((0, 0, 0))
    obs, r, d, _ = env.step((0, 2, 3))
    assert obs["action_mask"][0, 2, 3] == 0 and obs["action_mask"][0, 4, 0] == 1 and not d
    # :231-254 orientation 1 of a 4x2 after (0,0,0),(0,1,2)
    env = make(cfg)
    env.reset(instance=inst([(1, 2), (3, 2), (4, 2)]))
    env.step((0, 0, 0))
    obs, _, _, _ = env.step((0, 1, 2))
    assert obs["action_mask"][1, 1, 4] == 0 and obs["action_mask"][1, 4, 1] == 1
    # invalid action: reward 0, done, nothing raised (dummy_env_rectangular.py:424-432)
    obs2, r, d, info = env.step((0, 1, 2))
    assert r == 0.0 and d and info == {} and (obs2["grid"] == obs["grid"]).all()


# ---- pin_environment/test_component.py, test_env.py ----------------------------------------------------------
def pin_scenarios(make, kind="pin"):
    mk = EnvConfig.pin if kind == "pin" else EnvConfig.spatial
    # test_component.py:1-34 pin rotation for all four orientations: Component(4, 3) with pins (0,0), (0,2)
    want = {0: [(0, 0), (0, 2)], 1: [(0, 3), (2, 3)], 2: [(3, 2), (3, 0)], 3: [(2, 0), (0, 0)]}
    for o, rel in want.items():
        env = make(mk(10, 10, 1, 1, 2, 4, 2, 4, 4, 2, 4, 4, 2, 2, "centroid", 2, 0.5))
        env.reset(instance=inst([(4, 3), (2, 2)], [(0, 0, 0, 0, 0), (0, 2, 0, 0, 1)], 1))
        obs, r, d, _ = env.step((o, 1, 2))
        f = obs["all_pins_num_feature"].reshape(-1, 4)
        rows = [0, 1] if kind == "spatial" else [0, 1]  # spatial: global ids 0, 1; pin: [comp 0, pin_id 0 / 1]
        got = [tuple(f[i]) for i in rows]
        assert got == [(rx, ry, 1 + rx, 2 + ry) for rx, ry in rel], (o, got)
    if kind == "pin":
        # test_env.py:782-828 test_update_all_pins_feature (30x30, hand-built, one rotated placement)
        cfg = EnvConfig.pin(30, 30, 1, 1, 2, 5, 2, 5, 6, 1, 2, 4, 5, 2)
        env = make(cfg)
        pins = [(0, 0, 0, 0, 0), (2, 2, 0, 2, 1), (0, 2, 1, 0, 1), (3, 1, 1, 1, 0), (1, 0, 2, 1, 1), (2, 0, 2, 2, 0)]
        obs = env.reset(instance=inst([(1, 3), (4, 2), (5, 5)], pins, 3))
        assert not obs["all_pins_num_feature"][3:].any() and not obs["all_pins_cat_feature"][3:].any()
        obs, _, _, _ = env.step((0, 28, 26))
        n, c = obs["all_pins_num_feature"], obs["all_pins_cat_feature"]
        assert list(n[0, 0]) == [0, 0, 28, 26] and c[0, 0, 0] == 0
        assert list(n[0, 1]) == [0, 2, 28, 28] and c[0, 1, 0] == 1
        assert not n[0, 2:].any() and not c[0, 2:].any()
        obs, _, _, _ = env.step((1, 0, 0))
        n, c = obs["all_pins_num_feature"], obs["all_pins_cat_feature"]
        assert list(n[1, 0]) == [1, 0, 1, 0] and c[1, 0, 0] == 1
        assert list(n[1, 1]) == [0, 2, 0, 2] and c[1, 1, 0] == 2
        assert list(n[2, 0]) == [2, 0, -1, -1] and c[2, 0, 0] == 2
        assert list(n[2, 1]) == [2, 2, -1, -1] and c[2, 1, 0] == 0
        assert not n[2, 2:].any()
        # test_env.py:626-658 test_step: placement_mask [2, 3, 0, 0], reward 0, not done, info {}
        env = make(EnvConfig.pin(6, 6, 1, 1, 2, 4, 2, 4, 4, 2, 4, 4, 2))
        env.reset(instance=inst([(2, 2), (3, 3)], [(0, 0, 0, 0, 0), (0, 1, 0, 1, 1)], 1))
        obs, r, d, info = env.step((0, 0, 0))
        assert obs["grid"][:2, :2].all() and list(obs["placement_mask"]) == [2, 3, 0, 0]
        assert list(obs["all_components_feature"][0]) == [2, 2, 0, 0, 4 / 36] and r == 0 and not d and info == {}
        # :759-779 test_compute_if_done: after (0,0,0), (2,3,0)
        obs, r, d, info = env.step((2, 3, 0))
        assert d and set(info) == {"wirelength", "num_intersections"}


def reward_scenarios(orc):
    """pin_environment/test_env.py:40-391 on the oracle's stand-alone routing functions."""
    assert not orc.is_intersect(((1, 1), (3, 3)), ((1, 3), (1, 5)))            # :40-44
    assert orc.is_intersect(((1, 1), (3, 3)), ((1, 3), (2, 1)))                # :47-51
    r4 = [[((1, 1), (3, 3))], [((2, 1), (0, 3))], [((2, 3), (0, 1))], [((3, 2), (1, 3))]]
    assert orc.find_num_intersection(r4) == 4                                  # :54-68
    assert orc.find_num_intersection([[((4, 4), (3, 5))], [((3, 4), (4, 5))]]) == 1  # :71-85 -> (1, 1)
    cfg48 = EnvConfig.pin(6, 6, 1, 1, 2, 4, 2, 4, 4, 1, 2, 3, 4)
    assert cfg48.max_num_intersections == 48                                    # :88-93
    assert orc.lib().orc_upper_bound_intersections(orc.make_config(cfg48)) == 48
    cfg66 = EnvConfig.pin(6, 6, 1, 1, 2, 4, 2, 4, 4, 2, 4, 4, 2)
    assert np.isclose(cfg66.max_wirelength, 0.5 * 8 * math.sqrt(72))            # :185-191
    assert orc.lib().orc_upper_bound_wirelength(orc.make_config(cfg66)) == cfg66.max_wirelength
    assert orc.route([[(0, 0), (0, 1)], [(2, 2), (3, 3), (4, 4)]], "centroid") == [
        [((0, 0), (0, 1))], [((2, 2), (3.0, 3.0)), ((3, 3), (3.0, 3.0)), ((4, 4), (3.0, 3.0))]]  # :104-122
    pts = np.ascontiguousarray([0, 0, 0, 1, 1, 0, 3, 3], np.intc)
    assert orc.lib().orc_pin_outlier(orc._ip(pts), 4) == 3                      # :125-131
    assert orc.beam_search((0, 0), [(2, 2), (0, 1), (1, 0), (1, 1)], 4) == [(0, 0), (0, 1), (1, 0), (1, 1), (2, 2)]  # :134-141
    # :144-151 passes a set literal: the list below is that set's CPython insertion order
    assert orc.beam_search((0, 0), [(2, 2), (0, 1), (1, 0), (1, 1)], 2) == [(0, 0), (0, 1), (1, 1), (1, 0), (2, 2)]
    assert orc.route([[(3, 3), (3, 4)], [(0, 0), (0, 1), (1, 0), (1, 1), (2, 2)]], "beam", 2) == [
        [((3, 3), (3, 4))], [((2, 2), (1, 1)), ((1, 1), (0, 1)), ((0, 1), (0, 0)), ((0, 0), (1, 0))]]  # :154-173
    assert np.isclose(orc.find_wirelength([[((3, 1), (2, 2))], [((1, 2), (2, 2))], [((3, 3), (2, 2))]]), 1 + 2 * np.sqrt(2))  # :176-182
    assert orc.lib().orc_euclidean_distance(0, 0, 1, 1) == math.sqrt(2)         # :194-196
    # :199-379 find_reward on the hand-built 5-component instance (nets 1 and 2, defaultdict insertion order)
    nets = [[(0, 2), (5, 3), (8, 1)], [(2, 0), (3, 4), (4, 1), (7, 5)]]
    cfg = EnvConfig.pin(10, 10, 1, 1, 2, 4, 2, 4, 5, 2, 4, 4, 2)
    norm = min(3 * 3 * 3.5, 2 * 4)
    wl_beam = (np.sqrt(26) + np.sqrt(13) + np.sqrt(17) + np.sqrt(10) + np.sqrt(5)) / 20
    wl_cen = (13 / 3 + np.sqrt(13) / 3 + np.sqrt(130) / 3 + np.sqrt(41) / 2 + 3 / 2 + np.sqrt(61) / 2 + np.sqrt(13) / 2) / 20
    for rt, wl, ni in (("beam", wl_beam, 1), ("centroid", wl_cen, 2), ("both", wl_beam, 1)):
        cfg.reward_type = rt
        r, w, n = orc.find_reward(cfg, nets)
        assert np.isclose(r, -0.5 * (wl + ni / norm)) and np.isclose(w, wl) and np.isclose(n, ni / norm), rt
    cfg.reward_type = "both"
    r, _, _ = orc.find_reward(cfg, nets, placed_all=False)                      # :382-391
    assert np.isclose(r, -0.5 * 2 * math.sqrt(2) - 0.5 * 24 / 8)


# ---- the same reward KATs through reset(instance) / step(action) --------------------------------------------
def reward_instance(kind="pin"):
    """The hand-built instance of pin_environment/test_env.py:199-379 (conftest.py:68-117, :266-283) as something
    `reset(instance)` + five `step`s reach: the reference test sets `absolute_x/y` of seven pins by hand; here five
    components (3x3, 3x3, 2x1, 2x1, 2x2) are placed unrotated at (0,0), (3,2), (4,1), (7,5), (8,0) so that the pins
    land on exactly those cells, in the same net order (reference nets 1, 2 -> nets 0, 1).  One relative coordinate
    differs from the fixture's: the second pin of component 2 is (2,1) instead of (2,0), because the fixture's two
    hand-set absolute positions of that component imply two different component positions."""
    comps = [(3, 3), (3, 3), (2, 1), (2, 1), (2, 2)]
    #         rel_x rel_y net comp
    pins = [(0, 2, 0, 0), (2, 1, 0, 1), (0, 1, 0, 4),              # net 0: (0,2) (5,3) (8,1)
            (2, 0, 1, 0), (0, 2, 1, 1), (0, 0, 1, 2), (0, 0, 1, 3)]  # net 1: (2,0) (3,4) (4,1) (7,5)
    ids = list(range(7)) if kind == "spatial" else [0] * 7          # the fixture gives every pin pin_id 0 (pin env)
    placements = [(0, 0, 0), (0, 3, 2), (0, 4, 1), (0, 7, 5), (0, 8, 0)]
    return inst(comps, [p + (i,) for p, i in zip(pins, ids)], 2), placements


def reward_env_scenarios(make, reference_values=None):
    """pin_environment/test_env.py:199-391: find_reward for beam / centroid / both and the worst case, reached through
    the public surface.  `reference_values(rt) -> (reward, wl, ni)`: optional exact (bit-level) expectation."""
    norm = min(3 * 3 * 3.5, 2 * 4)
    wl_beam = (np.sqrt(26) + np.sqrt(13) + np.sqrt(17) + np.sqrt(10) + np.sqrt(5)) / 20
    wl_cen = (13 / 3 + np.sqrt(13) / 3 + np.sqrt(130) / 3 + np.sqrt(41) / 2 + 3 / 2 + np.sqrt(61) / 2 + np.sqrt(13) / 2) / 20
    instance, placements = reward_instance("pin")
    for rt, wl, ni in (("beam", wl_beam, 1), ("centroid", wl_cen, 2), ("both", wl_beam, 1)):
        env = make(EnvConfig.pin(10, 10, 1, 1, 2, 4, 2, 4, 5, 2, 4, 4, 2, 2, rt, 2, 0.5))
        env.reset(instance=instance)
        for k, a in enumerate(placements):
            obs, r, d, info = env.step(a)
            assert d == (k == 4) and (r == 0.0 or k == 4), (rt, k)
        assert np.isclose(r, -0.5 * (wl + ni / norm)), (rt, r)                     # :199-379
        assert np.isclose(info["wirelength"], wl) and np.isclose(info["num_intersections"], ni / norm), (rt, info)
        if reference_values is not None:
            want = reference_values(rt)
            got = (r, info["wirelength"], info["num_intersections"])
            assert np.array_equal(np.array(got).view(np.uint64), np.array(want).view(np.uint64)), (rt, got, want)
    env = make(EnvConfig.pin(10, 10, 1, 1, 2, 4, 2, 4, 5, 2, 4, 4, 2, 2, "both", 2, 0.5))
    env.reset(instance=instance)
    obs, r, d, info = env.step((0, 9, 9))                                             # :382-391: not all placed
    assert d and np.isclose(r, -0.5 * 2 * math.sqrt(2) - 0.5 * 24 / 8)
    assert info == {"wirelength": 0.5 * math.sqrt(200) * 8, "num_intersections": 24.0}  # the raw upper bounds (P:907-908)
