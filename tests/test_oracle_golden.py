"""Pins the CPU oracle (oracle/pcbenv_oracle.c) to the reference: every golden
episode recorded from the unmodified reference Python environments must be
reproduced bit for bit (observations, float64 reward, done, info)."""
import numpy as np
import pytest

from golden_util import case_names, load_case, pad_component_grid
from oracle import oracle as orc


def _check_obs(got, want_stack, t, cfg, tag):
    for k, stack in want_stack.items():
        want = stack[t]
        if k == "component_grid":
            want = pad_component_grid(want, cfg.max_num_components)
        assert got[k].shape == want.shape, (tag, k)
        assert np.array_equal(got[k], want), (tag, k, np.argwhere(got[k] != want)[:4])


@pytest.mark.parametrize("name", case_names())
def test_oracle_reproduces_reference_episode(name):
    meta, cfg, eps = load_case(name)
    for e in eps:
        env = orc.OracleEnv(cfg)
        obs = env.reset(instance=e.instance)
        _check_obs(obs, e.obs, 0, cfg, (name, e.seed, e.ep, "reset"))
        for t, act in enumerate(e.actions):
            obs, r, d, info = env.step(act)
            tag = (name, e.seed, e.ep, t, tuple(act))
            _check_obs(obs, e.obs, t + 1, cfg, tag)
            assert np.float64(r).tobytes() == np.float64(e.reward[t]).tobytes(), (tag, r, e.reward[t])
            assert d == bool(e.done[t]), tag
            if np.isnan(e.info[t, 0]):
                assert info == {}, tag
            else:
                assert info["wirelength"] == e.info[t, 0] and info["num_intersections"] == e.info[t, 1], tag


def test_norm2_matches_numpy_blas_fixture():
    """T1: np.linalg.norm of a length-2 float64 vector == sqrt(fma(dy, dy, dx*dx)) (fixture from the build container)."""
    import os
    from golden_util import GOLDEN_DIR
    rows = np.load(os.path.join(GOLDEN_DIR, "norm2.npz"))["rows"]
    L = orc.lib()
    bad = sum(np.float64(L.orc_norm2(dx, dy)).tobytes() != np.float64(w).tobytes() for dx, dy, w in rows)
    assert bad == 0, f"{bad}/{len(rows)} norm mismatches"


def test_set_order_fixture_and_live_cpython():
    """T2: iteration order of set(points) - visited; fixture from the build container and, when this
    interpreter is CPython 3.8-3.11 (same set/tuple-hash implementation), live against real sets."""
    import os
    import sys
    from golden_util import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, "setorder.npz"))
    for x, y, h in z["tuple_hash"]:
        assert orc.lib().orc_tuple_hash2(int(x), int(y)) == int(h)
    for pts, mask, order in zip(z["points"], z["visited_mask"], z["order"]):
        n = int((pts[:, 0] >= 0).sum())
        points = [tuple(int(v) for v in p) for p in pts[:n]]
        want = [points[i] for i in order if i >= 0]
        assert orc.set_difference_order(points, int(mask)) == want
        if sys.implementation.name == "cpython" and (3, 8) <= sys.version_info[:2] <= (3, 11):
            visited = set()
            for i in range(n):
                if mask >> i & 1:
                    visited = visited | {points[i]}
            assert list(set(points) - visited) == want
