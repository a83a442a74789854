"""Offline stand-in for the `gym` package (TEST INFRASTRUCTURE ONLY).

The reference environments (`/root/reference/environment/*.py`) import `gym`
only for the `gym.Env` base class and for `gym.spaces.*` declarations; gym is
not installed in this image and there is no network.  This ~60-line stand-in
is our own code; it is put first on PYTHONPATH *only* by
`tests/golden/make_golden.py` (run in the build container, where
/root/reference exists) so that the unmodified reference can be imported to
produce golden vectors.  Nothing in the product imports it.
"""
from . import spaces  # noqa: F401


class Env(object):
    metadata: dict = {}
    reward_range = (-float("inf"), float("inf"))
    action_space = None
    observation_space = None

    def reset(self, *a, **k):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.action_space = env.action_space
        self.observation_space = env.observation_space

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self, *a, **k):
        return self.env.reset(*a, **k)

    def step(self, action):
        return self.env.step(action)


class ObservationWrapper(Wrapper):
    def reset(self, *a, **k):
        return self.observation(self.env.reset(*a, **k))

    def step(self, action):
        obs, r, d, i = self.env.step(action)
        return self.observation(obs), r, d, i


class ActionWrapper(Wrapper):
    def step(self, action):
        return self.env.step(self.action(action))
