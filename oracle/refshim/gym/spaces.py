"""Minimal `gym.spaces` stand-in (TEST INFRASTRUCTURE ONLY, see __init__)."""
import numpy as np


class Space(object):
    def contains(self, x):
        raise NotImplementedError

    def __contains__(self, x):
        return self.contains(x)


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)

    def contains(self, x):
        try:
            return int(x) == x and 0 <= int(x) < self.n
        except Exception:
            return False

    def sample(self):
        return int(np.random.randint(self.n))


class Tuple(Space):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def contains(self, x):
        return len(x) == len(self.spaces) and all(
            s.contains(v) for s, v in zip(self.spaces, x)
        )

    def sample(self):
        return tuple(s.sample() for s in self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def __len__(self):
        return len(self.spaces)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

    def contains(self, x):
        x = np.asarray(x)
        return (
            x.shape == self.shape
            and np.can_cast(x.dtype, self.dtype, casting="same_kind")
            and bool(np.all(x >= self.low))
            and bool(np.all(x <= self.high))
        )


class Dict(Space):
    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]

    def contains(self, x):
        return set(x.keys()) == set(self.spaces.keys()) and all(
            self.spaces[k].contains(v) for k, v in x.items()
        )
