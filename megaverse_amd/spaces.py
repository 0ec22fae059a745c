"""Minimal stand-ins for gym.spaces (gym is not a dependency of this package).

If ``gym`` is importable its own classes are used so that RL frameworks see real spaces.
"""
import numpy as np

try:  # pragma: no cover - gym is absent in the build image
    from gym.spaces import Box, Discrete, Tuple  # type: ignore
except Exception:  # noqa: BLE001

    class Discrete:
        def __init__(self, n):
            self.n = int(n)
            self._rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return int(self._rng.integers(self.n))

        def contains(self, x):
            return 0 <= int(x) < self.n

        def __repr__(self):
            return f"Discrete({self.n})"

        def __eq__(self, other):
            return isinstance(other, Discrete) and other.n == self.n

    class Tuple:
        def __init__(self, spaces):
            self.spaces = tuple(spaces)

        def seed(self, seed=None):
            for i, s in enumerate(self.spaces):
                s.seed(None if seed is None else seed + i)

        def sample(self):
            return tuple(s.sample() for s in self.spaces)

        def __len__(self):
            return len(self.spaces)

        def __getitem__(self, i):
            return self.spaces[i]

        def __repr__(self):
            return "Tuple(" + ", ".join(map(repr, self.spaces)) + ")"

    class Box:
        def __init__(self, low, high, shape, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

        def sample(self):
            return np.random.randint(self.low, self.high + 1, self.shape).astype(self.dtype)

        def __repr__(self):
            return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"
