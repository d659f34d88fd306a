"""Minimal Box space used when neither gym nor gymnasium is importable (the reference builds its spaces with
gym.spaces.Box: spaceship_env.py:102-111,206-208; kepler.py:158-170)."""
import numpy as np


class Box:
    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)

    def contains(self, x):
        x = np.asarray(x)
        return bool(x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high))

    def sample(self, rng=None):
        rng = rng or np.random.default_rng()
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return rng.uniform(lo, hi).astype(self.dtype)

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    def __eq__(self, other):
        return isinstance(other, Box) and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high)


def batch_box(space, n):
    return Box(np.broadcast_to(space.low, (n,) + space.shape).copy(), np.broadcast_to(space.high, (n,) + space.shape).copy(),
               dtype=space.dtype)


class Discrete:
    """gym.spaces.Discrete(n) stand-in (DiscreteSpaceshipEnv._init_action_space, spaceship_env.py:184-187)."""

    def __init__(self, n):
        self.n, self.shape, self.dtype = int(n), (), np.dtype(np.int64)

    def contains(self, x):
        return np.ndim(x) == 0 and 0 <= int(x) < self.n

    def sample(self, rng=None):
        return int((rng or np.random.default_rng()).integers(self.n))

    def __repr__(self):
        return f"Discrete({self.n})"

    def __eq__(self, other):
        return isinstance(other, Discrete) and other.n == self.n


class MultiDiscrete:
    """Batched Discrete: what gym.vector gives for the action space of n Discrete(k) envs."""

    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape, self.dtype = self.nvec.shape, np.dtype(np.int64)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= 0) and np.all(x < self.nvec))

    def sample(self, rng=None):
        return (rng or np.random.default_rng()).integers(self.nvec)

    def __repr__(self):
        return f"MultiDiscrete({self.nvec[0] if self.nvec.size else 0} x {self.nvec.size})"
