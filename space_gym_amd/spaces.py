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
