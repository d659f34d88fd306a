"""Loader-only `gym` namespace used by tools/gen_golden.py (generator side, never shipped).

`gym` is not installed in the build container, and the reference's env layer
(`gym_space/envs/*.py`, `gym_space/hexagonal_tiling.py`) does `import gym` at module
scope.  This module provides just enough *namespace* for those imports to succeed:
an empty `Env` base, `spaces.Box/Discrete` containers with `contains`, a no-op
`register`, and `utils.seeding.np_random` returning a NumPy `RandomState`.

Nothing here computes anything on the step path (integration, events, observation,
reward all run the reference's own code with real numpy/scipy).  The one behavioural
piece is `np_random`: real gym hashes the seed before seeding MT19937, so the
seed -> stream mapping differs from a real gym install.  Step parity is unaffected
(inputs are injected); reset parity is distributional only.

It is installed into `sys.modules` by the generator ONLY for the env-layer fixtures;
the `make_step` fixtures are produced in a separate process without it.
"""
import sys
import types

import numpy as np


class Env:
    metadata = {}


class Box:
    def __init__(self, low, high, shape=None, dtype=None):
        self.low = np.asarray(low)
        self.high = np.asarray(high)
        self.dtype = self.low.dtype if dtype is None else np.dtype(dtype)
        self.shape = self.low.shape

    def contains(self, x):
        x = np.asarray(x)
        return bool(x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high))


class Discrete:
    def __init__(self, n):
        self.n = n

    def contains(self, x):
        return 0 <= int(x) < self.n


REGISTRY = {}


def register(id, entry_point=None, max_episode_steps=None, kwargs=None, **_):
    REGISTRY[id] = dict(entry_point=entry_point, max_episode_steps=max_episode_steps, kwargs=kwargs or {})


def np_random(seed=None):
    return np.random.RandomState(seed), seed


def install():
    gym = types.ModuleType("gym")
    gym.Env = Env
    spaces = types.ModuleType("gym.spaces")
    spaces.Box, spaces.Discrete = Box, Discrete
    envs = types.ModuleType("gym.envs")
    registration = types.ModuleType("gym.envs.registration")
    registration.register = register
    envs.registration = registration
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = np_random
    utils.seeding = seeding
    gym.spaces, gym.envs, gym.utils = spaces, envs, utils
    for name, mod in [("gym", gym), ("gym.spaces", spaces), ("gym.envs", envs),
                      ("gym.envs.registration", registration), ("gym.utils", utils),
                      ("gym.utils.seeding", seeding)]:
        sys.modules[name] = mod
    return gym
