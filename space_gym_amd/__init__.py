"""space_gym_amd -- MI355X-native batched step() engine for the Space-Gym environments (MIMUW-RL/space-gym).

Host surface: `make_vec(env_id, num_envs)` -> SpaceGymVectorEnv (gym.vector.VectorEnv-shaped), backed by
hand-written HIP kernels behind the C ABI in include/spacegym.h.  There is no CPU implementation in this package.
"""
from .registration import ENV_SPECS, register_with_gym  # noqa: F401
from .vector_env import SpaceGymVectorEnv, StepInfo, make_vec, make_vec_from_class  # noqa: F401
from .multi_device import MultiDeviceVectorEnv  # noqa: F401

__all__ = ["make_vec", "make_vec_from_class", "SpaceGymVectorEnv", "MultiDeviceVectorEnv", "StepInfo", "ENV_SPECS", "register_with_gym"]
