"""CPU oracle for the Space-Gym step path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package.
The product (space_gym_amd/) never does and fails loudly without its HIP library.
"""
from .pyoracle import Oracle, build, lib_path  # noqa: F401
