"""Environment wrappers (reference: moog/env_wrappers/__init__.py): the Gym surface
 and the mental-simulation wrapper."""
from . import gym_wrapper  # noqa: F401
from . import simulation  # noqa: F401
from .gym_wrapper import GymWrapper  # noqa: F401
from .simulation import SimulationEnvironment  # noqa: F401
