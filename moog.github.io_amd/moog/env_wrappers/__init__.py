"""Environment wrappers (reference: moog/env_wrappers/__init__.py): the Gym surface
(SURVEY 2 row 15), the mental-simulation wrapper and the episode logger."""
from . import abstract_wrapper  # noqa: F401
from . import gym_wrapper  # noqa: F401
from . import multi_agent  # noqa: F401
from . import logger  # noqa: F401
from . import simulation  # noqa: F401
from .abstract_wrapper import AbstractEnvironmentWrapper  # noqa: F401
from .gym_wrapper import GymWrapper  # noqa: F401
from .multi_agent import MultiAgentEnvironment  # noqa: F401
from .logger import LoggingEnvironment  # noqa: F401
from .simulation import SimulationEnvironment  # noqa: F401
