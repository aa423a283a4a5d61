"""Action spaces (reference: moog/action_spaces/__init__.py:3-7): Joystick
(joystick.py:10-77) and Grid (grid.py:8-82).  Parameter records + specs.
"""
import numpy as np

from .. import _dm_env as dm_env


class AbstractActionSpace(object):
    pass


class Joystick(AbstractActionSpace):
    def __init__(self, scaling_factor=1., action_layers='agent', constrained_lr=False,
                 control_velocity=False, momentum=0.):
        self._scaling_factor = scaling_factor
        if not isinstance(action_layers, (list, tuple)):
            action_layers = (action_layers,)
        self._action_layers = action_layers
        self._constrained_lr = constrained_lr
        self._control_velocity = control_velocity
        self._momentum = momentum
        self._action_spec = dm_env.specs.BoundedArray(
            shape=(2,), dtype=np.float32, minimum=-1, maximum=1)

    def random_action(self):
        return np.random.uniform(-1., 1., size=(2,))

    def action_spec(self):
        return self._action_spec


class Grid(AbstractActionSpace):
    _ACTIONS = (np.array([-1, 0]), np.array([1, 0]), np.array([0, -1]), np.array([0, 1]),
                np.array([0, 0]))

    def __init__(self, scaling_factor=1., action_layers='agent', control_velocity=False,
                 momentum=0.):
        self._scaling_factor = scaling_factor
        if not isinstance(action_layers, (list, tuple)):
            action_layers = (action_layers,)
        self._action_layers = action_layers
        self._control_velocity = control_velocity
        self._momentum = momentum
        self._action_spec = dm_env.specs.DiscreteArray(len(self._ACTIONS))

    def random_action(self):
        return np.random.randint(len(Grid._ACTIONS))

    def action_spec(self):
        return self._action_spec
