"""Action spaces (reference: moog/action_spaces/__init__.py:3-7): Joystick
(joystick.py:10-77), Grid (grid.py:8-82), SetPosition (set_position.py:13-67) and Composite
(composite.py:11-79).  Parameter records + specs.
"""
import collections

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


class SetPosition(AbstractActionSpace):
    """set_position.py:13-67: the action in [0, 1]^2 is where the sprites of action_layers go
    (blended with their position by `inertia`)."""

    def __init__(self, action_layers='agent', inertia=0.):
        if not isinstance(action_layers, (list, tuple)):
            action_layers = (action_layers,)
        self._action_layers = action_layers
        self._inertia = inertia
        self._action_spec = dm_env.specs.BoundedArray(
            shape=(2,), dtype=np.float32, minimum=0, maximum=1)

    def random_action(self):
        return np.random.uniform(0., 1., size=(2,))

    def action_spec(self):
        return self._action_spec


class Composite(AbstractActionSpace):
    """composite.py:11-79: several action spaces side by side (multi-agent tasks, eye + hand);
    an action is a dict with the same keys, applied in keyword order.  Batched actions are a
    dict of tensors (or one f64 tensor [N, n_spaces, 2]; a Grid move goes in component 0)."""

    def __init__(self, **action_spaces):
        self.action_spaces = collections.OrderedDict(action_spaces)
        self._action_keys = list(self.action_spaces.keys())
        self._action_spec = {k: v.action_spec() for k, v in self.action_spaces.items()}

    def random_action(self):
        return {k: v.random_action() for k, v in self.action_spaces.items()}

    def action_spec(self):
        return self._action_spec

    @property
    def action_keys(self):
        return list(self._action_keys)

def _submodules(**modules):
    """The reference keeps one class per file (`from moog.physics import collisions`, `moog.game_rules.vanish.Vanish`);
    here a package is one file, and those module paths are aliases that hold the same objects."""
    import sys
    import types
    for name, names in modules.items():
        m = types.ModuleType(__name__ + '.' + name)
        m.__doc__ = 'Alias module: the reference\'s moog/%s/%s.py (names defined in %s).' % (__name__.split('.')[-1], name, __name__)
        for n in names:
            setattr(m, n, globals()[n])
        sys.modules[m.__name__] = m
        globals()[name] = m

_submodules(abstract_action_space=('AbstractActionSpace',), composite=('Composite',), grid=('Grid',), joystick=('Joystick',), set_position=('SetPosition',))
