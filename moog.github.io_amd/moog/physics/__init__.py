"""Physics components (reference: moog/physics/__init__.py:3-21).

Parameter records only -- the integrators and the collision response run in the
HIP step kernel (csrc/moog_step.hip).  Constructor signatures follow the
reference: physics.py:15, collisions.py:466-467, friction.py:20,46,
gravity.py:13,36, distance_fn_force.py:17,50-53,77, random_force.py:11,
constant_speed.py:18.
"""
import numpy as np


class AbstractForce(object):
    """abstract_force.py:10-41"""

    def reset(self, state):
        pass


class AbstractNewtonianForce(AbstractForce):
    """abstract_force.py:44-74"""


class AbstractPhysics(object):
    """abstract_physics.py:6-47"""

    def __init__(self, updates_per_env_step=1):
        self._updates_per_env_step = updates_per_env_step

    def reset(self, state):
        pass

    @property
    def updates_per_env_step(self):
        return self._updates_per_env_step


class Drag(AbstractNewtonianForce):
    def __init__(self, coeff_friction=1.):
        self._coeff_friction = coeff_friction


class KineticFriction(AbstractNewtonianForce):
    def __init__(self, coeff_friction=1.):
        self._coeff_friction = coeff_friction


class DownGravity(AbstractNewtonianForce):
    def __init__(self, g=-1.):
        self._g = g


class Gravity(AbstractNewtonianForce):
    def __init__(self, g=-1., symmetric=True):
        self._g = g
        self._symmetric = symmetric


class _ForceFn(object):
    """Declarative distance->magnitude function (callable for API parity)."""

    def __init__(self, kind, **params):
        self.kind = kind
        self.params = params

    def __call__(self, distance):
        p = self.params
        if self.kind == 'linear':
            horizon = -1. * p['zero_intercept'] / p['slope']
            mag = p['zero_intercept'] + p['slope'] * distance
            if not p['apply_distant_force'] and distance > horizon:
                mag = 0
            if not p['apply_nearby_force'] and distance < horizon:
                mag = 0
            return mag
        return -1. * p['spring_constant'] * (distance - p['equilibrium'])


def linear_force_fn(zero_intercept, slope, apply_distant_force=False, apply_nearby_force=True):
    """distance_fn_force.py:50-74"""
    return _ForceFn('linear', zero_intercept=zero_intercept, slope=slope,
                    apply_distant_force=apply_distant_force,
                    apply_nearby_force=apply_nearby_force)


def spring_force_fn(spring_constant, equilibrium=0):
    """distance_fn_force.py:77-89"""
    return _ForceFn('spring', spring_constant=spring_constant, equilibrium=equilibrium)


class DistanceForce(AbstractNewtonianForce):
    """distance_fn_force.py:16-47.  `force_fn` is linear_force_fn(...) / spring_force_fn(...) (two dedicated force kinds) or
    any Python function of the scalar distance: that one is traced once with a symbolic argument (moog/_symbolic.py
    trace_scalar_fn: arithmetic, comparisons, `if` on the distance, np.sqrt / sin / cos / abs / minimum / maximum ...) and
    evaluated by the device's expression evaluator for every pair, in float64 as np.linalg.norm hands the distance over."""

    def __init__(self, force_fn, symmetric=False):
        self._force_node = None
        if not isinstance(force_fn, _ForceFn):
            if not callable(force_fn):
                raise TypeError('DistanceForce force_fn must be callable')
            from .. import _symbolic
            try:
                self._force_node = _symbolic.trace_scalar_fn(force_fn)
            except _symbolic.Unsupported as exc:
                raise NotImplementedError('DistanceForce force_fn is not lowered: %s' % (exc,))
        self._force_fn = force_fn
        self._symmetric = symmetric


class RandomForce(AbstractNewtonianForce):
    def __init__(self, max_force_magnitude):
        self._max_force_magnitude = max_force_magnitude


class Collision(AbstractForce):
    def __init__(self, elasticity=1., symmetric=False, update_angle_vel=True,
                 max_recursion_depth=0):
        self._elasticity = elasticity
        self._symmetric = symmetric
        self._update_angle_vel = update_angle_vel
        self._max_recursion_depth = max_recursion_depth


class ConstantSpeed(AbstractPhysics):
    def __init__(self, layer_names, speed):
        super(ConstantSpeed, self).__init__(1)
        if not isinstance(layer_names, (list, tuple)):
            layer_names = [layer_names]
        self._layer_names = list(layer_names)
        self._speed = speed


class Tether(AbstractPhysics):
    """Rigidly tethers all sprites of the given layers (tether_physics.py:94-136): after
    the forces of every substep their velocities / angular velocities are replaced by the
    rigid-body motion about the centre of mass (or about `anchor`)."""

    def __init__(self, layer_names, update_angle_vel=True, anchor=None):
        super(Tether, self).__init__(1)
        if not isinstance(layer_names, (list, tuple)):
            layer_names = [layer_names]
        self._layer_names = list(layer_names)
        self._update_angle_vel = update_angle_vel
        self._anchor = anchor


class TetherZippedLayers(Tether):
    """Tethers {i-th sprite of every layer} for each i (tether_physics.py:139-201); the
    layers must hold the same number of sprites (ValueError otherwise)."""


class MazePhysics(AbstractPhysics):
    """Constrains the sprites of `avatar_layers` to the corridors of the maze drawn by the wall sprites of
    `maze_layer` (maze_physics.py:18-211); used as the last corrective physics of a Physics."""

    def __init__(self, maze_layer='walls', avatar_layers=(), constant_speed=None, max_speed=None):
        super(MazePhysics, self).__init__(updates_per_env_step=1)
        self._maze_layer = maze_layer
        if not isinstance(avatar_layers, (list, tuple)):
            avatar_layers = [avatar_layers]
        self._avatar_layers = list(avatar_layers)
        self._constant_speed = constant_speed
        self._max_speed = max_speed


class AbstractMazeWalk(AbstractForce):
    """maze_walk.py:17-93"""

    def __init__(self, speed, maze_layer='walls'):
        self._speed = speed
        self._maze_layer = maze_layer


class RandomMazeWalk(AbstractMazeWalk):
    """Sprites walk the maze at constant speed and turn at random at corners and intersections
    (maze_walk.py:96-193)."""

    def __init__(self, speed, maze_layer='walls', prevent_backtracking=True, allow_wall_backtracking=False,
                 only_turn_at_wall=False):
        super(RandomMazeWalk, self).__init__(speed, maze_layer=maze_layer)
        self._prevent_backtracking = prevent_backtracking
        self._allow_wall_backtracking = allow_wall_backtracking
        self._only_turn_at_wall = only_turn_at_wall


class DeterministicMazeWalk(AbstractMazeWalk):
    """Sprites walk the maze along prescribed velocities (maze_walk.py:203-243): each time a sprite of the layer enters an
    intersection or stands still the next element of `step_velocities` is read -- the list is shared by the sprites and is
    never rewound, exactly as in the reference (and, as there, a prescribed velocity whose signs differ from the current
    one's sets the velocity to (-speed, -speed): np.clip(..., -speed, -speed), maze_walk.py:240-243)."""

    def __init__(self, speed, step_velocities, maze_layer='walls'):
        super(DeterministicMazeWalk, self).__init__(speed, maze_layer=maze_layer)
        import numpy as np
        self._step_velocities = [np.array(v) for v in step_velocities]


class Physics(AbstractPhysics):
    def step(self, state):
        """abstract_physics.py:39-42.  The engine steps the physics on the device; the one host-side call that is
        lowered is the look-ahead an initializer runs on its freshly built state (bounce_box_contact_prediction.py:48):
        recorded by the tracer as one env step of this physics inside the reset."""
        from .. import _trace
        t = _trace.active()
        if t is None:
            raise RuntimeError('physics.step(state) runs on the device; on the host it is only traced inside a state_initializer')
        t.sim_physics = self
        t.sim_step()

    def __init__(self, *forces, updates_per_env_step=1, corrective_physics=()):
        super(Physics, self).__init__(updates_per_env_step=updates_per_env_step)
        self._forces = forces
        if not isinstance(corrective_physics, (list, tuple)):
            corrective_physics = [corrective_physics]
        self._corrective_physics = list(corrective_physics)

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

_submodules(abstract_force=('AbstractForce', 'AbstractNewtonianForce'), abstract_physics=('AbstractPhysics',), collisions=('Collision',), constant_speed=('ConstantSpeed',), distance_fn_force=('DistanceForce', 'linear_force_fn', 'spring_force_fn'), friction=('Drag', 'KineticFriction'), gravity=('DownGravity', 'Gravity'), maze_walk=('DeterministicMazeWalk', 'RandomMazeWalk'), maze_physics=('MazePhysics',), physics=('Physics',), random_force=('RandomForce',), tether_physics=('Tether', 'TetherZippedLayers'))
