"""Open arena with predators: red discs hunt the green agent disc (a spring-like attraction plus random kicks), bounce
off the arena's border while the agent merely stops at it; an episode ends the moment the agent is caught (-1).
Across episodes a curriculum object adapts the predators' mass -- lighter, hence faster, after a long episode, heavier
after a short one -- and keeps that number between resets.
Parameters: reference moog_demos/example_configs/predators_arena.py:29-199 (get_config(num_predators)).

What the engine exercises here: an initializer that is a bound method of an object with state that outlives episodes
(the mass lives in a per-env slot that resets never clear, its update runs on the device at every reset but the
env's first), `sprite.mass = ...` on generated sprites, RandomForce, DistanceForce."""
import collections

from moog import action_spaces, observers, physics as physics_lib, shapes, tasks
from moog.state_initialization import distributions as distribs
from moog.state_initialization import sprite_generators


class Curriculum(object):
    """Builds every episode's state and carries the predators' mass from one episode to the next."""

    def __init__(self, num_predators, rate, long_episode):
        self._mass = 1.
        self._rate, self._long_episode = rate, long_episode
        anywhere = [distribs.Continuous('x', 0., 1.), distribs.Continuous('y', 0., 1.)]
        self._make_agent = sprite_generators.generate_sprites(
            distribs.Product(anywhere, shape='circle', scale=0.1, c0=0.33, c1=1., c2=0.66), num_sprites=1)
        self._make_predators = sprite_generators.generate_sprites(
            distribs.Product(anywhere, shape='circle', scale=0.1, c0=0., c1=1., c2=0.8), num_sprites=num_predators)
        self._border = shapes.border_walls(visible_thickness=0., c0=0., c1=0., c2=0.5)
        self._meta_state = None

    def state_initializer(self):
        agent = self._make_agent(without_overlapping=self._border)
        predators = self._make_predators(without_overlapping=self._border + agent)
        if self._meta_state is not None:   # (not before the first episode: the meta-state is made after the state)
            if self._meta_state['step_count'] > self._long_episode:
                self._mass -= self._mass * self._rate
            else:
                self._mass += self._mass * self._rate
        for predator in predators:
            predator.mass = self._mass
        return collections.OrderedDict([('walls', self._border), ('agent', agent), ('predators', predators)])

    def meta_state_initializer(self):
        self._meta_state = {'step_count': 0}
        return self._meta_state


def get_config(num_predators):
    curriculum = Curriculum(num_predators=num_predators, rate=0.1, long_episode=200)
    bounce = physics_lib.Collision(elasticity=1., symmetric=False)
    stop = physics_lib.Collision(elasticity=0., symmetric=False)
    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.25), 'agent'),
        (physics_lib.Drag(coeff_friction=0.04), 'predators'),
        (physics_lib.RandomForce(max_force_magnitude=0.03), 'predators'),
        (physics_lib.DistanceForce(physics_lib.linear_force_fn(zero_intercept=-0.0025, slope=0.0001)), 'agent', 'predators'),
        (bounce, 'predators', 'walls'),
        (stop, 'agent', 'walls'),
        updates_per_env_step=10)
    return {
        'state_initializer': curriculum.state_initializer,
        'physics': physics,
        'task': tasks.ContactReward(-1, layers_0='agent', layers_1='predators', reset_steps_after_contact=0),
        'action_space': action_spaces.Joystick(scaling_factor=0.01, action_layers='agent'),
        'observers': {'image': observers.PILRenderer(image_size=(64, 64), anti_aliasing=1, color_to_rgb='hsv_to_rgb')},
        'meta_state_initializer': curriculum.meta_state_initializer,
    }
