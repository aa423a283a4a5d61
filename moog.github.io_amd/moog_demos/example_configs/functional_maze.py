"""Walled arena with predator, prey, boosters and portals.
Parameters: reference moog_demos/example_configs/functional_maze.py:23-249
(the config-local `Booster` rule is re-stated; the engine lowers it through
game_rules.register_lowering, keyed on the class name and attributes)."""
import collections

import numpy as np
from moog import action_spaces, game_rules, observers, physics as physics_lib, shapes, sprite, tasks
from moog.state_initialization import distributions as distribs
from moog.state_initialization import sprite_generators


class Booster(game_rules.AbstractRule):
    """Agent gets lighter (faster) and brighter for `boost_duration` steps after
    touching a booster (functional_maze.py:23-78)."""

    def __init__(self, mass_multiplier=0.4, c2_multiplier=0.1, boost_duration=60,
                 agent_layer='agent', booster_layer='boosters'):
        self._mass_multiplier = mass_multiplier
        self._c2_multiplier = c2_multiplier
        self.boost_duration = boost_duration
        self._agent_layer = agent_layer
        self._booster_layer = booster_layer

    def step(self, state, meta_state):
        raise NotImplementedError('Booster is lowered to the device (MOOG_RULE_BOOSTER)')


def _spot(shape, c0, c1, c2):
    return distribs.Product(
        [distribs.Continuous('x', 0.1, 0.9), distribs.Continuous('y', 0.1, 0.9)],
        shape=shape, scale=0.1, c0=c0, c1=c1, c2=c2)


def get_config(_, image_size=(64, 64)):
    grey = dict(c0=0., c1=0., c2=0.5)
    portal_look = dict(shape='square', scale=0.1, c0=0., c1=0., c2=0.95)
    portals = [sprite.Sprite(x=0.125, y=0.125, **portal_look),
               sprite.Sprite(x=0.875, y=0.875, **portal_look)]
    block = np.array([[0.2, 0.2], [0.4, 0.2], [0.4, 0.4], [0.2, 0.4]])
    islands = [sprite.Sprite(shape=block + np.array([off]), x=0., y=0., **grey)
               for off in ([0., 0.], [0., 0.4], [0.4, 0.4], [0.4, 0.])]
    walls = shapes.border_walls(visible_thickness=0.05, **grey) + islands

    make_agent = sprite_generators.generate_sprites(_spot('circle', 0.33, 1., 0.7), num_sprites=1)
    make_predators = sprite_generators.generate_sprites(_spot('circle', 0., 1., 0.8), num_sprites=1)
    make_prey = sprite_generators.generate_sprites(
        _spot('circle', 0.2, 1., 1.), num_sprites=lambda: np.random.randint(2, 5))
    make_boosters = sprite_generators.generate_sprites(_spot('triangle', 0.6, 1., 1.), num_sprites=2)

    def state_initializer():
        agent = make_agent(without_overlapping=walls)
        predators = make_predators(without_overlapping=walls + agent)
        boosters = make_boosters(without_overlapping=walls + agent)
        prey = make_prey(without_overlapping=walls)
        return collections.OrderedDict([
            ('walls', walls), ('portals', portals), ('boosters', boosters), ('prey', prey),
            ('predators', predators), ('agent', agent)])

    hunt = physics_lib.DistanceForce(
        force_fn=physics_lib.linear_force_fn(zero_intercept=-0.002, slope=0.001))
    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.25), 'agent'),
        (physics_lib.Drag(coeff_friction=0.05), 'predators'),
        (physics_lib.RandomForce(max_force_magnitude=0.02), 'predators'),
        (physics_lib.Drag(coeff_friction=0.02), 'prey'),
        (physics_lib.RandomForce(max_force_magnitude=0.02), 'prey'),
        (hunt, 'agent', 'predators'),
        (physics_lib.Collision(elasticity=0.25, symmetric=False, update_angle_vel=False),
         ['agent', 'predators', 'prey'], 'walls'),
        updates_per_env_step=5)
    task = tasks.CompositeTask(
        tasks.ContactReward(-5, layers_0='agent', layers_1='predators', reset_steps_after_contact=0),
        tasks.ContactReward(1, layers_0='agent', layers_1='prey'),
        tasks.Reset(condition=lambda state: len(state['prey']) == 0, steps_after_condition=5),
        timeout_steps=400)
    rules = (game_rules.VanishOnContact(vanishing_layer='prey', contacting_layer='agent'),
             game_rules.Portal(teleporting_layer='agent', portal_layer='portals'),
             Booster())
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.Joystick(scaling_factor=0.01, action_layers='agent'),
        'observers': {'image': observers.PILRenderer(
            image_size=image_size, anti_aliasing=1, color_to_rgb='hsv_to_rgb')},
        'game_rules': rules,
    }
