"""First-person predators / prey (recipe restating the parameters of the reference's
moog_demos/example_configs/first_person_predators_prey.py; the reference file itself also
loads unchanged, see tests/test_host.py).

The agent stays at the centre of a first-person view (FirstPersonAgent renderer, :150-156)
over a background grid (:109-111) and drags a 102-vertex annulus with it (:79-82).
Predators and prey appear at random on a square boundary (Mixture, :38-51) with velocities
from a square annulus (SetMinus, :54-70), one Bernoulli draw per step each (:178-191),
vanish once they have drifted out of range (:193-201), and KeepNearCenter snaps everything
back by one grid cell (:203-207).  Touching a predator costs 2 x its scale and ends the
episode, catching a prey pays its scale (:129-147).
"""
import collections

import numpy as np

from moog import action_spaces
from moog import game_rules
from moog import observers
from moog import physics as physics_lib
from moog import shapes
from moog import sprite
from moog import tasks
from moog.state_initialization import distributions as distribs
from moog.state_initialization import sprite_generators

_FIELD_BUFFER = 0.7
_VANISH_DIST = 1.2
_GRID_SIZE = 0.4

# run-time sprite capacity of the layers the rules append to (the reference's lists are
# unbounded; see BatchedEnvironment(layer_capacity=...))
LAYER_CAPACITY = {'prey': 16, 'predators': 40}


def _boundary_positions(buf):
    rng = [-1. * buf, 1. + buf]
    return distribs.Mixture([
        distribs.Product([distribs.Continuous('y', *rng)], x=rng[0]),
        distribs.Product([distribs.Continuous('y', *rng)], x=rng[1]),
        distribs.Product([distribs.Continuous('x', *rng)], y=rng[0]),
        distribs.Product([distribs.Continuous('x', *rng)], y=rng[1]),
    ])


def _annulus_velocity(min_vel, max_vel):
    return distribs.SetMinus(
        distribs.Product([distribs.Continuous('x_vel', -1. * max_vel, max_vel),
                          distribs.Continuous('y_vel', -1. * max_vel, max_vel)]),
        hold_out=distribs.Product([distribs.Continuous('x_vel', -1. * min_vel, min_vel),
                                   distribs.Continuous('y_vel', -1. * min_vel, min_vel)]))


def get_config(_=0):
    agent = sprite.Sprite(x=0.5, y=0.5, shape='circle', scale=0.04, c0=0.33, c1=1., c2=0.66)
    agent_annulus = sprite.Sprite(
        x=0.5, y=0.5, shape=shapes.annulus_vertices(inner_radius=0.08, outer_radius=0.3), scale=1.,
        c0=0.6, c1=1., c2=1.)
    max_predator_vel, max_prey_vel = 0.02, 0.01
    predator_factors = distribs.Product(
        [_boundary_positions(_FIELD_BUFFER), _annulus_velocity(0.5 * max_predator_vel, max_predator_vel),
         distribs.Continuous('scale', 0.07, 0.13)], shape='circle', c0=0., c1=1., c2=0.8)
    prey_factors = distribs.Product(
        [_boundary_positions(_FIELD_BUFFER), _annulus_velocity(0.5 * max_prey_vel, max_prey_vel),
         distribs.Continuous('scale', 0.07, 0.13)], shape='circle', c0=0.2, c1=1., c2=1.)
    grid = shapes.grid_lines(grid_x=_GRID_SIZE, grid_y=_GRID_SIZE, buffer_border=1., c0=0., c1=0., c2=0.5)

    def state_initializer():
        return collections.OrderedDict([
            ('grid', grid), ('prey', []), ('agent', [agent]), ('predators', []),
            ('agent_annulus', [agent_annulus])])

    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.25), ['agent', 'agent_annulus']), updates_per_env_step=10)
    task = tasks.CompositeTask(
        tasks.ContactReward(reward_fn=lambda _, predator: -2. * predator.scale, layers_0='agent',
                            layers_1='predators', reset_steps_after_contact=0),
        tasks.ContactReward(reward_fn=lambda _, prey: prey.scale, layers_0='agent', layers_1='prey'))
    action_space = action_spaces.Joystick(
        scaling_factor=0.003, action_layers=('agent', 'agent_annulus'), constrained_lr=False)
    observer = observers.PILRenderer(
        image_size=(64, 64), anti_aliasing=1, color_to_rgb='hsv_to_rgb',
        polygon_modifier=observers.polygon_modifiers.FirstPersonAgent(agent_layer='agent'))

    predator_gen = sprite_generators.generate_sprites(predator_factors, num_sprites=1)
    prey_gen = sprite_generators.generate_sprites(prey_factors, num_sprites=1)
    vanish_range = [-1. * _VANISH_DIST, 1. + _VANISH_DIST]

    def _should_vanish(s):
        pos_too_small = (s.position < vanish_range[0]) * (s.velocity < 0.)
        pos_too_large = (s.position > vanish_range[1]) * (s.velocity > 0.)
        return any(pos_too_small) or any(pos_too_large)

    rules = (
        game_rules.ConditionalRule(condition=lambda state: np.random.binomial(1, p=0.5),
                                   rules=game_rules.CreateSprites('predators', predator_gen)),
        game_rules.ConditionalRule(condition=lambda state: np.random.binomial(1, p=0.2),
                                   rules=game_rules.CreateSprites('prey', prey_gen)),
        game_rules.VanishByFilter('prey', _should_vanish),
        game_rules.VanishByFilter('predators', _should_vanish),
        game_rules.KeepNearCenter(agent_layer='agent',
                                  layers_to_center=['agent_annulus', 'predators', 'prey'],
                                  grid_x=_GRID_SIZE),
        game_rules.VanishOnContact(vanishing_layer='prey', contacting_layer='agent'),
    )
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_space,
        'observers': {'image': observer},
        'game_rules': rules,
    }
