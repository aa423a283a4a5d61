"""Rule-combinator / dynamic-layer exercise (parity fixture recipe).

Level 0  timers and layer moves: TimedRule + VanishByFilter (the screen of
         bounce_box_contact_prediction.py:164-166), TemporaryRule, DelayedRule + ChangeLayer
         (timing.py, vanish.py:42-61, change_layer.py).
Level 2  first-person view: grid_lines background, FirstPersonAgent renderer and
         KeepNearCenter (re_center.py) snapping everything back by one grid cell, with prey
         appearing on a boundary Mixture and vanishing by a position / velocity filter
         (first_person_predators_prey.py:150-209 without its 102-vertex annulus).
Level 1  sprites appearing and vanishing at run time, as in
         first_person_predators_prey.py:178-209: ConditionalRule(np.random.binomial) around
         CreateSprites (with and without `without_overlapping`), VanishOnContact on a layer
         that grows, a TimedRule purge.
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


def _level0():
    target_factors = distribs.Product(
        [distribs.Continuous('x', 0.2, 0.8), distribs.Continuous('y', 0.3, 0.8),
         distribs.Continuous('x_vel', -0.03, 0.03), distribs.Continuous('y_vel', -0.03, 0.03),
         distribs.Discrete('shape', ['circle', 'triangle', 'square'])],
        scale=0.08, c0=255, c1=64, c2=64)
    target_gen = sprite_generators.generate_sprites(target_factors, num_sprites=3)

    def state_initializer():
        walls = shapes.border_walls(visible_thickness=0.05, c0=128, c1=128, c2=128)
        return collections.OrderedDict([
            ('walls', walls),
            ('targets', target_gen(without_overlapping=walls)),
            ('bin', []),
            ('fixation', [sprite.Sprite(x=0.5, y=0.5, shape='spoke_4', scale=0.05, c0=255, c1=255, c2=255)]),
            ('agent', [sprite.Sprite(x=0.5, y=0.1, shape='circle', scale=0.04, c1=255)]),
            ('screen', [sprite.Sprite(x=0.5, y=0.5, shape='square', c0=96, c1=96, c2=96, opacity=200)]),
        ])

    rules = (
        game_rules.TimedRule(step_interval=(2, 3), rules=(game_rules.VanishByFilter('screen'),)),
        game_rules.TemporaryRule(3, game_rules.VanishByFilter('fixation', lambda _: True)),
        game_rules.DelayedRule(4, game_rules.ChangeLayer('targets', 'bin')),
    )
    physics = physics_lib.Physics(
        (physics_lib.Collision(elasticity=1., symmetric=False, update_angle_vel=False),
         ['targets', 'bin'], 'walls'),
        (physics_lib.Drag(coeff_friction=0.25), 'agent'),
        updates_per_env_step=5)
    return state_initializer, rules, physics, tasks.CompositeTask(timeout_steps=8)


def _level1():
    rng = [-0.05, 1.05]
    boundary = distribs.Mixture([
        distribs.Product([distribs.Continuous('y', *rng)], x=rng[0]),
        distribs.Product([distribs.Continuous('y', *rng)], x=rng[1]),
        distribs.Product([distribs.Continuous('x', *rng)], y=rng[0]),
        distribs.Product([distribs.Continuous('x', *rng)], y=rng[1]),
    ])
    predator_factors = distribs.Product(
        [boundary, distribs.Continuous('x_vel', -0.02, 0.02), distribs.Continuous('y_vel', -0.02, 0.02),
         distribs.Continuous('scale', 0.07, 0.13)],
        shape='circle', c0=0., c1=1., c2=0.8)
    prey_factors = distribs.Product(
        [distribs.Continuous('x', 0.1, 0.9), distribs.Continuous('y', 0.1, 0.9),
         distribs.Discrete('shape', ['square', 'star_5'])],
        scale=0.09, c0=0.2, c1=1., c2=1.)
    predator_gen = sprite_generators.generate_sprites(predator_factors, num_sprites=1)
    prey_gen = sprite_generators.generate_sprites(prey_factors, num_sprites=1)

    def state_initializer():
        walls = shapes.border_walls(visible_thickness=0.02, c0=0., c1=0., c2=0.5)
        agent = sprite.Sprite(x=0.5, y=0.5, shape='circle', scale=0.06, c0=0.33, c1=1., c2=0.66)
        return collections.OrderedDict([
            ('walls', walls), ('prey', []), ('agent', [agent]), ('predators', [])])

    rules = (
        game_rules.ConditionalRule(
            condition=lambda state: np.random.binomial(1, p=0.5),
            rules=game_rules.CreateSprites('predators', predator_gen)),
        game_rules.ConditionalRule(
            condition=lambda state: np.random.binomial(1, p=0.4),
            rules=game_rules.CreateSprites('prey', prey_gen,
                                           without_overlapping=('walls', 'agent', 'prey'))),
        game_rules.VanishOnContact(vanishing_layer='prey', contacting_layer='agent'),
        game_rules.TimedRule((6, 8), (game_rules.VanishByFilter('predators'),)),
    )
    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.25), 'agent'),
        (physics_lib.Collision(elasticity=0.5, symmetric=False), 'agent', 'walls'),
        updates_per_env_step=5)
    task = tasks.CompositeTask(
        tasks.ContactReward(-1., layers_0='agent', layers_1='predators', reset_steps_after_contact=0),
        tasks.ContactReward(1., layers_0='agent', layers_1='prey'),
        timeout_steps=10)
    return state_initializer, rules, physics, task


def _level2():
    grid_size = 0.25
    rng = [-0.3, 1.3]
    boundary = distribs.Mixture([
        distribs.Product([distribs.Continuous('y', *rng)], x=rng[0]),
        distribs.Product([distribs.Continuous('y', *rng)], x=rng[1]),
        distribs.Product([distribs.Continuous('x', *rng)], y=rng[0]),
        distribs.Product([distribs.Continuous('x', *rng)], y=rng[1]),
    ])
    velocity = distribs.SetMinus(
        distribs.Product([distribs.Continuous('x_vel', -0.04, 0.04),
                          distribs.Continuous('y_vel', -0.04, 0.04)]),
        hold_out=distribs.Product([distribs.Continuous('x_vel', -0.02, 0.02),
                                   distribs.Continuous('y_vel', -0.02, 0.02)]))
    prey_factors = distribs.Product(
        [boundary, velocity, distribs.Continuous('scale', 0.07, 0.13)], shape='circle', c0=0.2, c1=1., c2=1.)
    prey_gen = sprite_generators.generate_sprites(prey_factors, num_sprites=1)
    grid = shapes.grid_lines(grid_x=grid_size, grid_y=grid_size, buffer_border=0.5, c0=0., c1=0., c2=0.5)

    def state_initializer():
        agent = sprite.Sprite(x=0.5, y=0.5, shape='circle', scale=0.06, c0=0.33, c1=1., c2=0.66)
        halo = sprite.Sprite(x=0.5, y=0.5, shape='square', scale=0.2, c0=0.6, c1=1., c2=1., opacity=96)
        return collections.OrderedDict([
            ('grid', shapes.grid_lines(grid_x=grid_size, grid_y=grid_size, buffer_border=0.5,
                                       c0=0., c1=0., c2=0.5)),
            ('prey', []), ('agent', [agent]), ('halo', [halo])])
    del grid
    vanish_range = [-0.5, 1.5]

    def _should_vanish(s):
        too_small = (s.position < vanish_range[0]) * (s.velocity < 0.)
        too_large = (s.position > vanish_range[1]) * (s.velocity > 0.)
        return any(too_small) or any(too_large)

    rules = (
        game_rules.ConditionalRule(condition=lambda state: np.random.binomial(1, p=0.3),
                                   rules=game_rules.CreateSprites('prey', prey_gen)),
        game_rules.VanishByFilter('prey', _should_vanish),
        game_rules.KeepNearCenter(agent_layer='agent', layers_to_center=['halo', 'prey'],
                                  grid_x=grid_size),
        game_rules.VanishOnContact(vanishing_layer='prey', contacting_layer='agent'),
    )
    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.25), ['agent', 'halo']), updates_per_env_step=5)
    task = tasks.CompositeTask(
        tasks.ContactReward(lambda _, prey: prey.scale, layers_0='agent', layers_1='prey'),
        timeout_steps=11)
    return state_initializer, rules, physics, task


def get_config(level=0):
    state_initializer, rules, physics, task = (_level0, _level1, _level2)[level]()
    if level == 2:
        return {
            'state_initializer': state_initializer,
            'physics': physics,
            'task': task,
            'action_space': action_spaces.Joystick(
                scaling_factor=0.05, action_layers=('agent', 'halo'), constrained_lr=False),
            'observers': {'image': observers.PILRenderer(
                image_size=(64, 64), color_to_rgb='hsv_to_rgb',
                polygon_modifier=observers.polygon_modifiers.FirstPersonAgent(agent_layer='agent'))},
            'game_rules': rules,
        }
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.Joystick(scaling_factor=0.02, action_layers='agent'),
        'observers': {'image': observers.PILRenderer(image_size=(64, 64),
                                                     color_to_rgb='hsv_to_rgb' if level else None)},
        'game_rules': rules,
    }
