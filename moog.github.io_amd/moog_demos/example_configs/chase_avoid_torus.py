"""Chase prey / avoid predators on a torus.
Parameters: reference moog_demos/example_configs/chase_avoid_torus.py:28-178."""
import collections

import numpy as np
from moog import action_spaces, game_rules, observers, physics as physics_lib, sprite, tasks
from moog.observers import polygon_modifiers
from moog.state_initialization import distributions as distribs
from moog.state_initialization import sprite_generators


def _mover(**color):
    return distribs.Product(
        [distribs.Continuous('x', 0., 1.), distribs.Continuous('y', 0., 1.),
         distribs.Continuous('x_vel', -0.02, 0.02), distribs.Continuous('y_vel', -0.02, 0.02)],
        scale=0.08, shape='circle', opacity=192, **color)


def _build(num_prey, num_predators):
    agent_factors = distribs.Product(
        [distribs.Continuous('x', 0., 1.), distribs.Continuous('y', 0., 1.)],
        scale=0.08, c0=0, c1=255, c2=0)
    make_predators = sprite_generators.generate_sprites(
        _mover(c0=255, c1=0, c2=0), num_sprites=num_predators)
    make_prey = sprite_generators.generate_sprites(
        _mover(c0=255, c1=255, c2=0), num_sprites=num_prey)

    def state_initializer():
        agent = sprite.Sprite(**agent_factors.sample())
        predators = make_predators(without_overlapping=(agent,))
        prey = make_prey(without_overlapping=(agent,))
        return collections.OrderedDict(
            [('prey', prey), ('predators', predators), ('agent', [agent])])

    chase = physics_lib.DistanceForce(
        physics_lib.linear_force_fn(zero_intercept=-0.001, slope=0.0005))
    flee = physics_lib.DistanceForce(
        physics_lib.linear_force_fn(zero_intercept=0.001, slope=-0.0005))
    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.25), 'agent'),
        (physics_lib.RandomForce(max_force_magnitude=0.01), ['predators', 'prey']),
        (chase, 'agent', 'predators'),
        (flee, 'agent', 'prey'),
        updates_per_env_step=10,
        corrective_physics=[physics_lib.ConstantSpeed(layer_names=['prey', 'predators'],
                                                      speed=0.015)])
    task = tasks.CompositeTask(
        tasks.Reset(condition=lambda state: len(state['prey']) == 0, steps_after_condition=5),
        tasks.ContactReward(-5, layers_0='agent', layers_1='predators', reset_steps_after_contact=0),
        tasks.ContactReward(1, layers_0='agent', layers_1='prey'),
        timeout_steps=300)

    def wrap(s):
        s.position = np.remainder(s.position, 1)

    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.Joystick(
            scaling_factor=0.025, action_layers='agent', control_velocity=True),
        'observers': {'image': observers.PILRenderer(
            image_size=(64, 64), anti_aliasing=1,
            polygon_modifier=polygon_modifiers.TorusGeometry(['agent', 'predators', 'prey']))},
        'game_rules': (
            game_rules.VanishOnContact(vanishing_layer='prey', contacting_layer='agent'),
            game_rules.ModifySprites(('agent', 'predators', 'prey'), wrap)),
    }


def get_config(level):
    if level == 0:
        return _build(num_prey=1, num_predators=2)
    if level == 1:
        return _build(num_prey=lambda: np.random.randint(1, 3),
                      num_predators=lambda: np.random.randint(1, 3))
    raise ValueError('Invalid level {}'.format(level))
