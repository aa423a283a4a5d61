"""Coverage recipe (not a reference task): exercises the hot-path components that
the five BASELINE configs do not reach -- KineticFriction, Gravity, the spring
DistanceForce, Grid velocity control with momentum, a background colour and the
FirstPersonAgent polygon modifier -- so that they are pinned by golden vectors
captured from the reference as well (tests/golden/forces_zoo_s*.npz)."""
import collections

import numpy as np
from moog import action_spaces, observers, physics as physics_lib, shapes, tasks
from moog.observers import polygon_modifiers
from moog.state_initialization import distributions as distribs
from moog.state_initialization import sprite_generators


def get_config(_):
    mover_factors = distribs.Product(
        [distribs.Continuous('x', 0.15, 0.85), distribs.Continuous('y', 0.15, 0.85),
         distribs.Discrete('shape', ['triangle', 'pentagon', 'star_4']),
         distribs.Continuous('angle', 0., 2 * np.pi),
         distribs.Continuous('x_vel', -0.03, 0.03), distribs.Continuous('y_vel', -0.03, 0.03)],
        scale=0.09, c0=0.1, c1=0.8, c2=0.9, mass=1.5)
    planet_factors = distribs.Product(
        [distribs.Continuous('x', 0.2, 0.8), distribs.Continuous('y', 0.2, 0.8)],
        shape='circle', scale=0.08, c0=0.55, c1=1., c2=1., mass=2., opacity=160)
    agent_factors = distribs.Product(
        [distribs.Continuous('x', 0.3, 0.7), distribs.Continuous('y', 0.3, 0.7)],
        shape='square', scale=0.07, c0=0.33, c1=1., c2=0.7)
    walls = shapes.border_walls(visible_thickness=0.04, c0=0., c1=0., c2=0.4)
    make_movers = sprite_generators.generate_sprites(mover_factors, num_sprites=3)
    make_planets = sprite_generators.generate_sprites(planet_factors, num_sprites=2)
    make_agent = sprite_generators.generate_sprites(agent_factors, num_sprites=1)

    def state_initializer():
        movers = make_movers(disjoint=True, without_overlapping=walls)
        planets = make_planets(disjoint=True, without_overlapping=walls + movers)
        agent = make_agent(without_overlapping=walls + movers + planets)
        return collections.OrderedDict(
            [('walls', walls), ('movers', movers), ('planets', planets), ('agent', agent)])

    bounce = physics_lib.Collision(elasticity=0.8, symmetric=False, update_angle_vel=False)
    physics = physics_lib.Physics(
        (physics_lib.KineticFriction(coeff_friction=0.0004), 'movers'),
        (physics_lib.Gravity(g=-0.002, symmetric=True), 'planets', 'planets'),
        (physics_lib.DistanceForce(physics_lib.spring_force_fn(0.004, equilibrium=0.3),
                                   symmetric=True), 'agent', 'planets'),
        (physics_lib.Drag(coeff_friction=0.1), ['agent', 'planets']),
        (bounce, ['movers', 'planets', 'agent'], 'walls'),
        updates_per_env_step=4)
    task = tasks.CompositeTask(
        tasks.ContactReward(2, layers_0='agent', layers_1=['movers', 'planets']),
        tasks.StayAlive(reward_period=7, reward_value=0.5),
        timeout_steps=40)
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.Grid(
            scaling_factor=0.02, action_layers='agent', control_velocity=True, momentum=0.6),
        'observers': {'image': observers.PILRenderer(
            image_size=(64, 64), anti_aliasing=1, color_to_rgb='hsv_to_rgb', bg_color=(20, 30, 40),
            polygon_modifier=polygon_modifiers.FirstPersonAgent('agent'))},
    }
