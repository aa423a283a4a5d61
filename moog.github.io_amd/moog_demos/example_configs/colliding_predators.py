"""Avoid bouncing polygonal predators (collision showcase).
Parameters: reference moog_demos/example_configs/colliding_predators.py:23-134.
`build()` also makes the scaled 32-sprite variant of SURVEY.md 8(d) config 3."""
import collections

import numpy as np
from moog import action_spaces, observers, physics as physics_lib, shapes, tasks
from moog.state_initialization import distributions as distribs
from moog.state_initialization import sprite_generators

QUAD = 1.8 * np.array([[-0.3, -0.3], [0.1, -0.7], [0.4, 0.6], [-0.1, 0.25]])
PENT = 1.5 * np.array([[-0.5, -0.3], [-0.1, -0.7], [0.7, 0.1], [0., -0.1], [-0.3, 0.25]])


def build(num_predators=5, predator_xy=(0.2, 0.8), predator_scale=(0.1, 0.15), agent_scale=0.1,
          image_size=(64, 64)):
    agent_factors = distribs.Product(
        [distribs.Continuous('x', 0.1, 0.9), distribs.Continuous('y', 0.1, 0.9)],
        shape='circle', scale=agent_scale, c0=0.33, c1=1., c2=0.66)
    predator_factors = distribs.Product(
        [distribs.Continuous('x', *predator_xy),
         distribs.Continuous('y', *predator_xy),
         distribs.Discrete('shape', [QUAD, PENT, 'star_5', 'triangle', 'spoke_5']),
         distribs.Continuous('angle', 0., 2 * np.pi),
         distribs.Continuous('aspect_ratio', 0.75, 1.25),
         distribs.Continuous('scale', *predator_scale),
         distribs.Continuous('x_vel', -0.03, 0.03),
         distribs.Continuous('y_vel', -0.03, 0.03),
         distribs.Continuous('angle_vel', -0.05, 0.05)],
        c0=0., c1=1., c2=0.8)
    walls = shapes.border_walls(visible_thickness=0.05, c0=0., c1=0., c2=0.5)
    make_agent = sprite_generators.generate_sprites(agent_factors, num_sprites=1)
    make_predators = sprite_generators.generate_sprites(predator_factors, num_sprites=num_predators)

    def state_initializer():
        predators = make_predators(disjoint=True, without_overlapping=walls)
        agent = make_agent(without_overlapping=walls + predators)
        return collections.OrderedDict(
            [('walls', walls), ('predators', predators), ('agent', agent)])

    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.25), 'agent'),
        (physics_lib.Collision(elasticity=1., symmetric=True, update_angle_vel=True),
         'predators', 'predators'),
        (physics_lib.Collision(elasticity=1., symmetric=False, update_angle_vel=True),
         'predators', 'walls'),
        (physics_lib.Collision(elasticity=0., symmetric=False, update_angle_vel=False),
         'agent', 'walls'),
        updates_per_env_step=10)
    task = tasks.CompositeTask(
        tasks.ContactReward(-5, layers_0='agent', layers_1='predators'),
        tasks.StayAlive(reward_period=20, reward_value=0.2),
        timeout_steps=200)
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.Joystick(scaling_factor=0.01, action_layers='agent'),
        'observers': {'image': observers.PILRenderer(
            image_size=image_size, anti_aliasing=1, color_to_rgb='hsv_to_rgb')},
    }


def get_config(_):
    return build()
