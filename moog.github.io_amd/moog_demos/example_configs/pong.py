"""Pong: intercept a falling ball with a left/right paddle.
Parameters: reference moog_demos/example_configs/pong.py:27-128."""
import collections

import numpy as np
from moog import action_spaces, game_rules, observers, physics as physics_lib, sprite, tasks
from moog.state_initialization import distributions as distribs

WALL_HUE = dict(c0=0., c1=0., c2=0.5)


def get_config(_):
    occluder = sprite.Sprite(
        x=0., y=0., scale=1., c0=0.6, c1=1., c2=1.,
        shape=np.array([[-0.1, 0.2], [1.1, 0.2], [1.1, 0.6], [-0.1, 0.6]]))
    ball = distribs.Product(
        [distribs.Continuous('x', 0.1, 0.8), distribs.Continuous('x_vel', -0.03, 0.03)],
        y=1.2, y_vel=-0.02, shape='circle', scale=0.07, c0=0.2, c1=1., c2=1.)
    side_walls = [
        sprite.Sprite(shape=np.array(outline), x=0, y=0, **WALL_HUE)
        for outline in ([[0.05, -0.2], [0.05, 2], [-1, 2], [-1, -0.2]],
                        [[0.95, -0.2], [0.95, 2], [2, 2], [2, -0.2]])]

    def state_initializer():
        paddle = sprite.Sprite(x=0.5, y=0.1, shape='square', aspect_ratio=0.2, scale=0.1,
                               c0=0.33, c1=1., c2=0.66)
        return collections.OrderedDict([
            ('walls', side_walls),
            ('prey', [sprite.Sprite(**ball.sample())]),
            ('agent', [paddle]),
            ('occluder', [occluder]),
        ])

    bounce = physics_lib.Collision(elasticity=1., symmetric=False, update_angle_vel=False)
    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.25), 'agent'),
        (bounce, 'prey', 'walls'),
        updates_per_env_step=10)
    task = tasks.CompositeTask(
        tasks.ContactReward(1., layers_0='agent', layers_1='prey'),
        tasks.Reset(condition=lambda state: all([s.y < 0. for s in state['prey']]),
                    steps_after_condition=15))
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.Joystick(
            scaling_factor=0.005, action_layers='agent', constrained_lr=True),
        'observers': {'image': observers.PILRenderer(
            image_size=(64, 64), anti_aliasing=1, color_to_rgb='hsv_to_rgb')},
        'game_rules': (game_rules.VanishOnContact(vanishing_layer='prey', contacting_layer='agent'),),
    }
