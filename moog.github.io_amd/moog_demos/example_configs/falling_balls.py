"""Balls falling under gravity onto walls (multi-collision stress).
Parameters: reference moog_demos/example_configs/falling_balls.py:36-123.
`build()` also makes the scaled 64-sprite variant of SURVEY.md 8(d) config 5."""
import collections

import numpy as np
from moog import action_spaces, observers, physics as physics_lib, sprite as sprite_lib, tasks
from moog.state_initialization import distributions as distribs
from moog.state_initialization import sprite_generators

WALL_OUTLINES = (
    [[-1, 0.1], [2, 0.1], [2, -1], [-1, -1]],            # floor
    [[0.05, -0.1], [0.05, 1.1], [-1, 1.1], [-1, -0.1]],  # left
    [[0.95, -0.1], [0.95, 1.1], [2, 1.1], [2, -0.1]],    # right
    [[0.45, -1], [0.45, 0.3], [0.55, 0.3], [0.55, -1]],  # divider
)


def build(num_balls=4, x_range=(0.25, 0.75), y_range=(0.5, 0.9), scale=0.1):
    ball_factors = distribs.Product(
        [distribs.Continuous('x', *x_range), distribs.Continuous('y', *y_range),
         distribs.Continuous('x_vel', -0.01, 0.01)],
        scale=scale, shape='circle', c0=0, c1=0, c2=255, mass=1.)
    make_balls = sprite_generators.generate_sprites(ball_factors, num_sprites=num_balls)
    walls = [sprite_lib.Sprite(shape=np.array(o), x=0, y=0, c0=128, c1=128, c2=128)
             for o in WALL_OUTLINES]

    def state_initializer():
        return collections.OrderedDict(
            [('walls', walls), ('balls', make_balls(disjoint=True)), ('agent', [])])

    bounce = physics_lib.Collision(
        elasticity=0.6, symmetric=False, update_angle_vel=False, max_recursion_depth=2)
    physics = physics_lib.Physics(
        (bounce, 'balls', ['balls', 'walls']),
        (physics_lib.DownGravity(g=-0.001), 'balls'),
        updates_per_env_step=20)
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': tasks.CompositeTask(timeout_steps=100),
        'action_space': action_spaces.Grid(action_layers='agent'),
        'observers': {'image': observers.PILRenderer(image_size=(64, 64), anti_aliasing=1)},
        'game_rules': (),
    }


def get_config(_):
    return build()
