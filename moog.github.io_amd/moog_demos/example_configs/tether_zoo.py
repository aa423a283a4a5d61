"""Tether / TetherZippedLayers exercise (parity fixture recipe).

Level 0-2 are the scenario of the reference's tests/moog/physics/test_tether_physics.py
(:92-106: three triangles inside border walls, inelastic asymmetric collisions with
angular velocity, K=10) with the three tethers that file pins with known answers:
    0  Tether('sprites', update_angle_vel=True)                      (:175-194)
    1  Tether('sprites', update_angle_vel=False)                     (:154-173)
    2  Tether('sprites', anchor=[0.2, 0.2])                          (:196-215)
made into an environment: a joystick pushes the tethered sprites (an in-place velocity
update, which in level 1 lands on the ONE ndarray the sprites share after
tether_physics.py:90), Drag acts on them, and episodes time out so resets are covered.
Levels 10-12 are the bare known-answer scenarios (no Drag; the joystick is parked on the
infinite-mass walls), stepped with `physics.step` only.
Level 3 / 4 zip two layers pairwise (TetherZippedLayers, :139-201) with sampled float32
velocities, as multi_tracking_with_feature.py:161-162 / match_to_sample.py:155-156 do.
"""
import collections

import numpy as np

from moog import action_spaces
from moog import observers
from moog import physics as physics_lib
from moog import shapes
from moog import sprite
from moog import tasks
from moog.state_initialization import distributions as distribs
from moog.state_initialization import sprite_generators


def _triangles():
    return [
        sprite.Sprite(x=0.5, y=0.7, scale=0.1, shape='triangle', x_vel=0.04, y_vel=-0.02, c0=255,
                      angle=2.),
        sprite.Sprite(x=0.2, y=0.6, scale=0.1, shape='triangle', x_vel=0., y_vel=0., c1=255,
                      angle=1.),
        sprite.Sprite(x=0.6, y=0.3, scale=0.1, shape='triangle', x_vel=0., y_vel=0., c2=255),
    ]


def get_config(level=0):
    collision = physics_lib.Collision(elasticity=0., symmetric=False, update_angle_vel=True)
    if level in (0, 1, 2, 10, 11, 12):
        tether = [physics_lib.Tether('sprites', update_angle_vel=True),
                  physics_lib.Tether('sprites', update_angle_vel=False),
                  physics_lib.Tether('sprites', anchor=np.array([0.2, 0.2]))][level % 10]

        def state_initializer():
            walls = shapes.border_walls(visible_thickness=0.05, c0=128, c1=128, c2=128)
            return collections.OrderedDict([('walls', walls), ('sprites', _triangles())])

        if level >= 10:
            physics = physics_lib.Physics(
                (collision, 'sprites', 'walls'),
                corrective_physics=[tether], updates_per_env_step=10)
            action_layers = 'walls'
        else:
            physics = physics_lib.Physics(
                (collision, 'sprites', 'walls'),
                (physics_lib.Drag(coeff_friction=0.05), 'sprites'),
                corrective_physics=[tether], updates_per_env_step=10)
            action_layers = 'sprites'
    else:
        target_factors = distribs.Product(
            [distribs.Continuous('x', 0.2, 0.8), distribs.Continuous('y', 0.2, 0.8),
             distribs.Continuous('x_vel', -0.03, 0.03), distribs.Continuous('y_vel', -0.03, 0.03)],
            shape='circle', scale=0.07, c0=255, c1=64, c2=64)
        bar_factors = distribs.Product(
            [distribs.Continuous('x', 0.2, 0.8), distribs.Continuous('y', 0.2, 0.8),
             distribs.Continuous('angle', 0., 3.)],
            shape='square', scale=0.1, aspect_ratio=0.3, c0=64, c1=64, c2=255, mass=0.5)
        target_gen = sprite_generators.generate_sprites(target_factors, num_sprites=3)
        bar_gen = sprite_generators.generate_sprites(bar_factors, num_sprites=3)

        def state_initializer():
            walls = shapes.border_walls(visible_thickness=0.05, c0=128, c1=128, c2=128)
            return collections.OrderedDict([
                ('walls', walls), ('targets', target_gen(without_overlapping=walls)),
                ('bars', bar_gen())])

        tether = physics_lib.TetherZippedLayers(('targets', 'bars'), update_angle_vel=(level == 4))
        physics = physics_lib.Physics(
            (physics_lib.Collision(elasticity=1., symmetric=False, update_angle_vel=False),
             'targets', 'walls'),
            corrective_physics=[tether], updates_per_env_step=10)
        action_layers = 'targets'

    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': tasks.CompositeTask(timeout_steps=20),
        'action_space': action_spaces.Joystick(scaling_factor=0.005, action_layers=action_layers),
        'observers': {'image': observers.PILRenderer(image_size=(64, 64))},
    }
