"""Coverage recipe (not a reference task): `DependentDistribution` (reference moog/state_initialization/
distributions.py:420-475) -- factors that are a deterministic function of other, sampled factors -- inside a
rejection-sampling generator and in a direct `Sprite(**dist.sample())`.  Pinned by golden vectors captured from the
reference (tests/golden/dependent_zoo_*.npz).

The dependent factors are computed from float32 samples with Python scalars, so they are float32 values in the reference
(NEP 50); a velocity whose two components are such values is a float32 array there, which the fixtures check."""
import collections

import numpy as np
from moog import action_spaces, observers, physics as physics_lib, shapes, sprite, tasks
from moog.state_initialization import distributions as distribs
from moog.state_initialization import sprite_generators


def get_config(level):
    del level
    mirrored = distribs.DependentDistribution(
        distribs.Product([distribs.Continuous('x', 0.2, 0.8), distribs.Continuous('x_vel', -0.02, 0.02)]),
        dependent_fn=lambda s: {'y': 1. - s['x'], 'y_vel': -2 * s['x_vel'], 'c0': 0.5 * s['x']},
        dependent_fn_keys=['y', 'y_vel', 'c0'])
    movers = sprite_generators.generate_sprites(
        distribs.Product([mirrored], shape='circle', scale=0.08, c1=1., c2=1.), num_sprites=3)
    ring = distribs.DependentDistribution(
        distribs.Continuous('angle', 0., 6.283, dtype='float64'),
        dependent_fn=lambda s: {'x': 0.5 + 0.3 * np.cos(s['angle']), 'y': 0.5 + 0.3 * np.sin(s['angle'])},
        dependent_fn_keys=['x', 'y'])

    def state_initializer():
        walls = shapes.border_walls(visible_thickness=0.05, c0=0., c1=0., c2=0.5)
        marker = sprite.Sprite(shape='triangle', scale=0.07, c0=0.6, c1=1., c2=1., **ring.sample())
        agent = sprite.Sprite(x=0.5, y=0.5, shape='square', scale=0.06, c0=0.33, c1=1., c2=0.7)
        return collections.OrderedDict([
            ('walls', walls), ('movers', movers(disjoint=True, without_overlapping=walls)), ('marker', [marker]),
            ('agent', [agent])])

    physics = physics_lib.Physics(
        (physics_lib.Collision(elasticity=1., symmetric=False, update_angle_vel=False), 'movers', 'walls'),
        (physics_lib.Collision(elasticity=1., symmetric=True, update_angle_vel=True), 'movers', 'movers'),
        (physics_lib.Drag(coeff_friction=0.25), 'agent'),
        updates_per_env_step=5)
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': tasks.CompositeTask(tasks.ContactReward(1., layers_0='agent', layers_1='movers'), timeout_steps=9),
        'action_space': action_spaces.Joystick(scaling_factor=0.01, action_layers='agent'),
        'observers': {'image': observers.PILRenderer(image_size=(64, 64), anti_aliasing=1, color_to_rgb='hsv_to_rgb')},
        'game_rules': (),
    }
