"""Factor-distribution exercise (parity fixture recipe).

Every reset samples sprites from the composite distributions of
moog/state_initialization/distributions.py:159-405:
    prey     the boundary-position Mixture and velocity-annulus SetMinus of
             first_person_predators_prey.py:38-70
    blobs    Selection, Intersection (index_for_sampling=1), a Mixture with non-uniform
             probs whose components draw a factor as float32 in one branch and pick a
             constant in the other, Discrete with probs
Episodes time out after 4 steps so that a fixture run crosses many resets.
"""
import collections

from moog import action_spaces
from moog import observers
from moog import physics as physics_lib
from moog import shapes
from moog import sprite
from moog import tasks
from moog.state_initialization import distributions as distribs
from moog.state_initialization import sprite_generators


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
    prey_factors = distribs.Product(
        [_boundary_positions(0.1), _annulus_velocity(0.01, 0.02),
         distribs.Continuous('scale', 0.07, 0.13)],
        shape='circle', c0=0.2, c1=1., c2=1.)
    blob_position = distribs.Intersection([
        distribs.Product([distribs.Continuous('x', 0.1, 0.6), distribs.Continuous('y', 0.1, 0.9)]),
        distribs.Product([distribs.Continuous('x', 0.4, 0.9), distribs.Continuous('y', 0.2, 0.5)]),
    ], index_for_sampling=1)
    blob_motion = distribs.Mixture([
        distribs.Product([distribs.Continuous('x_vel', -0.02, 0.02),
                          distribs.Continuous('y_vel', -0.02, 0.02),
                          distribs.Continuous('angle_vel', -0.2, 0.2)]),
        distribs.Product([distribs.Continuous('x_vel', -0.02, 0.02)], y_vel=0.01, angle_vel=0.),
        distribs.Product([distribs.Discrete('x_vel', [-0.01, 0.01]),
                          distribs.Discrete('y_vel', [-0.01, 0., 0.01], probs=[0.25, 0.5, 0.25])],
                         angle_vel=0.1),
    ], probs=[0.5, 0.2, 0.3])
    blob_look = distribs.Selection(
        distribs.Product([distribs.Continuous('c0', 0., 1.), distribs.Continuous('angle', 0., 6.),
                          distribs.Discrete('shape', ['square', 'triangle', 'star_5', 'circle'],
                                            probs=[0.1, 0.4, 0.3, 0.2])]),
        filtering=distribs.Mixture([distribs.Continuous('c0', 0.1, 0.3),
                                    distribs.Continuous('c0', 0.6, 0.95)]))
    blob_factors = distribs.Product(
        [blob_position, blob_motion, blob_look, distribs.Continuous('scale', 0.05, 0.12)],
        c1=0.8, c2=0.9, mass=2.)
    prey_gen = sprite_generators.generate_sprites(prey_factors, num_sprites=3)
    blob_gen = sprite_generators.generate_sprites(blob_factors, num_sprites=4)

    def state_initializer():
        walls = shapes.border_walls(visible_thickness=0.05, c0=0., c1=0., c2=0.5)
        agent = [sprite.Sprite(x=0.5, y=0.5, shape='circle', scale=0.04, c0=0.33, c1=1., c2=0.66)]
        return collections.OrderedDict([
            ('walls', walls), ('prey', prey_gen()), ('blobs', blob_gen(without_overlapping=agent)),
            ('agent', agent)])

    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.1), ['agent', 'blobs']),
        (physics_lib.Collision(elasticity=1., symmetric=False), 'blobs', 'walls'),
        updates_per_env_step=5)
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': tasks.CompositeTask(timeout_steps=4),
        'action_space': action_spaces.Joystick(scaling_factor=0.01, action_layers='agent'),
        'observers': {'image': observers.PILRenderer(image_size=(64, 64), color_to_rgb='hsv_to_rgb')},
    }
