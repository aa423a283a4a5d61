"""Config-callable exercise (parity fixture recipe): the lambdas configs
hand to rules and tasks, traced symbolically by the engine (moog/_symbolic.py).

Modelled on cleanup.py:150-216 (ModifySprites with filter_fn / sample_one, ModifyOnContact
with filters, ContactReward with a pair condition) and first_person_predators_prey.py:129-147,
193-201 (reward_fn of the contacted sprite's scale, VanishByFilter on position / velocity).
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

_GOOD, _BAD, _THRESH = 1., 0.3, 0.6


def get_config(_=0):
    mover_factors = distribs.Product(
        [distribs.Continuous('x', 0.15, 0.85), distribs.Continuous('y', 0.35, 0.65),
         distribs.Continuous('x_vel', -0.06, 0.06), distribs.Continuous('y_vel', -0.06, 0.06),
         distribs.Continuous('scale', 0.05, 0.1), distribs.Discrete('c0', [0.0, 0.8])],
        shape='circle', c1=1., c2=0.8)
    mover_gen = sprite_generators.generate_sprites(mover_factors, num_sprites=4)
    xs = np.linspace(0.15, 0.85, 5)
    fountains = [sprite.Sprite(x=x, y=0.85, shape='circle', scale=0.05, c0=0.6, c1=1., c2=_BAD)
                 for x in xs]
    fruits = [sprite.Sprite(x=x, y=0.15, shape='circle', scale=0.05, c0=0.3, c1=1.,
                            c2=_GOOD if i % 2 else _BAD) for i, x in enumerate(xs)]

    def state_initializer():
        walls = shapes.border_walls(visible_thickness=0.03, c0=0., c1=0., c2=0.5)
        agent = sprite.Sprite(x=0.5, y=0.5, shape='circle', scale=0.08, c0=0.33, c1=1., c2=0.66)
        # the module-level sprites are mutated by the rules: fresh copies every episode
        return collections.OrderedDict([
            ('walls', walls),
            ('fountains', [sprite.Sprite(x=x, y=0.85, shape='circle', scale=0.05, c0=0.6, c1=1., c2=_BAD)
                           for x in xs]),
            ('fruits', [sprite.Sprite(x=x, y=0.15, shape='circle', scale=0.05, c0=0.3, c1=1.,
                                      c2=_GOOD if i % 2 else _BAD) for i, x in enumerate(xs)]),
            ('bin', []),
            ('movers', mover_gen()),
            ('agent', [agent]),
        ])
    del fountains, fruits

    def _set_c2(value):
        def _modifier(s):
            s.c2 = value
        return _modifier

    def _reverse(s):
        s.velocity = -s.velocity

    def _fatten(s):
        if s.mass < 3.:
            s.mass = s.mass * 1.5
        s.angle_vel = 0.1

    lo, hi = 0.1, 0.9

    def _should_vanish(s):
        too_small = (s.position < lo) * (s.velocity < 0.)
        too_large = (s.position > hi) * (s.velocity > 0.)
        return any(too_small) or any(too_large)

    rules = (
        game_rules.ModifySprites('fountains', _set_c2(_GOOD), sample_one=True,
                                 filter_fn=lambda s: s.c2 < _THRESH),
        game_rules.ModifyOnContact('fruits', ('agent', 'movers'), modifier_0=_set_c2(_BAD),
                                   filter_0=lambda s: s.c2 > _THRESH),
        game_rules.ModifyOnContact('movers', 'agent', modifier_0=_reverse, modifier_1=_fatten,
                                   filter_0=lambda s: np.linalg.norm(s.velocity) > 0.02),
        game_rules.VanishByFilter('movers', _should_vanish),
        game_rules.ChangeLayer('fountains', 'bin', filter_fn=lambda s: s.c2 > _THRESH and s.x > 0.6),
        game_rules.ModifySprites(['fruits', 'bin'], _set_c2(0.95), filter_fn=lambda s: s.x < 0.2),
    )
    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.25), 'agent'),
        (physics_lib.Collision(elasticity=0.25, symmetric=False), 'agent', 'walls'),
        updates_per_env_step=5)
    task = tasks.CompositeTask(
        tasks.ContactReward(lambda _, m: -2. * m.scale, layers_0='agent', layers_1='movers',
                            condition=lambda a, m: m.c0 > 0.5),
        tasks.ContactReward(lambda a, m: m.scale + a.mass, layers_0='agent', layers_1='movers',
                            condition=lambda a, m: m.c0 < 0.5),
        tasks.ContactReward(1, layers_0='agent', layers_1='fruits',
                            condition=lambda s_0, s_1: s_1.c2 > _THRESH),
        timeout_steps=14)
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.Joystick(scaling_factor=0.03, action_layers='agent'),
        'observers': {'image': observers.PILRenderer(image_size=(64, 64), color_to_rgb='hsv_to_rgb')},
        'game_rules': rules,
    }
