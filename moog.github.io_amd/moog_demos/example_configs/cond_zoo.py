"""State-level condition exercise (parity fixture recipe): conditions that
read the environment state, traced symbolically by the engine
(moog/_symbolic.py trace_state_condition) or evaluated by the device.

  * Reset(all(s.c1 < 0.5 for s in state['prey'])) once every prey has been touched and greyed by
    ModifyOnContact, and the ContactReward pair condition on the prey's colour
    (parallelogram_catch.py:111-119,135-143)
  * ConditionalRule(get_contact_counter(...), CreateSprites(...)): the example of
    conditional.py:24-34 -- a new sprite for every prey / agent contact
  * ConditionalRule(state['agent'][0] is moving, ModifySprites(unglue)) (pacman.py:128-136)
  * ConditionalRule(any(s.x > c for s in state['extras']), VanishByFilter)
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


def get_config(_=0):
    extra_factors = distribs.Product(
        [distribs.Continuous('x', 0.1, 0.9), distribs.Continuous('y', 0.6, 0.9),
         distribs.Continuous('x_vel', 0.01, 0.05)],
        shape='triangle', scale=0.07, c0=0.8, c1=1., c2=1.)
    extra_gen = sprite_generators.generate_sprites(extra_factors, num_sprites=1)

    def state_initializer():
        walls = shapes.border_walls(visible_thickness=0.03, c0=0., c1=0., c2=0.5)
        prey = [sprite.Sprite(x=x, y=0.35, shape='square', scale=0.09, c0=0.2, c1=1., c2=1.)
                for x in (0.25, 0.5, 0.75)]
        ghosts = [sprite.Sprite(x=x, y=0.15, shape='circle', scale=0.06, mass=np.inf, c0=0., c1=1., c2=0.8,
                                x_vel=0.02, y_vel=0.01) for x in (0.3, 0.7)]
        agent = sprite.Sprite(x=0.5, y=0.55, shape='circle', scale=0.08, c0=0.33, c1=1., c2=0.66)
        return collections.OrderedDict([
            ('walls', walls), ('prey', prey), ('ghosts', ghosts), ('extras', []), ('agent', [agent])])

    def _make_prey_gray(s):
        s.c1 = 0.
        s.c2 = 0.6

    def _unglue(s):
        s.mass = 1.

    rules = (
        game_rules.ModifyOnContact(layers_0='agent', layers_1='prey', modifier_1=_make_prey_gray),
        game_rules.ConditionalRule(
            condition=game_rules.get_contact_counter('prey', 'agent'),
            rules=game_rules.CreateSprites('extras', extra_gen, without_overlapping=['walls', 'agent'])),
        game_rules.ConditionalRule(
            condition=lambda state: not np.all(state['agent'][0].velocity == 0),
            rules=game_rules.ModifySprites(('ghosts',), _unglue)),
        game_rules.ConditionalRule(
            condition=lambda state: any(s.x > 0.8 for s in state['extras']),
            rules=game_rules.VanishByFilter('extras', lambda s: s.x > 0.8)),
    )
    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.25), 'agent'),
        (physics_lib.Collision(elasticity=1., symmetric=False), ['ghosts', 'agent'], 'walls'),
        updates_per_env_step=5)
    task = tasks.CompositeTask(
        tasks.ContactReward(1, layers_0='agent', layers_1='prey',
                            condition=lambda s_agent, s_prey: s_prey.c1 > 0.5),
        tasks.Reset(condition=lambda state: all([s.c1 < 0.5 for s in state['prey']]),
                    steps_after_condition=3),
        timeout_steps=40)
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.Joystick(scaling_factor=0.05, action_layers='agent'),
        'observers': {'image': observers.PILRenderer(image_size=(64, 64), color_to_rgb='hsv_to_rgb')},
        'game_rules': rules,
    }
