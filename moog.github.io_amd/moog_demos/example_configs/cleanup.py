"""Cleanup, the three-agent social dilemma (recipe restating the parameters of the reference's
moog_demos/example_configs/cleanup.py; the reference file itself also loads unchanged, see
tests/test_host.py).

Three agents on a Composite joystick action space (:150-157).  Fruits pay a reward to agent_0
while they are ripe (ContactReward with a pair condition, :145-148); touching a ripe fruit spoils
it (ModifyOnContact with a filter, :207-211); for every agent standing on a ripe fruit one clean
fountain gets poisoned, and for every agent on a poisoned fountain one fruit ripens
(ConditionalRule over ModifySprites(sample_one, filter_fn) with the loop condition of :181-205);
touching a poisoned fountain cleans it (:212-216).
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

_GOOD_VALUE = 1.
_BAD_VALUE = 0.3
_VALUE_THRESHOLD = 0.6


def get_config(_=0):
    agent_factors = distribs.Product(
        [distribs.Continuous('x', 0., 1.), distribs.Continuous('y', 0.35, 0.65)],
        shape='circle', scale=0.1, c1=1., c2=0.7)
    generators = [sprite_generators.generate_sprites(distribs.Product([agent_factors], c0=c0), num_sprites=1)
                  for c0 in (0.2, 0.1, 0.)]
    walls = shapes.border_walls(visible_thickness=0.05, c0=0., c1=0., c2=0.5)

    def grid(ys, c0):
        gx, gy = np.meshgrid(np.linspace(0.1, 0.9, 6), ys)
        return [sprite.Sprite(x=x, y=y, shape='circle', scale=0.05, c0=c0, c1=1., c2=_BAD_VALUE)
                for x, y in zip(np.ravel(gx), np.ravel(gy))]
    fountain_sprites = grid(np.linspace(0.75, 0.9, 2), 0.6)
    fruit_sprites = grid(np.linspace(0.1, 0.25, 2), 0.3)

    def state_initializer():
        agent_0 = generators[0](without_overlapping=walls)
        agent_1 = generators[1](without_overlapping=walls)
        agent_2 = generators[2](without_overlapping=walls)
        return collections.OrderedDict([
            ('walls', walls), ('fountains', fountain_sprites), ('fruits', fruit_sprites),
            ('agent_2', agent_2), ('agent_1', agent_1), ('agent_0', agent_0)])

    agents = ['agent_0', 'agent_1', 'agent_2']
    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.25), agents),
        (physics_lib.Collision(elasticity=0.25, symmetric=False), agents, 'walls'),
        updates_per_env_step=5)
    task = tasks.ContactReward(1, layers_0='agent_0', layers_1='fruits',
                               condition=lambda s_0, s_1: s_1.c2 > _VALUE_THRESHOLD)
    action_space = action_spaces.Composite(
        agent_0=action_spaces.Joystick(scaling_factor=0.005, action_layers='agent_0'),
        agent_1=action_spaces.Joystick(scaling_factor=0.005, action_layers='agent_1'),
        agent_2=action_spaces.Joystick(scaling_factor=0.005, action_layers='agent_2'))

    def _set_c2(value):
        def _modifier(s):
            s.c2 = value
        return _modifier

    def agents_contacting_layer(state, layer, value):
        n_contact = 0
        for s in state[layer]:
            if s.c2 != value:
                continue
            n_contact += (s.overlaps_sprite(state['agent_0'][0]) or
                          s.overlaps_sprite(state['agent_1'][0]) or
                          s.overlaps_sprite(state['agent_2'][0]))
        return n_contact

    poison_fountains = game_rules.ConditionalRule(
        condition=lambda s: agents_contacting_layer(s, 'fruits', _GOOD_VALUE),
        rules=game_rules.ModifySprites(layers='fountains', modifier=_set_c2(_BAD_VALUE), sample_one=True,
                                       filter_fn=lambda s: s.c2 > _VALUE_THRESHOLD))
    ripen_fruits = game_rules.ConditionalRule(
        condition=lambda s: agents_contacting_layer(s, 'fountains', _BAD_VALUE),
        rules=game_rules.ModifySprites(layers='fruits', modifier=_set_c2(_GOOD_VALUE), sample_one=True,
                                       filter_fn=lambda s: s.c2 < _VALUE_THRESHOLD))
    spoil_fruits = game_rules.ModifyOnContact(
        layers_0='fruits', layers_1=('agent_0', 'agent_1', 'agent_2'), modifier_0=_set_c2(_BAD_VALUE),
        filter_0=lambda s: s.c2 > _VALUE_THRESHOLD)
    clean_fountains = game_rules.ModifyOnContact(
        layers_0='fountains', layers_1=('agent_0', 'agent_1', 'agent_2'), modifier_0=_set_c2(_GOOD_VALUE),
        filter_0=lambda s: s.c2 < _VALUE_THRESHOLD)
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_space,
        'observers': {'image': observers.PILRenderer(image_size=(64, 64), anti_aliasing=1,
                                                     color_to_rgb='hsv_to_rgb'),
                      'state': observers.RawState()},
        'game_rules': (poison_fountains, spoil_fruits, ripen_fruits, clean_fountains),
    }
