"""Config Python that round 4 lowers (parity fixture recipe; VERDICT r03 "missing" 2, 4, 5 and ADVICE r03):

  * DistanceForce(force_fn=<any function of the distance>) (distance_fn_force.py:16-47): a softened inverse-square pull
    with a cut-off, written as ordinary Python with an `if`
  * ContactReward(condition(s0, s1, meta_state)) (contact_reward.py:58-63) reading the phase a PhaseSequence publishes
  * DelayedRule / TemporaryRule / TimedRule with callable (random) intervals (timing.py:28-31,84-86,105-108)
  * Reset(condition = "layer empty", reward_fn = <reads the state>) (reset.py:52-57): the kernel variant that evaluates
    expressions must be selected for the reward alone
  * level 1: Reset(condition = <sprites named by position>) while the state's FIRST layer is empty: the condition is
    anchored to slots, not to a layer's first live sprite
  * level 3: DelayedRule(start=<callable>, ..., duration=<callable>): two np.random.randint draws per reset (timing.py:84-86)
  * level 2: PILRenderer(color_to_rgb=<a Python function>) (pil_renderer.py:72-76,108): branches, int(), components that
    Pillow clips to 0 .. 255 -- evaluated on the host per distinct colour (moog_engine_set_color_override)
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


def _pull(distance):
    if distance < 0.45:
        return 0.0006 / (distance * distance + 0.02)
    return 0.


def _palette(color):
    h, s, v = color
    if h > 0.8:   # the tinted sprites
        return (300, int(64 * s), -20)   # (Pillow clips: 255, 64, 0)
    if s == 0.:
        return (int(90 * v), int(90 * v), int(90 * v) + 100)
    return (int(255 * v * (1 - s * h)), int(200 * h) + 20, 255 - int(255 * h))


def get_config(level=0):
    prey_factors = distribs.Product(
        [distribs.Continuous('x', 0.15, 0.85), distribs.Continuous('y', 0.55, 0.85)],
        shape='circle', scale=0.06, c0=0.15, c1=1., c2=1.)
    prey_gen = sprite_generators.generate_sprites(prey_factors, num_sprites=3)

    def state_initializer():
        walls = shapes.border_walls(visible_thickness=0.03, c0=0., c1=0., c2=0.5)
        agent = sprite.Sprite(x=0.5, y=0.3, shape='square', scale=0.08, c0=0.33, c1=1., c2=0.66)
        magnet = sprite.Sprite(x=0.5, y=0.5, shape='triangle', scale=0.07, c0=0.6, c1=1., c2=1., mass=np.inf)
        prey = prey_gen(disjoint=True, without_overlapping=walls + [agent, magnet])
        first = [] if level == 1 else walls
        rest = walls if level == 1 else []
        return collections.OrderedDict([
            ('walls', first), ('prey', prey), ('magnet', [magnet]), ('agent', [agent]), ('fence', rest)])

    def _speed_up(s):
        s.velocity = 1.5 * s.velocity

    def _tint(s):
        s.c0 = 0.85

    phases = game_rules.PhaseSequence(
        game_rules.Phase(duration=6, name='wait'),
        game_rules.Phase(duration=np.inf, name='go'),
        meta_state_phase_name_key='phase')
    rules = (
        phases,
        # (level 3: the duration is a callable too -- two draws per reset, the start first: timing.py:84-86)
        game_rules.DelayedRule(lambda: np.random.randint(3, 9), game_rules.ModifySprites('prey', _tint),
                               duration=(lambda: np.random.randint(2, 7)) if level == 3 else 4),
        game_rules.TemporaryRule(lambda: np.random.randint(5, 12), game_rules.ModifySprites('agent', _speed_up)),
        game_rules.TimedRule(lambda: (2, np.random.randint(10, 20)), game_rules.ModifySprites('magnet', _tint)),
        game_rules.VanishOnContact('prey', 'agent'),
    )
    wall_layer = 'fence' if level == 1 else 'walls'
    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.2), ['agent', 'prey']),
        (physics_lib.DistanceForce(_pull, symmetric=False), 'magnet', 'prey'),
        (physics_lib.DistanceForce(lambda d: -0.0004 * np.sqrt(d) + 0.0001, symmetric=True), 'prey', 'prey'),
        (physics_lib.Collision(elasticity=0.9, symmetric=False), ['agent', 'prey'], wall_layer),
        updates_per_env_step=5)
    if level == 1:   # sprites named by position, with the state's first layer empty
        reset = tasks.Reset(condition=lambda state: state['agent'][0].y > 0.8 or state['magnet'][0].c0 > 0.8,
                            reward_fn=lambda state: 0.5, steps_after_condition=2)
    else:
        reset = tasks.Reset(condition=lambda state: len(state['prey']) == 0,
                            reward_fn=lambda state: 10. * state['agent'][0].x + 1., steps_after_condition=2)
    task = tasks.CompositeTask(
        tasks.ContactReward(1., layers_0='agent', layers_1='prey',
                            condition=lambda s_agent, s_prey, meta_state: meta_state['phase'] == 'go'),
        reset,
        timeout_steps=60)
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.Joystick(scaling_factor=0.03, action_layers='agent'),
        'observers': {'image': observers.PILRenderer(image_size=(64, 64),
                                                     color_to_rgb=_palette if level == 2 else 'hsv_to_rgb')},
        'game_rules': rules,
        'meta_state_initializer': lambda: {'phase': ''},
    }
