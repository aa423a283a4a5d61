"""Phase / PhaseSequence exercise (parity fixture recipe), modelled on the
trial structure of match_to_sample.py:214-262: a screen phase, a visible phase, a motion phase
whose one-time rules cover the targets and set them moving, and a response phase with a
continual ModifyOnContact; the Reset task tests the state and the phase name the sequence
publishes in the meta-state (:176-186).  A stand-alone Phase with an end condition runs next
to the sequence.  Level 1: random durations (task_phases.py:72: drawn in reset, AFTER the phase's
own rules were reset) on a phase of the sequence and on a stand-alone phase that holds another
random-duration phase as its continual rule -- the inner duration is drawn first.
"""
import collections

import numpy as np

from moog import action_spaces
from moog import game_rules as gr
from moog import observers
from moog import physics as physics_lib
from moog import sprite
from moog import tasks


def get_config(level=0):
    positions = [(0.25, 0.7), (0.5, 0.75), (0.75, 0.7)]

    def state_initializer():
        targets = [sprite.Sprite(x=x, y=y, shape='circle', scale=0.085, c0=0.1 + 0.3 * i, c1=1., c2=1.)
                   for i, (x, y) in enumerate(positions)]
        covers = [sprite.Sprite(x=x, y=y, shape='circle', scale=0.1, c0=0., c1=0., c2=0.5, opacity=0)
                  for x, y in positions]
        agent = sprite.Sprite(x=0.5, y=0.3, shape='circle', scale=0.1, c0=0.4, c1=0., c2=1., mass=np.inf)
        cue = sprite.Sprite(x=0.5, y=0.1, shape='square', scale=0.06, c0=0.1, c1=1., c2=1., opacity=0)
        screen = sprite.Sprite(x=0.5, y=0.5, shape='square', scale=2., c0=0.6, c1=0.7, c2=0.7)
        return collections.OrderedDict([
            ('targets', targets), ('covers', covers), ('agent', [agent]), ('cue', [cue]),
            ('screen', [screen])])

    def _make_opaque(s):
        s.opacity = 255

    def _make_transparent(s):
        s.opacity = 0

    def _drift(s):
        s.velocity = np.array([0.02, -0.03])

    def _stop(s):
        s.angle_vel = 0.
        s.velocity = np.zeros(2)

    def _unglue(s):
        s.mass = 1.

    screen_phase = gr.Phase(duration=1, name='screen')
    visible_phase = gr.Phase(one_time_rules=gr.ModifySprites('screen', _make_transparent), duration=2,
                             name='visible')
    motion_phase = gr.Phase(
        one_time_rules=[gr.ModifySprites('covers', _make_opaque),
                        gr.ModifySprites(('targets', 'covers'), _drift)],
        duration=4 if level == 0 else (lambda: np.random.randint(3, 7)), name='motion')
    response_phase = gr.Phase(
        one_time_rules=[gr.ModifySprites('cue', _make_opaque), gr.ModifySprites(('targets', 'covers'), _stop),
                        gr.ModifySprites(('agent', 'cue'), _unglue)],
        continual_rules=gr.ModifyOnContact(layers_0='agent', layers_1='covers',
                                           modifier_1=_make_transparent),
        name='response')
    phase_sequence = gr.PhaseSequence(screen_phase, visible_phase, motion_phase, response_phase,
                                      meta_state_phase_name_key='phase')
    # a stand-alone phase: dims the cue every step until the agent has moved to the right
    fade = gr.Phase(
        continual_rules=gr.ModifySprites('cue', lambda s: setattr(s, 'c2', s.c2 * 0.9)),
        end_condition=lambda state: any(s.x > 0.6 for s in state['agent']), name='fade')
    if level == 1:
        inner = gr.Phase(
            continual_rules=gr.ModifySprites('cue', lambda s: setattr(s, 'c2', s.c2 * 0.9)),
            duration=lambda: np.random.randint(2, 5), name='inner')
        fade = gr.Phase(continual_rules=inner, duration=lambda: np.random.randint(6, 10), name='outer')

    def _should_reset(state, meta_state):
        return state['covers'][0].opacity == 0 and meta_state['phase'] == 'response'

    task = tasks.CompositeTask(
        tasks.ContactReward(1, layers_0='agent', layers_1='covers'),
        tasks.Reset(condition=_should_reset, steps_after_condition=4),
        timeout_steps=40)
    physics = physics_lib.Physics((physics_lib.Drag(coeff_friction=0.25), ['agent', 'cue']),
                                  updates_per_env_step=1)
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.Joystick(scaling_factor=0.03, action_layers=['agent', 'cue']),
        'observers': {'image': observers.PILRenderer(image_size=(64, 64), color_to_rgb='hsv_to_rgb')},
        'game_rules': (phase_sequence, fade),
        'meta_state_initializer': lambda: {'phase': ''},
    }
