"""Delayed match-to-sample: coloured discs sit on a ring around the agent; a grey screen lifts, the discs are shown,
then grey covers hide them and the ring turns for a while; when it stops a cue in the colour of ONE disc appears on the
agent, which may now move: touching the cover of the matching disc pays +1, any other cover -1.
Parameters: reference moog_demos/example_configs/match_to_sample.py:28-284 (get_config(num_targets)).

What the engine exercises here: an initializer that rejection-samples well separated angles in a `while` loop over
np.random (one device-side redraw loop per angle), sorts them (compare-exchange cells), a cue built from a deep copy of
the first disc's sampled factors, per-sprite metadata read by the reward function, a config-local rule that draws from
np.random when stepped and walks two layers in lock step, TetherZippedLayers about an anchor, FirstPersonAgent."""
import collections
import copy

import numpy as np
from moog import action_spaces, game_rules as gr, observers, physics as physics_lib, shapes, sprite, tasks
from moog.state_initialization import distributions as distribs

RING_RADIUS = 0.35
MIN_GAP = 0.7   # radians between any two discs


def ring_directions(count, gap):
    """Unit vectors (sin, cos) of `count` directions at least `gap` apart, the set turned by a random angle."""
    chosen = [0.]
    while len(chosen) < count:
        candidate = np.random.uniform(gap, 2 * np.pi - gap)
        if all([np.abs(candidate - other) > gap for other in chosen]):
            chosen.append(candidate)
    turned = np.sort(chosen)
    turned += np.random.uniform(0., 2 * np.pi)
    return np.stack((np.sin(turned), np.cos(turned)), axis=1)


class SpinRing(gr.AbstractRule):
    """Sets discs and covers turning about the centre, all with one angular speed of random size and sense."""

    def __init__(self, speed_range):
        self._speed_range = speed_range

    def step(self, state, meta_state):
        del meta_state
        omega = np.random.uniform(*self._speed_range)
        omega *= (2 * np.random.randint(2) - 1)
        quarter_turn = np.array([[0, -1], [1, 0]])
        for disc, cover in zip(state['targets'], state['covers']):
            arm = disc.position - 0.5
            tangent = np.matmul(quarter_turn, arm)
            swing = tangent * np.linalg.norm(arm) * omega
            disc.velocity = swing
            cover.velocity = swing


def get_config(num_targets):
    if num_targets == 0 or not isinstance(num_targets, int):
        raise ValueError('num_targets is %r, but must be a positive integer' % (num_targets,))
    screen = sprite.Sprite(x=0.5, y=0.5, shape='square', scale=2., c0=0.6, c1=0.7, c2=0.7)
    disc_colour = distribs.Product([distribs.Continuous('c0', 0., 1.)], shape='circle', scale=0.085, c1=1., c2=1.)
    cover_look = dict(mass=0., shape='circle', scale=0.1, c0=0., c1=0., c2=0.5, opacity=0)

    def state_initializer():
        spots = 0.5 + RING_RADIUS * ring_directions(num_targets, MIN_GAP)
        colours = [disc_colour.sample() for _ in range(num_targets)]
        discs = [sprite.Sprite(x=p[0], y=p[1], **look) for p, look in zip(spots, colours)]
        covers = [sprite.Sprite(x=p[0], y=p[1], **cover_look) for p in spots]
        for k, cover in enumerate(covers):   # the first disc is the one the cue will match
            cover.metadata = {'prey': k == 0}
        cue_look = copy.deepcopy(colours[0])
        cue_look['scale'] = 0.7 * colours[0]['scale']
        cue = sprite.Sprite(x=0.5, y=0.501, opacity=0, mass=np.inf, **cue_look)
        agent = sprite.Sprite(x=0.5, y=0.5, shape='circle', scale=0.1, c0=0.4, c1=0., c2=1., mass=np.inf)
        ring = sprite.Sprite(x=0.5, y=0.5, shape=shapes.annulus_vertices(0.34, 0.36), scale=1., c0=0., c1=0., c2=0.3)
        return collections.OrderedDict([
            ('annulus', [ring]), ('targets', discs), ('covers', covers), ('agent', [agent]), ('cue', [cue]),
            ('screen', [screen])])

    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.25), ['agent', 'cue']),
        updates_per_env_step=1,
        corrective_physics=[physics_lib.TetherZippedLayers(('targets', 'covers'), anchor=np.array([0.5, 0.5]))])

    def uncovered_in_response(state, meta_state):
        return state['covers'][0].opacity == 0 and meta_state['phase'] == 'response'

    task = tasks.CompositeTask(
        tasks.ContactReward(reward_fn=lambda _, s: 1 if s.metadata['prey'] else -1, layers_0='agent', layers_1='covers'),
        tasks.Reset(condition=uncovered_in_response, steps_after_condition=15),
        timeout_steps=800)

    def opaque(s):
        s.opacity = 255

    def transparent(s):
        s.opacity = 0

    def halt(s):
        s.angle_vel = 0.
        s.velocity = np.zeros(2)

    def unglue(s):
        s.mass = 1.

    phases = gr.PhaseSequence(
        gr.Phase(duration=1, name='screen'),
        gr.Phase(one_time_rules=gr.ModifySprites('screen', transparent), duration=2, name='visible'),
        gr.Phase(one_time_rules=[gr.ModifySprites('covers', opaque), SpinRing(speed_range=(0.1, 0.3))], duration=100,
                 name='motion'),
        gr.Phase(one_time_rules=[gr.ModifySprites('cue', opaque), gr.ModifySprites(('targets', 'covers'), halt),
                                 gr.ModifySprites(('agent', 'cue'), unglue)],
                 continual_rules=gr.ModifyOnContact(layers_0='agent', layers_1='covers', modifier_1=transparent),
                 name='response'),
        meta_state_phase_name_key='phase')

    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.Joystick(scaling_factor=0.01, action_layers=['agent', 'cue']),
        'observers': {'image': observers.PILRenderer(
            image_size=(64, 64), anti_aliasing=1, color_to_rgb='hsv_to_rgb',
            polygon_modifier=observers.polygon_modifiers.FirstPersonAgent(agent_layer='agent'))},
        'game_rules': (phases,),
        'meta_state_initializer': lambda: {'phase': ''},
    }
