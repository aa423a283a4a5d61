"""Multi-object tracking with a feature change: discs bounce around carrying oriented bars; after a fixation period an
occluder hides them, one bar then turns by 90 degrees, and the subject has to fixate the disc whose bar changed.
Parameters: reference moog_demos/example_configs/multi_tracking_with_feature.py:24-284 (get_config(num_targets)).

What the engine exercises here: a config-local distribution class that draws from the rng itself (RadialVelocity), bars
built from the discs' sampled factors (`Sprite(x=disc.x, ...)`), np.random.binomial in the initializer, Fixation rules
read by Phase end conditions through the meta-state, a Phase whose duration is drawn at reset, a config-local rule
class (lowered by tracing its step), TetherZippedLayers, SetPosition on two layers, a Reset task that pays a reward."""
import collections

import numpy as np
from moog import action_spaces, game_rules as gr, observers, physics as physics_lib, shapes, sprite, tasks
from moog.state_initialization import distributions as distribs

FIXATION_RADIUS = 0.1
_TOLERANCE = 1e-4


class RadialVelocity(distribs.AbstractDistribution):
    """A velocity of fixed speed in a uniformly drawn direction."""

    def __init__(self, speed):
        self._speed = speed

    def sample(self, rng):
        heading = self._get_rng(rng).uniform(0., 2 * np.pi)
        return {'x_vel': self._speed * np.cos(heading), 'y_vel': self._speed * np.sin(heading)}

    def contains(self, spec):
        return ('x_vel' in spec and 'y_vel' in spec and
                bool(np.abs(np.linalg.norm([spec['x_vel'], spec['y_vel']]) - self._speed) < _TOLERANCE))

    def to_str(self, indent):
        return indent * '  ' + 'RadialVelocity({})'.format(self._speed)

    @property
    def keys(self):
        return set(['x_vel', 'y_vel'])


class TurnFirstBar(gr.AbstractRule):
    """The first disc's bar turns by a quarter turn."""

    def step(self, state, meta_state):
        del meta_state
        bar = state['bars'][0]
        bar.angle = bar.angle + 0.5 * np.pi


def get_config(num_targets):
    disc_factors = distribs.Product(
        [distribs.Continuous('x', 0.1, 0.9), distribs.Continuous('y', 0.1, 0.9), RadialVelocity(speed=0.01)],
        scale=0.1, shape='circle', c0=0., c1=0., c2=0.9)
    bar_look = dict(scale=0.1, shape='square', aspect_ratio=0.3, c0=0., c1=0., c2=0.2)
    walls = [sprite.Sprite(shape=np.array(outline), x=0, y=0, c0=0., c1=0., c2=0.5)
             for outline in ([[-1, 0], [2, 0], [2, -1], [-1, -1]], [[-1, 1], [2, 1], [2, 2], [-1, 2]],
                             [[0, -1], [0, 4], [-1, 4], [-1, -1]], [[1, -1], [1, 4], [2, 4], [2, -1]])]
    cross = 0.1 * np.array([[-5, 1], [-1, 1], [-1, 5], [1, 5], [1, 1], [5, 1], [5, -1], [1, -1], [1, -5], [-1, -5],
                            [-1, -1], [-5, -1]])

    def state_initializer():
        fixation = sprite.Sprite(x=0.5, y=0.5, shape=cross, scale=0.1, c0=0., c1=0., c2=0.)
        screen = sprite.Sprite(x=0.5, y=0.5, shape='square', scale=2., c0=0., c1=0., c2=1.)
        agent = sprite.Sprite(x=0.5, y=0.5, scale=0.04, shape=cross, c0=0.33, c1=1., c2=1.)
        occluder = sprite.Sprite(shape=shapes.annulus_vertices(0.13, 2.), x=0.5, y=0.5, c0=0.6, c1=0.25, c2=0.5, opacity=0)
        discs = [sprite.Sprite(**disc_factors.sample()) for _ in range(num_targets)]
        quarter_turns = 0.5 * np.pi * np.random.binomial(1, 0.5, (num_targets))
        bars = [sprite.Sprite(x=d.x, y=d.y, x_vel=d.x_vel, y_vel=d.y_vel, angle=turn, **bar_look)
                for d, turn in zip(discs, quarter_turns)]
        return collections.OrderedDict([
            ('walls', walls), ('targets', discs), ('bars', bars), ('occluder', [occluder]), ('screen', [screen]),
            ('fixation', [fixation]), ('agent', [agent])])

    physics = physics_lib.Physics(
        (physics_lib.Collision(elasticity=1., symmetric=False, update_angle_vel=False), 'targets', 'walls'),
        updates_per_env_step=10,
        corrective_physics=[physics_lib.TetherZippedLayers(layer_names=('targets', 'bars'), update_angle_vel=False)])

    task = tasks.Reset(condition=lambda _, meta_state: meta_state['phase'] == 'reward', reward_fn=lambda _: 1,
                       steps_after_condition=10)

    def opaque(s):
        s.opacity = 255

    def transparent(s):
        s.opacity = 0

    def halt(s):
        s.velocity = np.zeros(2)

    phases = gr.PhaseSequence(
        gr.Phase(continual_rules=gr.Fixation('agent', 'fixation', FIXATION_RADIUS, 'fixation_duration'),
                 end_condition=lambda _, meta_state: meta_state['fixation_duration'] >= 15, name='fixation'),
        gr.Phase(one_time_rules=[gr.VanishByFilter('fixation', lambda _: True), gr.VanishByFilter('screen', lambda _: True)],
                 duration=5, name='visible'),
        gr.Phase(one_time_rules=gr.ModifySprites('occluder', opaque), duration=lambda: np.random.randint(40, 80),
                 name='tracking'),
        gr.Phase(one_time_rules=TurnFirstBar(),
                 continual_rules=gr.Fixation('agent', 'targets', FIXATION_RADIUS, 'response_duration'),
                 end_condition=lambda _, meta_state: meta_state['response_duration'] >= 30, name='change'),
        gr.Phase(one_time_rules=(gr.ModifySprites('occluder', transparent), gr.ModifySprites(('targets', 'bars'), halt)),
                 name='reward'),
        meta_state_phase_name_key='phase')

    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.SetPosition(action_layers=('agent', 'occluder')),
        'observers': {'image': observers.PILRenderer(image_size=(64, 64), anti_aliasing=1,
                                                     color_to_rgb=observers.color_maps.hsv_to_rgb),
                      'state': observers.RawState()},
        'game_rules': (phases,),
        'meta_state_initializer': lambda: {'phase': ''},
    }
