"""Red or green first?  A blue ball bounces around a walled arena that holds a red box, a green box and `level` grey
obstacles; the subject predicts which of the two coloured boxes the ball will reach first by walking a token onto the
red or the green answer square at the bottom (after Smith, Peres, Vul & Tenenbaum, CogSci 2017).
Parameters: reference moog_demos/example_configs/red_green.py:31-268 (get_config(level) = number of obstacles).

What the engine exercises here: an initializer that REJECTS whole trials -- it starts over when a fail_gracefully
generator came out short, and when its look-ahead (the physics stepped for up to 150 steps inside the reset) says the
ball reaches a box too early or not at all --, a `for step in range(n)` look-ahead with a test on the step number, a
config-local distribution that draws inside generate_sprites, one generator call whose sprites are spread over three
layers in another order and two of which are repainted, and a reward function that compares the outcome the look-ahead
stored in the token's metadata with the answer square's colour."""
import collections

import numpy as np
from moog import action_spaces, game_rules, observers, physics as physics_lib, sprite, tasks
from moog.state_initialization import distributions as distribs
from moog.state_initialization import sprite_generators

_TOLERANCE = 1e-4


class Heading(distribs.AbstractDistribution):
    """A velocity of given speed in a uniformly drawn direction."""

    def __init__(self, speed):
        self._speed = speed

    def sample(self, rng):
        angle = self._get_rng(rng).uniform(0., 2 * np.pi)
        return {'x_vel': self._speed * np.cos(angle), 'y_vel': self._speed * np.sin(angle)}

    def contains(self, spec):
        return ('x_vel' in spec and 'y_vel' in spec and
                bool(np.abs(np.linalg.norm([spec['x_vel'], spec['y_vel']]) - self._speed) < _TOLERANCE))

    def to_str(self, indent):
        return indent * '  ' + 'Heading({})'.format(self._speed)

    @property
    def keys(self):
        return set(['x_vel', 'y_vel'])


def get_config(level):
    if not isinstance(level, int):
        raise ValueError('level is %r, but must be an integer.' % (level,))
    num_obstacles, earliest, latest = level, 50, 150
    physics = physics_lib.Physics(
        (physics_lib.Collision(elasticity=1., symmetric=False, update_angle_vel=False), 'ball', 'walls'),
        updates_per_env_step=10)

    def play_forward(state):
        """(the ball reaches a coloured box within [earliest, latest) steps, 0 for red / 1 for green)"""
        ball = state['ball'][0]
        for step in range(latest):
            on_red = ball.overlaps_sprite(state['red'][0])
            on_green = ball.overlaps_sprite(state['green'][0])
            if on_red or on_green:
                if step < earliest:
                    return False, None
                return True, (0 if on_red else 1)
            physics.step(state)
        return False, None

    make_ball = sprite_generators.generate_sprites(
        distribs.Product([distribs.Continuous('x', 0.15, 0.85), distribs.Continuous('y', 0.15, 0.85), Heading(speed=0.03)],
                         scale=0.05, shape='circle', c0=64, c1=64, c2=255),
        num_sprites=1, max_recursion_depth=100, fail_gracefully=True)
    make_boxes = sprite_generators.generate_sprites(
        distribs.Product([distribs.Continuous('x', 0.2, 0.8), distribs.Continuous('y', 0.2, 0.8)],
                         scale=0.2, shape='square', c0=128, c1=128, c2=128),
        num_sprites=2 + num_obstacles, max_recursion_depth=100, fail_gracefully=True)
    border = [sprite.Sprite(shape=np.array(outline), x=0, y=0, c0=128, c1=128, c2=128)
              for outline in ([[-1, 0.1], [2, 0.1], [2, -1], [-1, -1]], [[-1, 0.95], [2, 0.95], [2, 2], [-1, 2]],
                              [[0.05, -1], [0.05, 4], [-1, 4], [-1, -1]], [[0.95, -1], [0.95, 4], [2, 4], [2, -1]])]

    def state_initializer():
        boxes = make_boxes(disjoint=True)
        ball = make_ball(without_overlapping=boxes)
        if len(boxes) < num_obstacles + 2 or not ball:   # no room was found: another try
            return state_initializer()
        red, green, grey = boxes[0], boxes[1], boxes[2:]
        red.c0, red.c1, red.c2 = 255, 0, 0
        green.c0, green.c1, green.c2 = 0, 255, 0
        token = sprite.Sprite(x=0.5, y=0.06, shape='spoke_4', scale=0.03, c0=255, c1=255, c2=255)
        answers = [sprite.Sprite(x=0.6, y=0.06, shape='square', scale=0.03, c0=255, c1=0, c2=0),
                   sprite.Sprite(x=0.4, y=0.06, shape='square', scale=0.03, c0=0, c1=255, c2=0)]
        state = collections.OrderedDict([
            ('walls', border + grey), ('red', [red]), ('green', [green]), ('ball', ball), ('responses', answers),
            ('agent', [token])])
        started_at = np.copy(ball[0].position)
        started_with = np.copy(ball[0].velocity)
        usable, first_colour = play_forward(state)
        if not usable:                                   # too early, or never: another trial
            return state_initializer()
        ball[0].position = started_at
        ball[0].velocity = started_with
        token.metadata = {'true_contact_color': first_colour}
        return state

    def verdict(token, answer):
        says_green = answer.c0 < 128
        return 1. if token.metadata['true_contact_color'] == says_green else -1.

    task = tasks.CompositeTask(
        tasks.ContactReward(reward_fn=verdict, layers_0='agent', layers_1='responses', reset_steps_after_contact=10),
        timeout_steps=400)

    def halt(s):
        s.velocity = np.zeros(2)

    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.Grid(scaling_factor=0.015, action_layers='agent', control_velocity=True),
        'observers': {'image': observers.PILRenderer(image_size=(64, 64), anti_aliasing=1)},
        'game_rules': (game_rules.ModifyOnContact(layers_0='ball', layers_1=('red', 'green'), modifier_0=halt),),
    }
