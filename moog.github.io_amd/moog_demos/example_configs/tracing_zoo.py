"""Coverage recipe (not a reference task): the host-side tracing features of round 3 in shapes other than the reference
configs that motivated them (match_to_sample.py, predators_arena.py) -- pinned by golden vectors captured from the
reference (tests/golden/tracing_zoo_*.npz).

level 0: an initializer that rejection-samples FIVE heights at least a gap apart (up to four accept tests per draw),
    sorts them, and builds a ladder of bars from them; bars carry a numeric `worth` in their metadata (a per-slot
    table on the device: the layer has several sprites) that the reward function multiplies; a config-local rule that
    takes THREE draws when stepped (a speed, a sign, a phase) and walks three layers in lock step.
level 1: an initializer object that keeps TWO numbers across episodes (a speed that grows by 7 % per reset up to a
    ceiling, and a shrinking size) and builds the movers from them."""
import collections

import numpy as np
from moog import action_spaces, game_rules as gr, observers, physics as physics_lib, shapes, sprite, tasks
from moog.state_initialization import distributions as distribs
from moog.state_initialization import sprite_generators


def spaced_heights(count, gap):
    heights = []
    while len(heights) < count:
        h = np.random.uniform(0.15, 0.9)
        if all([np.abs(h - other) > gap for other in heights]):
            heights.append(h)
    return np.sort(heights)


class Shove(gr.AbstractRule):
    """Every bar, its marker and its shadow get the same sideways velocity: one speed, one sense and one phase drawn per
    step of the rule, scaled by the bar's height."""

    def step(self, state, meta_state):
        del meta_state
        speed = np.random.uniform(0.004, 0.012)
        sense = 2 * np.random.randint(2) - 1
        phase = np.random.uniform(0., 1.)
        for bar, marker, shadow in zip(state['bars'], state['markers'], state['shadows']):
            push = np.array([1., 0.]) * (speed * sense) * (0.5 + bar.y) + np.array([0., 0.002]) * np.cos(phase + marker.x)
            bar.velocity = push
            marker.velocity = push
            shadow.velocity = push * 0.5


class Ramp(object):
    """Keeps a speed and a size across episodes."""

    def __init__(self, count):
        self._speed, self._size = 0.01, 0.12
        self._count = count
        self._walls = shapes.border_walls(visible_thickness=0.03, c0=0.6, c1=0.3, c2=0.4)
        self._meta_state = None

    def state_initializer(self):
        spots = sprite_generators.generate_sprites(
            distribs.Product([distribs.Continuous('x', 0.2, 0.8), distribs.Continuous('y', 0.3, 0.8)],
                             shape='square', scale=0.05, c0=0.9, c1=0.2, c2=0.3), num_sprites=self._count)
        anchors = spots(without_overlapping=self._walls)
        if self._meta_state is not None:
            self._speed = np.minimum(self._speed + self._speed * 0.07, 0.016)
            self._size -= self._size * 0.05
        movers = [sprite.Sprite(x=0.2 + 0.2 * k, y=0.55, shape='circle', scale=self._size, x_vel=self._speed * (1 + k),
                                y_vel=-0.5 * self._speed, c0=0.1 * k, c1=1., c2=1.) for k in range(3)]
        agent = sprite.Sprite(x=0.5, y=0.15, shape='triangle', scale=0.06, c0=0.33, c1=1., c2=1.)
        return collections.OrderedDict([('walls', self._walls), ('anchors', anchors), ('movers', movers),
                                        ('agent', [agent])])

    def meta_state_initializer(self):
        self._meta_state = {'episodes': 0}
        return self._meta_state


def get_config(level):
    renderer = observers.PILRenderer(image_size=(64, 64), color_to_rgb='hsv_to_rgb')
    if level == 1:
        ramp = Ramp(2)
        bounce = physics_lib.Collision(elasticity=1., symmetric=False, update_angle_vel=False)
        return {
            'state_initializer': ramp.state_initializer,
            'physics': physics_lib.Physics((bounce, 'movers', 'walls'), updates_per_env_step=4),
            'task': tasks.CompositeTask(tasks.ContactReward(1, layers_0='agent', layers_1='movers',
                                                            reset_steps_after_contact=2), timeout_steps=30),
            'action_space': action_spaces.Joystick(scaling_factor=0.02, action_layers='agent'),
            'observers': {'image': renderer},
            'meta_state_initializer': ramp.meta_state_initializer,
        }
    worths = (3, -2, 5, 1, -4)

    def state_initializer():
        heights = spaced_heights(5, 0.06)
        bars = [sprite.Sprite(x=0.5, y=h, shape='square', scale=0.2, aspect_ratio=0.15, c0=0.15 * k, c1=1., c2=1.)
                for k, h in enumerate(heights)]
        for bar, worth in zip(bars, worths):
            bar.metadata = {'worth': worth}
        markers = [sprite.Sprite(x=0.15, y=h, shape='circle', scale=0.04, c0=0.5, c1=0.5, c2=1.) for h in heights]
        shadows = [sprite.Sprite(x=0.85, y=h, shape='circle', scale=0.04, c0=0., c1=0., c2=0.4, opacity=120) for h in heights]
        agent = sprite.Sprite(x=0.5, y=0.05, shape='triangle', scale=0.06, c0=0.33, c1=1., c2=1.)
        return collections.OrderedDict([('bars', bars), ('markers', markers), ('shadows', shadows), ('agent', [agent])])

    task = tasks.CompositeTask(
        tasks.ContactReward(reward_fn=lambda a, bar: 2 * bar.metadata['worth'], layers_0='agent', layers_1='bars',
                            reset_steps_after_contact=4),
        timeout_steps=60)
    rules = (gr.TimedRule(step_interval=(3, 50), rules=(Shove(),)),
             gr.ModifySprites(('bars', 'markers', 'shadows'), lambda s: setattr(s, 'position', np.remainder(s.position, 1))))
    return {
        'state_initializer': state_initializer,
        'physics': physics_lib.Physics((physics_lib.Drag(coeff_friction=0.05), 'agent'), updates_per_env_step=2),
        'task': task,
        'action_space': action_spaces.Joystick(scaling_factor=0.02, action_layers='agent'),
        'observers': {'image': renderer},
        'game_rules': rules,
    }
