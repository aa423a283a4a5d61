"""Will they touch?  Two red balls drop into an open-topped box, bounce off its floor and walls and leave through the
top; the lower part of the box is hidden by an occluder (opaque, or translucent at level True).  The subject answers by
walking a token into the left box ("they will touch") or the right one ("they will not"); the correct answer is known
because the initializer plays the whole episode forward before it starts.
Parameters: reference moog_demos/example_configs/bounce_box_contact_prediction.py:24-187 (get_config(translucent_occluder)).

What the engine exercises here: an initializer that STEPS THE PHYSICS in a loop on the state it has just built to look
ahead (one device-side op: physics steps until one of the loop's exits holds), snapshots positions / velocities with
np.copy beforehand and assigns them back afterwards, stores the outcome in a sprite's metadata, which a state-level
reward function reads together with overlap tests between sprites named by position."""
import collections

import numpy as np
from moog import action_spaces, game_rules, observers, physics as physics_lib, sprite, tasks
from moog.state_initialization import distributions as distribs

FALL_SPEED = 0.02


def get_config(translucent_occluder):
    physics = physics_lib.Physics(
        (physics_lib.Collision(elasticity=1., symmetric=False, update_angle_vel=False), 'targets', 'walls'),
        updates_per_env_step=10)

    def balls_will_touch(state):
        """Plays the episode forward: True as soon as the balls overlap, False once both have left upwards."""
        first, second = state['targets']
        while True:
            if first.overlaps_sprite(second):
                return True
            if all(ball.y > 1.1 and ball.y_vel > 0 for ball in state['targets']):
                return False
            physics.step(state)

    ball_look = distribs.Product(
        [distribs.Continuous('x', 0.15, 0.85), distribs.Continuous('x_vel', -FALL_SPEED, FALL_SPEED)],
        y_vel=-FALL_SPEED, scale=0.16, shape='circle', opacity=192, c0=255, c1=0, c2=0)
    occluder = sprite.Sprite(x=0.5, y=0.2, shape='square', scale=1., c0=192, c1=192, c2=128,
                             opacity=128 if translucent_occluder else 255)
    walls = [sprite.Sprite(shape=np.array(outline), x=0, y=0, c0=128, c1=128, c2=128)
             for outline in ([[-1, 0.1], [2, 0.1], [2, -1], [-1, -1]], [[0.05, -1], [0.05, 4], [-1, 4], [-1, -1]],
                             [[0.95, -1], [0.95, 4], [2, 4], [2, -1]])]
    box_look = dict(y=0.05, scale=0.12, shape='square', aspect_ratio=0.5, c0=0, c1=0, c2=0)
    answer_boxes = [sprite.Sprite(x=0.4, **box_look), sprite.Sprite(x=0.6, **box_look)]
    dot_look = dict(y=0.05, scale=0.03, shape='circle', c0=255, c1=0, c2=0, opacity=192)
    answer_dots = [sprite.Sprite(x=x, **dot_look) for x in (0.37, 0.43, 0.59, 0.61)]

    def state_initializer():
        token = sprite.Sprite(x=0.5, y=0.05, scale=0.03, shape='spoke_4', c0=255, c1=255, c2=255)
        low_ball = sprite.Sprite(y=1.4, **ball_look.sample())
        high_ball = sprite.Sprite(y=np.random.uniform(1.7, 2.4), **ball_look.sample())
        screen = sprite.Sprite(x=0.5, y=0.5, shape='square', c0=128, c1=128, c2=128)
        state = collections.OrderedDict([
            ('targets', [low_ball, high_ball]), ('occluders', [occluder]), ('walls', walls),
            ('response_boxes', answer_boxes), ('response_tokens', answer_dots), ('agent', [token]), ('screen', [screen])])
        where = [np.copy(ball.position) for ball in state['targets']]      # (the look-ahead moves the balls:
        how_fast = [np.copy(ball.velocity) for ball in state['targets']]   #  they are put back afterwards)
        token.metadata = {'will_contact': balls_will_touch(state)}
        for ball, spot, speed in zip(state['targets'], where, how_fast):
            ball.position = spot
            ball.velocity = speed
        return state

    def answer_value(state):
        token = state['agent'][0]
        if token.overlaps_sprite(state['response_boxes'][0]):     # "they will touch"
            return -1 if token.metadata['will_contact'] else 1
        elif token.overlaps_sprite(state['response_boxes'][1]):   # "they will not"
            return 1 if token.metadata['will_contact'] else -1
        return 0

    task = tasks.CompositeTask(
        tasks.Reset(condition=lambda state: answer_value(state) != 0, reward_fn=answer_value, steps_after_condition=5),
        timeout_steps=1000)
    lift_screen = game_rules.TimedRule(step_interval=(15, 16), rules=(game_rules.VanishByFilter('screen'),))
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.Grid(scaling_factor=0.015, action_layers='agent', control_velocity=True),
        'observers': {'image': observers.PILRenderer(image_size=(64, 64), anti_aliasing=1)},
        'game_rules': (lift_screen,),
    }
