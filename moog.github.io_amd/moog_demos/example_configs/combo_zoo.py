"""Coverage recipe (not a reference task): features of round 3 in COMBINATION -- pinned by golden vectors captured from
the reference (tests/golden/combo_zoo_*.npz).

An initializer plays 25 steps of physics forward on two star-shaped pucks that spin, collide with each other
(symmetric, with angular velocity) and are kicked by a RandomForce -- so the look-ahead itself draws from np.random inside
the reset, and leaves the pucks ROTATED and spinning differently when it puts only their positions and velocities back,
as the reference does; the outcome (which puck is higher after the look-ahead, or a tie band) goes to the agent's
metadata; a trial whose pucks touch within the first 3 steps is rejected.  The goal posts are built outside the
initializer and a rule recolours the one that was hit (it stays recoloured over resets).  The frame is 50 x 37 pixels
(no multiple of 16) and the recordings are made with float32 actions."""
import collections

import numpy as np
from moog import action_spaces, game_rules, observers, physics as physics_lib, sprite, tasks
from moog.state_initialization import distributions as distribs


def get_config(_=0):
    clash = physics_lib.Collision(elasticity=0.8, symmetric=True, update_angle_vel=True)
    bounce = physics_lib.Collision(elasticity=1., symmetric=False, update_angle_vel=True)
    physics = physics_lib.Physics(
        (physics_lib.RandomForce(max_force_magnitude=0.004), 'pucks'),
        (physics_lib.Drag(coeff_friction=0.03), 'pucks'),
        (clash, 'pucks', 'pucks'), (bounce, 'pucks', 'walls'),
        updates_per_env_step=3)
    walls = [sprite.Sprite(shape=np.array(v), x=0, y=0, c0=0.6, c1=0.3, c2=0.4)
             for v in ([[-1, 0.05], [2, 0.05], [2, -1], [-1, -1]], [[-1, 0.95], [2, 0.95], [2, 2], [-1, 2]],
                       [[0.05, -1], [0.05, 3], [-1, 3], [-1, -1]], [[0.95, -1], [0.95, 3], [2, 3], [2, -1]])]
    posts = [sprite.Sprite(x=0.12, y=0.5, shape='square', scale=0.1, c0=0.0, c1=0.8, c2=1.),
             sprite.Sprite(x=0.88, y=0.5, shape='square', scale=0.1, c0=0.33, c1=0.8, c2=1.)]
    puck_look = distribs.Product(
        [distribs.Continuous('x', 0.3, 0.7), distribs.Continuous('y', 0.3, 0.7),
         distribs.Continuous('x_vel', -0.04, 0.04), distribs.Continuous('y_vel', -0.04, 0.04),
         distribs.Continuous('angle_vel', -0.3, 0.3), distribs.Continuous('c0', 0.5, 0.9)],
        shape='star_5', scale=0.13, c1=1., c2=1.)

    def look_ahead(state):
        first, second = state['pucks']
        for step in range(25):
            if first.overlaps_sprite(second) and step < 3:
                return None
            physics.step(state)
        if first.y > second.y + 0.05:
            return 0
        if second.y > first.y + 0.05:
            return 1
        return 2

    def state_initializer():
        pucks = [sprite.Sprite(**puck_look.sample()), sprite.Sprite(**puck_look.sample())]
        agent = sprite.Sprite(x=0.5, y=0.12, shape='triangle', scale=0.06, c0=0.15, c1=1., c2=1.)
        state = collections.OrderedDict([('walls', walls), ('posts', posts), ('pucks', pucks), ('agent', [agent])])
        places = [np.copy(p.position) for p in pucks]
        speeds = [np.copy(p.velocity) for p in pucks]
        outcome = look_ahead(state)
        if outcome is None:
            return state_initializer()
        for p, where, how in zip(pucks, places, speeds):
            p.position = where
            p.velocity = how
        agent.metadata = {'higher': outcome}
        return state

    def verdict(state):
        agent, told = state['agent'][0], state['agent'][0].metadata['higher']
        if agent.overlaps_sprite(state['posts'][0]):
            return 1 if told == 0 else -1
        if agent.overlaps_sprite(state['posts'][1]):
            return 1 if told == 1 else -1
        if agent.y > 0.85:
            return 2 if told == 2 else -2
        return 0

    def recolour(s):
        s.c0 = s.c0 + 0.07
        s.c1 = 0.5

    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': tasks.CompositeTask(
            tasks.Reset(condition=lambda state: verdict(state) != 0, reward_fn=verdict, steps_after_condition=2),
            timeout_steps=40),
        'action_space': action_spaces.Joystick(scaling_factor=0.02, action_layers='agent'),
        'observers': {'image': observers.PILRenderer(image_size=(50, 37), color_to_rgb='hsv_to_rgb')},
        'game_rules': (game_rules.ModifyOnContact(layers_0='posts', layers_1='agent', modifier_0=recolour),),
    }
