"""Coverage recipe (not a reference task): initializers that play the physics forward before the episode starts, in
shapes other than the two reference configs that do so (bounce_box_contact_prediction.py:40-50, red_green.py:92-116) --
pinned by golden vectors captured from the reference (tests/golden/lookahead_zoo_*.npz).

level 0: a `while True` look-ahead with THREE exits (the puck reaches the left pocket / the right pocket / comes to
    rest near the floor) whose outcome (0 / 1 / 2) goes to the cue's metadata and decides the reward of a state-level
    reward function; gravity and drag make the look-ahead a different physics than pure bouncing.
level 1: a `for step in range(60)` look-ahead that REJECTS the trial (the initializer starts over) when the puck reaches
    a pocket before step 8 or not at all, behind a fail_gracefully generator whose short result also starts over; the
    pockets are built OUTSIDE the initializer and a rule dims the one that was hit, so the dimming outlives the reset."""
import collections

import numpy as np
from moog import action_spaces, game_rules, observers, physics as physics_lib, sprite, tasks
from moog.state_initialization import distributions as distribs
from moog.state_initialization import sprite_generators


def get_config(level):
    bounce = physics_lib.Collision(elasticity=0.9, symmetric=False, update_angle_vel=False)
    if level == 0:
        physics = physics_lib.Physics(
            (physics_lib.DownGravity(g=-0.0008), 'puck'), (physics_lib.Drag(coeff_friction=0.02), 'puck'),
            (bounce, 'puck', 'walls'), updates_per_env_step=4)
    else:
        physics = physics_lib.Physics((bounce, 'puck', ['walls', 'blocks']), updates_per_env_step=5)
    walls = [sprite.Sprite(shape=np.array(v), x=0, y=0, c0=0.6, c1=0.2, c2=0.5)
             for v in ([[-1, 0.08], [2, 0.08], [2, -1], [-1, -1]], [[-1, 0.95], [2, 0.95], [2, 2], [-1, 2]],
                       [[0.05, -1], [0.05, 3], [-1, 3], [-1, -1]], [[0.95, -1], [0.95, 3], [2, 3], [2, -1]])]
    pockets = [sprite.Sprite(x=0.2, y=0.3, shape='square', scale=0.16, c0=0.0, c1=0.9, c2=0.9),
               sprite.Sprite(x=0.8, y=0.3, shape='square', scale=0.16, c0=0.33, c1=0.9, c2=0.9)]

    def where_it_ends(state):
        puck = state['puck'][0]
        while True:
            if puck.overlaps_sprite(state['pockets'][0]):
                return 0
            if puck.overlaps_sprite(state['pockets'][1]):
                return 1
            if puck.y < 0.2 and np.abs(puck.y_vel) < 0.004:
                return 2
            physics.step(state)

    def first_pocket(state):
        puck = state['puck'][0]
        for step in range(60):
            left = puck.overlaps_sprite(state['pockets'][0])
            right = puck.overlaps_sprite(state['pockets'][1])
            if left or right:
                if step < 8:
                    return None
                return 1 if right else 0
            physics.step(state)
        return None

    puck_look = distribs.Product(
        [distribs.Continuous('x', 0.3, 0.7), distribs.Continuous('y', 0.5, 0.85),
         distribs.Continuous('x_vel', -0.03, 0.03), distribs.Continuous('y_vel', -0.02, 0.02)],
        shape='circle', scale=0.07, c0=0.6, c1=1., c2=1.)
    make_blocks = sprite_generators.generate_sprites(
        distribs.Product([distribs.Continuous('x', 0.25, 0.75), distribs.Continuous('y', 0.45, 0.9)],
                         shape='square', scale=0.13, c0=0.1, c1=0.3, c2=0.6),
        num_sprites=3, max_recursion_depth=6, fail_gracefully=True)

    def initializer_0():
        puck = sprite.Sprite(**puck_look.sample())
        cue = sprite.Sprite(x=0.5, y=0.04, shape='triangle', scale=0.05, c0=0.15, c1=1., c2=1.)
        agent = sprite.Sprite(x=0.5, y=0.5, shape='spoke_4', scale=0.04, c0=0., c1=0., c2=1.)
        state = collections.OrderedDict([('walls', walls), ('pockets', pockets), ('puck', [puck]), ('cue', [cue]),
                                         ('agent', [agent])])
        at, going = np.copy(puck.position), np.copy(puck.velocity)
        cue.metadata = {'ends': where_it_ends(state)}
        puck.position = at
        puck.velocity = going
        return state

    def initializer_1():
        blocks = make_blocks(disjoint=True)
        if len(blocks) < 3:
            return initializer_1()
        puck = sprite.Sprite(**puck_look.sample())
        cue = sprite.Sprite(x=0.5, y=0.04, shape='triangle', scale=0.05, c0=0.15, c1=1., c2=1.)
        agent = sprite.Sprite(x=0.5, y=0.5, shape='spoke_4', scale=0.04, c0=0., c1=0., c2=1.)
        state = collections.OrderedDict([('walls', walls), ('blocks', blocks), ('pockets', pockets), ('puck', [puck]),
                                         ('cue', [cue]), ('agent', [agent])])
        at, going = np.copy(puck.position), np.copy(puck.velocity)
        found = first_pocket(state)
        if found is None:
            return initializer_1()
        puck.position = at
        puck.velocity = going
        cue.metadata = {'ends': found}
        return state

    def verdict(state):
        agent, told = state['agent'][0], state['cue'][0].metadata['ends']
        if agent.overlaps_sprite(state['pockets'][0]):
            return 2 if told == 0 else -1
        if agent.overlaps_sprite(state['pockets'][1]):
            return 3 if told == 1 else -1
        if agent.y < 0.15:
            return 1 if told == 2 else -0.5
        return 0

    def dim(s):
        s.c2 = s.c2 * 0.8

    rules = (game_rules.ModifyOnContact(layers_0='pockets', layers_1='puck', modifier_0=dim),)
    task = tasks.CompositeTask(
        tasks.Reset(condition=lambda state: verdict(state) != 0, reward_fn=verdict, steps_after_condition=3),
        timeout_steps=45)
    return {
        'state_initializer': initializer_0 if level == 0 else initializer_1,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.Joystick(scaling_factor=0.02, action_layers='agent'),
        'observers': {'image': observers.PILRenderer(image_size=(64, 64), color_to_rgb='hsv_to_rgb')},
        'game_rules': rules,
    }
