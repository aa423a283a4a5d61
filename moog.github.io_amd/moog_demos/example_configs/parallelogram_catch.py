"""Parallelogram catch: four pellets tethered into a rigid parallelogram of random orientation and aspect drift (or
rest) in a first-person view; the agent, carrying an annulus-shaped window, has to touch all four.
Parameters: reference moog_demos/example_configs/parallelogram_catch.py:30-196 (levels 0 / 1 / 2: pellet speed up to
0 / 0.01 / 0.02 per step).

The initializer draws from np.random directly (orientation, axis ratio, every pellet's velocity) and builds the
pellets' polygon and positions from those draws with numpy arithmetic: on the engine these become reset-time
expressions evaluated per env (moog/_trace.py `hdraw`, moog/_symbolic.py `SymMat`)."""
import collections

import numpy as np
from moog import action_spaces, game_rules, observers, physics as physics_lib, shapes, sprite, tasks

CELL = 0.3   # background grid pitch
MAX_SPEED = (0., 0.01, 0.02)


def random_parallelogram(min_axis_ratio):
    """Corners of a parallelogram inscribed in the unit circle: two diameters, the second one shortened."""
    turn = np.random.uniform(0, 2)
    corner_angles = np.pi * (np.array([0., 0.5, 1., 1.5]) + turn)
    corners = np.stack((np.sin(corner_angles), np.cos(corner_angles)), axis=1)
    ratio = np.random.uniform(min_axis_ratio, 1.)
    corners *= np.array([[1.], [ratio], [1.], [ratio]])
    return corners


def get_config(level):
    max_vel = MAX_SPEED[level]
    grid = shapes.grid_lines(grid_x=CELL, grid_y=CELL, buffer_border=1., c0=0., c1=0., c2=0.5)

    def state_initializer():
        agent = sprite.Sprite(x=0.5, y=0.5, shape='circle', scale=0.04, c0=0.33, c1=1., c2=0.66)
        window = sprite.Sprite(x=0.5, y=0.5, shape=shapes.annulus_vertices(0.15, 2.), scale=1., c0=0.6, c1=1., c2=1.)
        corners = random_parallelogram(min_axis_ratio=0.5)
        pellet_shape = 0.075 * corners
        centres = 0.4 * corners
        centres += np.array([0.5, 0.5]) - centres[0]   # the first pellet starts under the agent
        pellets = [
            sprite.Sprite(x=c[0], y=c[1], shape=pellet_shape, scale=1.,
                          x_vel=np.random.uniform(-1 * max_vel, max_vel), y_vel=np.random.uniform(-1 * max_vel, max_vel),
                          c0=0.2, c1=1., c2=1.)
            for c in centres]
        return collections.OrderedDict([('grid', grid), ('prey', pellets), ('agent', [agent]), ('agent_annulus', [window])])

    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.25), ['agent', 'agent_annulus']),
        updates_per_env_step=10,
        corrective_physics=physics_lib.Tether(('prey',), update_angle_vel=True))

    task = tasks.CompositeTask(
        tasks.ContactReward(1, layers_0='agent', layers_1='prey', condition=lambda s_agent, s_prey: s_prey.c1 > 0.5),
        tasks.Reset(condition=lambda state: all([s.c1 < 0.5 for s in state['prey']]), steps_after_condition=10),
        timeout_steps=500)

    def fade(pellet):   # a caught pellet turns grey
        pellet.c1 = 0.
        pellet.c2 = 0.6

    rules = (
        game_rules.ModifyOnContact(layers_0='agent', layers_1='prey', modifier_1=fade),
        game_rules.KeepNearCenter(agent_layer='agent', layers_to_center=['agent_annulus', 'prey'], grid_x=CELL),
    )
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.Joystick(scaling_factor=0.01, action_layers=('agent', 'agent_annulus')),
        'observers': {'image': observers.PILRenderer(
            image_size=(64, 64), anti_aliasing=1, color_to_rgb='hsv_to_rgb',
            polygon_modifier=observers.polygon_modifiers.FirstPersonAgent(agent_layer='agent'))},
        'game_rules': rules,
    }
