"""Pac-Man: eat every pellet of a maze that is redrawn each episode while ghosts wander it at random.
Parameters: reference moog_demos/example_configs/pacman.py:21-176 (level 0: two ghosts in an 8 x 8 maze, level 1:
three ghosts in 10 x 10; both inside a 12 x 12 frame of walls, rendered at 256 x 256).

Everything moves on the corridor grid at one speed (MazePhysics, constant_speed).  Ghosts and pellets are created with
infinite / unit mass and the ghosts only start walking once the agent has moved (the `unglue` rule gives them mass 1:
RandomMazeWalk skips sprites of infinite mass, maze_walk.py:160-161).
"""
import collections

import numpy as np
from moog import action_spaces, game_rules, maze_lib, observers, physics as physics_lib, sprite, tasks

SPEED = 0.015
AMBIENT = 12
LEVELS = ((2, 8), (3, 10))   # (ghosts, maze size)


def _cell_centre(maze, cell):
    """(y, x) of the centre of maze cell (row, column)."""
    return maze.grid_side * (0.5 + np.array(cell))


def get_config(level):
    n_ghosts, maze_size = LEVELS[level]

    def state_initializer():
        matrix = maze_lib.generate_random_maze_matrix(size=maze_size, ambient_size=AMBIENT)
        maze = maze_lib.Maze(np.flip(matrix, axis=0))
        walls = maze.to_sprites(c0=0., c1=0., c2=0.8)
        starts = [_cell_centre(maze, cell) for cell in maze.sample_distinct_open_points(1 + n_ghosts)]
        agent = sprite.Sprite(x=starts[0][1], y=starts[0][0], shape='circle', scale=0.05, c0=0.33, c1=1., c2=0.66)
        ghosts = [sprite.Sprite(x=p[1], y=p[0], shape='circle', scale=0.05, mass=np.inf, c0=0., c1=1., c2=0.8)
                  for p in starts[1:]]
        pellets = []
        for cell in np.argwhere(maze.maze == 0):   # one pellet on every open cell
            p = _cell_centre(maze, cell)
            pellets.append(sprite.Sprite(x=p[1], y=p[0], shape='circle', scale=0.025, c0=0.2, c1=1., c2=1.))
        return collections.OrderedDict([('walls', walls), ('prey', pellets), ('ghosts', ghosts), ('agent', [agent])])

    physics = physics_lib.Physics(
        (physics_lib.RandomMazeWalk(speed=SPEED), ['ghosts']),
        updates_per_env_step=1,
        corrective_physics=[physics_lib.MazePhysics(
            maze_layer='walls', avatar_layers=('agent', 'prey', 'ghosts'), constant_speed=SPEED)])

    task = tasks.CompositeTask(
        tasks.ContactReward(-5, layers_0='agent', layers_1='ghosts', reset_steps_after_contact=0),
        tasks.ContactReward(1, layers_0='agent', layers_1='prey'),
        tasks.Reset(condition=lambda state: len(state['prey']) == 0, steps_after_condition=5),
        timeout_steps=1000)

    def set_unit_mass(s):
        s.mass = 1.

    def agent_has_moved(state):
        return not np.all(state['agent'][0].velocity == 0)

    rules = (
        game_rules.VanishOnContact(vanishing_layer='prey', contacting_layer='agent'),
        game_rules.ConditionalRule(condition=agent_has_moved,
                                   rules=game_rules.ModifySprites(('prey', 'ghosts'), set_unit_mass)),
    )
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        # (with constant_speed the momentum value plays no role, pacman.py:106-111)
        'action_space': action_spaces.Grid(scaling_factor=SPEED, action_layers='agent', control_velocity=True,
                                           momentum=0.5),
        'observers': {'image': observers.PILRenderer(image_size=(256, 256), anti_aliasing=1,
                                                     color_to_rgb='hsv_to_rgb')},
        'game_rules': rules,
    }
