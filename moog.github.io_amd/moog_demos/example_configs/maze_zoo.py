"""Coverage recipe (not a reference task): the maze components of the reference's pacman task on fixed mazes --
`maze_lib.Maze` (wall sprites from a binary matrix, reference moog/maze_lib/maze.py:20-120), `MazePhysics`
(maze_physics.py:18-211: avatars stay on the corridor grid, turn at intersections, rotate with their heading)
and `RandomMazeWalk` (maze_walk.py:96-193: ghosts wander without backtracking) -- pinned by golden vectors
captured from the reference (tests/golden/maze_zoo_*.npz).  Parameters follow
moog_demos/example_configs/pacman.py:23-160 where they exist there.
level 0: 8 x 8 maze, constant_speed (pacman's setting);  level 1: 10 x 10 maze, max_speed instead, ghosts
that may turn back at walls;  level 2: the 8 x 8 maze with `DeterministicMazeWalk` ghosts (maze_walk.py:203-243: a list of
prescribed velocities read front to back by whichever ghost reaches an intersection next, never rewound -- across the
recording's resets too)."""
import collections

import numpy as np
from moog import action_spaces, game_rules, maze_lib, observers, physics as physics_lib, sprite, tasks
from moog.state_initialization import distributions as distribs
from moog.state_initialization import sprite_generators

MAZES = (
    np.array([[1, 1, 1, 1, 1, 1, 1, 1],
              [1, 0, 0, 0, 0, 0, 0, 1],
              [1, 0, 1, 1, 0, 1, 0, 1],
              [1, 0, 1, 0, 0, 1, 0, 1],
              [1, 0, 0, 0, 1, 1, 0, 1],
              [1, 0, 1, 0, 0, 0, 0, 1],
              [1, 0, 0, 0, 1, 1, 0, 1],
              [1, 1, 1, 1, 1, 1, 1, 1]]),
    np.array([[1, 1, 1, 1, 1, 1, 1, 1, 1, 1],
              [1, 0, 0, 0, 0, 1, 0, 0, 0, 1],
              [1, 0, 1, 1, 0, 1, 0, 1, 0, 1],
              [1, 0, 1, 0, 0, 0, 0, 1, 0, 1],
              [1, 0, 0, 0, 1, 1, 0, 0, 0, 1],
              [1, 1, 1, 0, 1, 1, 0, 1, 1, 1],
              [1, 0, 0, 0, 0, 0, 0, 0, 0, 1],
              [1, 0, 1, 1, 0, 1, 1, 1, 0, 1],
              [1, 0, 0, 0, 0, 0, 0, 0, 0, 1],
              [1, 1, 1, 1, 1, 1, 1, 1, 1, 1]]),
)


def get_config(level):
    maze = maze_lib.Maze(MAZES[1 if level == 1 else 0])
    speed = 0.03 if level == 1 else 0.02
    walls = maze.to_sprites(c0=0., c1=0., c2=0.8)
    cells = [maze.grid_side * (0.5 + np.array(p)) for p in np.argwhere(maze.maze == 0)]   # (row, column) -> (y, x)
    ghost_cells = cells[len(cells) // 2:]
    ghost_factors = distribs.Mixture(
        [distribs.Product([], x=c[1], y=c[0], shape='circle', scale=0.05, c0=0., c1=1., c2=0.8) for c in ghost_cells])
    make_ghosts = sprite_generators.generate_sprites(ghost_factors, num_sprites=2)

    def state_initializer():
        agent = sprite.Sprite(x=cells[0][1], y=cells[0][0], shape='triangle', scale=0.05, c0=0.33, c1=1., c2=0.66)
        # (a fresh list every episode: VanishOnContact pops from it)
        prey = [sprite.Sprite(x=c[1], y=c[0], shape='circle', scale=0.025, c0=0.2, c1=1., c2=1.) for c in cells[3:]]
        return collections.OrderedDict(
            [('walls', walls), ('prey', prey), ('ghosts', make_ghosts()), ('agent', [agent])])

    if level == 2:   # a fixed itinerary: more entries than the recording consumes in its first episodes, fewer than in all
        rs = np.random.RandomState(5)
        dirs = [(speed, 0.), (0., speed), (-speed, 0.), (0., -speed), (0., 0.), (speed, speed)]
        walk = physics_lib.DeterministicMazeWalk(speed=speed, step_velocities=[dirs[k] for k in rs.randint(0, 6, size=70)])
    else:
        walk = (physics_lib.RandomMazeWalk(speed=speed) if level == 0 else
                physics_lib.RandomMazeWalk(speed=speed, allow_wall_backtracking=True, only_turn_at_wall=True))
    maze_physics = (physics_lib.MazePhysics(maze_layer='walls', avatar_layers=('agent', 'ghosts'), constant_speed=speed)
                    if level != 1 else
                    physics_lib.MazePhysics(maze_layer='walls', avatar_layers=('agent', 'ghosts'), max_speed=0.8 * speed))
    physics = physics_lib.Physics((walk, ['ghosts']), updates_per_env_step=1, corrective_physics=[maze_physics])
    task = tasks.CompositeTask(
        tasks.ContactReward(-5, layers_0='agent', layers_1='ghosts', reset_steps_after_contact=0),
        tasks.ContactReward(1, layers_0='agent', layers_1='prey'),
        tasks.Reset(condition=lambda state: len(state['prey']) == 0, steps_after_condition=5),
        timeout_steps=70)
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.Grid(scaling_factor=speed, action_layers='agent', control_velocity=True,
                                           momentum=0.5),
        'observers': {'image': observers.PILRenderer(image_size=(128, 128), anti_aliasing=1, color_to_rgb='hsv_to_rgb')},
        'game_rules': (game_rules.VanishOnContact(vanishing_layer='prey', contacting_layer='agent'),),
    }
