"""Maze helpers (reference: moog/maze_lib/__init__.py:3-5, maze.py, maze_generators.py).

`Maze` mediates between a binary wall matrix and wall sprites; `MazePhysics` and the maze walks of
`moog.physics` read the matrix that `Maze.from_state` infers from a wall layer.  On the engine the
inference happens once, when a config is lowered (the wall layer must consist of constant sprites), and
the matrix travels in `moog_program_t.maze`.
"""
from .maze import Maze
from .maze_generators import generate_random_maze_matrix
from .maze_generators import get_connected_open_blob
