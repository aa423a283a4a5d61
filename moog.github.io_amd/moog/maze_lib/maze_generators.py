"""Random mazes without dead ends or open 2 x 2 blocks (reference: moog/maze_lib/maze_generators.py:96-246).

Host-side numpy generators with the reference's consumption of np.random (randint for the seed cell, a
shuffle of the frontier before every growth step), so a seeded run gives the reference's maze.
"""
import numpy as np

_MAX_ITERS = int(1e5)


def _neighbours(size, cell):
    i, j = cell
    return [c for c in ((i - 1, j), (i + 1, j), (i, j - 1), (i, j + 1)) if 0 <= c[0] < size and 0 <= c[1] < size]


def _blocks_around(size, cell):
    """Top-left corners of the 2 x 2 blocks that contain `cell` (maze_generators.py:36-62)."""
    i, j = cell
    out = []
    for k in (i - 1, i):
        for l in (j - 1, j):
            if 0 <= k < size - 1 and 0 <= l < size - 1:
                out.append((k, l))
    return out


def _fill_dead_ends(maze):
    """Closes, one at a time and always restarting the scan, every open cell with fewer than two open
    neighbours (maze_generators.py:65-93)."""
    size = maze.shape[0]
    changed = True
    while changed:
        changed = False
        for i in range(size):
            for j in range(size):
                if maze[i, j]:
                    continue
                if np.sum([1 - maze[c[0], c[1]] for c in _neighbours(size, (i, j))]) < 2:
                    maze[i, j] = 1
                    changed = True
                    break
            if changed:
                break


def generate_random_maze_matrix(size, ambient_size=None):
    """A connected maze of `size` x `size` cells grown from a random cell (maze_generators.py:96-162), embedded in
    an `ambient_size` matrix of walls when that is larger.  Inside a traced state_initializer the matrix is drawn on
    the device at every reset and a symbolic stand-in is returned (maze_lib/_traced.py)."""
    from .. import _trace
    if _trace.active() is not None:
        from . import _traced
        return _traced.generate(size, ambient_size)
    maze = np.ones((size, size))
    frontier = []   # walls next to open cells

    def open_cell(cell):
        for c in _neighbours(size, cell):
            if maze[c[0], c[1]] and c not in frontier:
                frontier.append(c)
        maze[cell[0], cell[1]] = 0

    open_cell(tuple(np.random.randint(0, size, size=(2,))))
    grew = True
    while grew:
        grew = False
        np.random.shuffle(frontier)
        for cell in frontier:
            if not maze[cell[0], cell[1]]:
                continue
            if any(np.sum(maze[i:i + 2, j:j + 2]) <= 1 for i, j in _blocks_around(size, cell)):
                continue   # opening it would leave a 2 x 2 block without walls
            open_cell(cell)
            grew = True
            break
    _fill_dead_ends(maze)
    if np.sum(1 - maze) == 0:
        return generate_random_maze_matrix(size, ambient_size=ambient_size)
    if ambient_size is not None and ambient_size > size:
        framed = np.ones((ambient_size, ambient_size))
        k = (ambient_size - size) // 2
        framed[k:k + size, k:k + size] = maze
        maze = framed
    return maze


def _grow_blob(maze, num_points):
    graph = maze.get_neighbor_dict()
    blob = [maze.sample_open_point()]
    for _ in range(num_points - 1):
        tries = 0
        while True:
            tries += 1
            root = blob[np.random.randint(len(blob))]
            options = graph[root]
            cand = options[np.random.randint(len(options))]
            if cand and cand not in blob:
                break
            if tries > _MAX_ITERS:
                return False
        blob.append(cand)
    out = np.zeros_like(maze.maze)
    for i, j in blob:
        out[i, j] = 1
    return out


def get_connected_open_blob(maze, num_points):
    """A connected set of `num_points` open cells as a binary matrix (maze_generators.py:226-246)."""
    for _ in range(_MAX_ITERS):
        blob = _grow_blob(maze, num_points)
        if not isinstance(blob, bool):
            return blob
    raise ValueError('Could not generate an open connected blob.')
