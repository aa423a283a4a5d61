"""Maze: a square binary matrix (1 = wall) and its wall sprites (reference: moog/maze_lib/maze.py:20-261)."""
import numpy as np

from .. import sprite as sprite_lib
from . import _traced

_EPSILON = 1e-4       # maze.py:12: tolerance when testing wall vertices against multiples of 1 / N
_MAX_MAZE_SIZE = 100  # maze.py:16


def world_vertices(sp):
    """World vertices of a constant sprite recipe: scale -> rotate -> translate of the centred shape, with the
    raw shape's centroid added to the position (sprite.py:329-424)."""
    from .. import shapes
    f = sp.factors
    if any(isinstance(v, sprite_lib.SymbolicFactor) for v in f.values()):
        raise NotImplementedError('maze walls must be constant sprites (their factors are sampled)')
    raw = shapes.SHAPES[f['shape']] if isinstance(f['shape'], str) else np.asarray(f['shape'], dtype=np.float64)
    centred, centroid, _ = sprite_lib.shape_record(raw)
    sx, sy = f['scale'], f['scale'] * f['aspect_ratio']
    c, s = np.cos(f['angle']), np.sin(f['angle'])
    x = c * sx * centred[:, 0] - s * sy * centred[:, 1] + f['x'] + centroid[0]
    y = s * sx * centred[:, 0] + c * sy * centred[:, 1] + f['y'] + centroid[1]
    return np.stack([x, y], axis=1)


def _inside(poly, pts):
    """Even-odd containment of points [m, 2] in a polygon [n, 2] (matplotlib point_in_path_impl)."""
    inside = np.zeros(len(pts), dtype=bool)
    n = len(poly)
    for i in range(n):
        x0, y0 = poly[i]
        x1, y1 = poly[(i + 1) % n]
        f0, f1 = y0 >= pts[:, 1], y1 >= pts[:, 1]
        cross = ((y1 - pts[:, 1]) * (x0 - x1) >= (x1 - pts[:, 0]) * (y0 - y1)) == f1
        inside ^= (f0 != f1) & cross
    return inside


class Maze(object):
    def __init__(self, maze):
        """maze: square array of 0 / 1 (or bool); the ones are walls (maze.py:23-36)."""
        self.maze = maze
        self.maze_size = maze.shape[0]
        self.grid_side = 1. / self.maze_size
        self.half_grid_side = 0.5 * self.grid_side
        self.side_vertices = np.linspace(self.half_grid_side, 1. - self.half_grid_side, self.maze_size)
        if isinstance(maze, _traced.TracedMatrix):   # a maze drawn per reset on the device (pacman.py:40-42)
            _traced.bind(self)

    @classmethod
    def from_state(cls, state, maze_layer='walls'):
        """The maze of a wall layer (maze.py:39-84): the smallest N such that every wall vertex is a multiple
        of 1 / N, then the cells whose centre lies in a wall sprite."""
        walls = [world_vertices(s) if isinstance(s, sprite_lib.Sprite) else np.asarray(s.vertices)
                 for s in state[maze_layer]]
        allv = np.concatenate(walls) if walls else np.zeros((0, 2))
        size = 1
        while not np.allclose(np.round(allv * size) / size, allv, atol=_EPSILON):
            size += 1
            if size > _MAX_MAZE_SIZE:
                raise ValueError('Cannot find a maze grid size. Your maze sprites are invalid.')
        half = 1. / (2 * size)
        centres = np.linspace(half, 1 - half, size)
        gx, gy = np.meshgrid(centres, centres)
        pts = np.stack([gx.ravel(), gy.ravel()], axis=1)
        hit = np.zeros(size * size, dtype=bool)
        for w in walls:
            hit |= _inside(w, pts)
        return cls(hit.reshape(size, size).astype(int))

    def to_sprites(self, **color):
        """One square sprite per wall cell, columns outer, rows inner (maze.py:86-105)."""
        if isinstance(self.maze, _traced.TracedMatrix):
            return _traced.wall_sprites(self, color)
        n = self.maze_size
        v = np.linspace(0., 1., n + 1)
        out = []
        for x in range(n):
            for y in range(n):
                if self.maze[y, x]:
                    square = np.array([[v[x], v[y]], [v[x], v[y + 1]], [v[x + 1], v[y + 1]], [v[x + 1], v[y]]])
                    out.append(sprite_lib.Sprite(x=0., y=0., shape=square, **color))
        return out

    def open_vertex(self, i, j):
        """Cell (i, j) is inside the matrix and not a wall (maze.py:107-112; the matrix is indexed [j, i])."""
        if i < 0 or j < 0 or i >= self.maze_size or j >= self.maze_size:
            return False
        return not self.maze[j, i]

    def valid_directions(self, i, j):
        """[[west, east], [south, north]] openness of the neighbours of (i, j) (maze.py:114-120)."""
        return np.array([[self.open_vertex(k, j) for k in (i - 1, i + 1)],
                         [self.open_vertex(i, k) for k in (j - 1, j + 1)]])

    def sample_random_position(self, off_intersection=True):
        """A random point on the open edges of the maze grid (maze.py:122-146)."""
        free = 1 - self.maze
        v_edges = np.stack(np.nonzero(np.logical_and(free[1:], free[:-1]))[::-1]).T
        h_edges = np.stack(np.nonzero(np.logical_and(free[:, 1:], free[:, :-1]))[::-1]).T
        n_h, n_v = len(h_edges), len(v_edges)
        if np.random.rand() < float(n_h) / (n_h + n_v):
            position = h_edges[np.random.choice(n_h)]
            if off_intersection:
                position = position + np.random.rand() * np.array([1., 0.])
        else:
            position = v_edges[np.random.choice(n_v)]
            if off_intersection:
                position = position + np.random.rand() * np.array([0., 1.])
        return self.half_grid_side + position * self.grid_side

    def to_background_grid(self, line_thickness=0.01, **color):
        """Thin line sprites along the corridors' centre lines (maze.py:148-180): vertical ones first."""
        out = []

        def bar(x0, x1, y0, y1):
            out.append(sprite_lib.Sprite(x=0., y=0., shape=np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1]]), **color))
        for x in self.side_vertices:
            bar(x - 0.5 * line_thickness, x + 0.5 * line_thickness, 0., 1.)
        for y in self.side_vertices:
            bar(0., 1., y - 0.5 * line_thickness, y + 0.5 * line_thickness)
        return out

    def sample_open_point(self):
        """(i, j) of a uniformly drawn open cell (maze.py:182-192)."""
        if np.sum(1 - self.maze) == 0:
            raise ValueError('Maze has no open point.')
        cells = np.argwhere(self.maze == 0)
        return tuple(cells[np.random.randint(len(cells))])

    def sample_distinct_open_points(self, num_points):
        """`num_points` different open cells (maze.py:194-208)."""
        if isinstance(self.maze, _traced.TracedMatrix):
            return _traced.sample_points(self, num_points)
        if np.sum(1 - self.maze) < num_points:
            raise ValueError('Maze has no open point.')
        cells = np.argwhere(self.maze == 0)
        return [tuple(cells[i]) for i in np.random.choice(len(cells), size=num_points, replace=False)]

    def get_neighbors(self, i, j):
        """Open cells next to matrix entry (i, j), in the order up, down, left, right (maze.py:210-221)."""
        out = []
        if i > 0 and not self.maze[i - 1, j]:
            out.append((i - 1, j))
        if i < self.maze_size - 1 and not self.maze[i + 1, j]:
            out.append((i + 1, j))
        if j > 0 and not self.maze[i, j - 1]:
            out.append((i, j - 1))
        if j < self.maze_size - 1 and not self.maze[i, j + 1]:
            out.append((i, j + 1))
        return out

    def get_neighbor_dict(self):
        return {(i, j): self.get_neighbors(i, j) for i in range(self.maze_size) for j in range(self.maze_size)}

    def add_wall(self, x_range, y_range):
        """maze.py:237-244"""
        self.maze[x_range[0]:x_range[1] + 1, y_range[0]:y_range[1] + 1] = 1

    def add_outer_walls(self):
        """maze.py:246-261"""
        self.maze[0, :] = 1
        self.maze[-1, :] = 1
        self.maze[:, 0] = 1
        self.maze[:, -1] = 1
