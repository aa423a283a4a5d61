"""Symbolic stand-ins for a maze that is drawn anew at every reset (reference: pacman.py:39-72).

Inside a traced `state_initializer` (moog/_trace.py) `generate_random_maze_matrix` cannot return numbers: the
matrix differs per env and per episode and is drawn on the device.  It returns a `TracedMatrix` instead; the few
numpy idioms the reference's configs apply to it are intercepted:

    np.flip(matrix, axis=0)              -> the same matrix, flagged as flipped         (pacman.py:42)
    Maze(matrix).to_sprites(**color)     -> one placeholder sprite per possible wall     (maze.py:86-112)
    maze.sample_distinct_open_points(k)  -> k traced cells                               (maze.py:200-214)
    np.argwhere(maze.maze == 0)          -> one traced cell per possible open cell       (pacman.py:62)
    scalar arithmetic on np.array(cell)  -> tabulated for every index of the axis        (pacman.py:47,64)

A traced cell is a (row, column) pair of `CellIndex` factors.  Arithmetic on a `CellIndex` composes a host
function of the integer index; the lowering evaluates it for every index 0 .. N-1 with numpy's own scalar
arithmetic and ships the table, so the device reproduces the reference's value bit for bit whatever the
expression was.
"""
import numpy as np

from .. import _abi
from .. import _trace
from .. import sprite as sprite_lib


class MazeOp(_trace.GenOp):
    """A generation op tied to the traced maze: `cell` = (MOOG_CELL_*, argument)."""

    def __init__(self, cell, sprites):
        _trace.GenOp.__init__(self, None, len(sprites), len(sprites), False, [], 0, sprites)
        self.cell = cell


class CellIndex(sprite_lib.SymbolicFactor):
    """Row (axis 0) or column (axis 1) index of a traced cell, possibly under scalar arithmetic."""

    def __init__(self, cell, axis, fn=None):
        sprite_lib.SymbolicFactor.__init__(self, None)
        self.cell, self.axis, self.fn = cell, axis, fn

    def table(self, n):
        out = []
        for v in range(n):
            x = np.int64(v)   # np.array(tuple_of_ints) is an int64 array in the reference
            out.append(float(x if self.fn is None else self.fn(x)))
        return out

    def _then(self, g):
        f = self.fn
        return CellIndex(self.cell, self.axis, g if f is None else (lambda v: g(f(v))))

    @staticmethod
    def _scalar(o):
        if isinstance(o, (int, float, np.integer, np.floating)):
            return o
        raise NotImplementedError('arithmetic between a maze cell index and %r' % (type(o).__name__,))

    def __add__(self, o):
        o = self._scalar(o)
        return self._then(lambda v: v + o)

    def __radd__(self, o):
        o = self._scalar(o)
        return self._then(lambda v: o + v)

    def __sub__(self, o):
        o = self._scalar(o)
        return self._then(lambda v: v - o)

    def __rsub__(self, o):
        o = self._scalar(o)
        return self._then(lambda v: o - v)

    def __mul__(self, o):
        o = self._scalar(o)
        return self._then(lambda v: v * o)

    def __rmul__(self, o):
        o = self._scalar(o)
        return self._then(lambda v: o * v)

    def __truediv__(self, o):
        o = self._scalar(o)
        return self._then(lambda v: v / o)

    def __neg__(self):
        return self._then(lambda v: -v)

    def __bool__(self):
        raise NotImplementedError('branching on a maze cell index inside a state_initializer')


class CellShape(sprite_lib.SymbolicFactor):
    """The wall square of a traced cell (maze.py:104-109)."""

    def __init__(self, cell):
        sprite_lib.SymbolicFactor.__init__(self, None)
        self.cell = cell


def traced_cell(sel, arg):
    """(row, column) of a traced cell: a plain tuple, so that `np.array(cell)` is a 2-element object array."""
    cell = (sel, arg)
    return (CellIndex(cell, 0), CellIndex(cell, 1))


class _OpenMask(object):
    """`maze.maze == 0`"""

    def __init__(self, matrix):
        self.matrix = matrix

    def __array_function__(self, func, types, args, kwargs):
        if func is np.argwhere and len(args) == 1 and not kwargs:
            return [traced_cell(_abi.MOOG_CELL_OPEN_RANK, k) for k in range(self.matrix.max_open)]
        raise NotImplementedError('np.%s on the open cells of a traced maze' % func.__name__)


class TracedMatrix(object):
    """The matrix `generate_random_maze_matrix(size, ambient_size)` draws at every reset."""

    def __init__(self, gen_size, ambient_size, flipped=False):
        self.gen_size = int(gen_size)
        self.ambient = int(ambient_size if ambient_size is not None and ambient_size > gen_size else gen_size)
        self.flipped = flipped
        self.shape = (self.ambient, self.ambient)
        if self.gen_size > _abi.MOOG_MAX_MAZE_GEN or self.ambient > _abi.MOOG_MAX_MAZE:
            raise NotImplementedError('random mazes beyond size %d in %d' % (_abi.MOOG_MAX_MAZE_GEN, _abi.MOOG_MAX_MAZE))

    # generate_random_maze_matrix leaves no fully open 2 x 2 block (one wall per disjoint block at least), no dead
    # end, and is never empty: every open cell lies on a cycle, and a cycle without an open 2 x 2 block has at
    # least 8 cells (maze_generators.py:99-103).  More than the bound is a fault on the device, not a silent loss.
    @property
    def max_open(self):
        return self.gen_size ** 2 - (self.gen_size // 2) ** 2

    @property
    def max_walls(self):
        return self.ambient ** 2 - 8

    def __array_function__(self, func, types, args, kwargs):
        if func is np.flip and args and args[0] is self:
            axis = kwargs.get('axis', args[1] if len(args) > 1 else None)
            if axis != 0:
                raise NotImplementedError('np.flip of a traced maze along axis %r' % (axis,))
            return TracedMatrix(self.gen_size, self.ambient, not self.flipped)
        raise NotImplementedError('np.%s on a traced maze matrix' % func.__name__)

    def __eq__(self, other):
        if isinstance(other, (int, float)) and other == 0:
            return _OpenMask(self)
        raise NotImplementedError('comparing a traced maze matrix with %r' % (other,))

    __hash__ = None


def generate(size, ambient_size):
    t = _trace.active()
    if getattr(t, 'maze', None) is not None:
        raise NotImplementedError('more than one random maze per state_initializer')
    m = TracedMatrix(size, ambient_size)
    t.maze = {'gen_size': m.gen_size, 'ambient': m.ambient, 'flip': None}
    t.add_op(MazeOp((_abi.MOOG_CELL_GENERATE, 0), []))
    return m


def bind(maze_obj):
    """Maze(traced matrix): fixes the orientation the device stores the matrix in."""
    t = _trace.active()
    if t is None or getattr(t, 'maze', None) is None:
        raise RuntimeError('a traced maze matrix outside a traced state_initializer')
    if t.maze['flip'] is not None:
        raise NotImplementedError('more than one Maze over the traced matrix')
    t.maze['flip'] = bool(maze_obj.maze.flipped)


def wall_sprites(maze_obj, color):
    t = _trace.active()
    out = []
    for k in range(maze_obj.maze.max_walls):
        t.suspend = True
        try:
            s = sprite_lib.Sprite(x=0., y=0., shape=CellShape((_abi.MOOG_CELL_WALL_RANK, k)), **color)
        finally:
            t.suspend = False
        t.add_op(MazeOp((_abi.MOOG_CELL_WALL_RANK, k), [s]))
        out.append(s)
    t.maze['walls'] = out
    return out


def sample_points(maze_obj, num_points):
    t = _trace.active()
    if num_points > _abi.MOOG_MAX_MAZE_POINTS:
        raise NotImplementedError('sample_distinct_open_points beyond %d points' % _abi.MOOG_MAX_MAZE_POINTS)
    if t.maze.get('sampled'):
        raise NotImplementedError('more than one sample_distinct_open_points per state_initializer')
    t.maze['sampled'] = True
    t.add_op(MazeOp((_abi.MOOG_CELL_SAMPLE, int(num_points)), []))
    return [traced_cell(_abi.MOOG_CELL_SAMPLED, k) for k in range(num_points)]


def cell_of(sprite):
    """The traced cell a sprite recipe is placed on (None: an ordinary sprite)."""
    cells = set()
    for v in sprite.factors.values():
        if isinstance(v, (CellIndex, CellShape)):
            cells.add(v.cell)
    if len(cells) > 1:
        raise NotImplementedError('a sprite whose factors come from different maze cells')
    return cells.pop() if cells else None
