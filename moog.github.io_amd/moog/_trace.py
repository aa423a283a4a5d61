"""Tracing context used while lowering a config's `state_initializer`.

The reference calls `state_initializer()` at every reset and gets fresh Python
sprites (environment.py:86).  The engine instead runs the initializer *once*,
symbolically: distributions return `SymbolicFactor`s, `generate_sprites`
returns placeholder sprites bound to a generation op, and
`np.random.randint` (used for `num_sprites=lambda: np.random.randint(a, b)`,
functional_maze.py:146) is recorded instead of drawn.  The recorded op list is
what the device-side sampler executes at every reset.
"""
import contextlib

import numpy as np

_ACTIVE = None


class GenOp(object):
    def __init__(self, dist, count_min, count_max, disjoint, avoid, max_tries, sprites):
        self.dist = dist
        self.count_min = count_min
        self.count_max = count_max
        self.disjoint = disjoint
        self.avoid = avoid          # list of Sprite objects (static or placeholder)
        self.max_tries = max_tries
        self.sprites = sprites      # placeholder Sprite objects, one per reserved slot


class HDrawOp(GenOp):
    """One direct np.random call of the initializer (np.random.uniform / binomial outside a distribution): an op
    without sprites that takes draw `index` of the reset; expressions refer to it as Node('hdraw', index)."""

    def __init__(self, index, seq):
        GenOp.__init__(self, None, 0, 0, False, [], 0, [])
        self.index, self.seq = index, seq
        from . import _abi
        self.cell = (_abi.MOOG_CELL_HDRAW, index)


class ShuffleOp(GenOp):
    """sprite_generators.shuffle: permutes the slots of `members` (generated just before) at every reset."""

    def __init__(self, members):
        GenOp.__init__(self, None, 0, 0, False, [], 0, [])
        self.members = members


class ChoiceOp(GenOp):
    """np.random.choice over the alternatives of a sample_generator: the picked index goes to direct-draw slot `index`."""

    def __init__(self, index, n, p):
        GenOp.__init__(self, None, 0, 0, False, [], 0, [])
        self.index, self.n, self.p = index, n, p


class Tracer(object):
    def __init__(self):
        self.ops = []               # GenOp, in randomness-consumption order
        self.op_of = {}             # id(sprite) -> (op, k)
        self.randint_calls = []
        self.n_hdraws = 0
        self.alias = {}             # id(placeholder of a later sample_generator alternative) -> the first alternative's
        self.alias_keep = []        # (keeps those placeholders alive: ids must stay unique)
        self.maze = None            # the per-reset random maze of the initializer (maze_lib/_traced.py)
        self.seq = 0                # order of sampling events (deferred factor samples and direct draws)

    def next_seq(self):
        self.seq += 1
        return self.seq

    def hdraw(self):
        """A new direct draw: a symbolic uniform in [0, 1)."""
        from . import _abi, _symbolic
        if getattr(self, 'suspend', False):
            raise NotImplementedError('np.random calls inside a distribution sampled by generate_sprites')
        if self.n_hdraws >= _abi.MOOG_MAX_HDRAWS:
            raise NotImplementedError('more than %d direct np.random draws per reset' % _abi.MOOG_MAX_HDRAWS)
        k = self.n_hdraws
        self.n_hdraws += 1
        self.add_op(HDrawOp(k, self.next_seq()))
        return _symbolic.Sym(_symbolic.Node('hdraw', k))

    def choose(self, generators, p, args, kwargs):
        """sample_generator: runs every alternative; their ops become conditional on the drawn index."""
        from . import _abi
        if self.n_hdraws >= _abi.MOOG_MAX_HDRAWS:
            raise NotImplementedError('more than %d direct np.random draws per reset' % _abi.MOOG_MAX_HDRAWS)
        k = self.n_hdraws
        self.n_hdraws += 1
        self.add_op(ChoiceOp(k, len(generators), None if p is None else [float(x) for x in p]))
        first = None
        for j, g in enumerate(generators):
            mark = len(self.ops)
            sprites = g(*args, **kwargs)
            for op in self.ops[mark:]:
                if getattr(op, 'cond', None) is not None:
                    raise NotImplementedError('a sample_generator inside a sample_generator')
                op.cond = (k, j)
            if first is None:
                first = sprites
            else:
                if len(sprites) != len(first):
                    raise NotImplementedError('sample_generator alternatives that return different numbers of sprites')
                for a, b in zip(sprites, first):
                    self.alias[id(a)] = b   # the alternatives fill the same slots
                    self.alias_keep.append(a)
        return first

    def add_op(self, op):
        self.ops.append(op)
        for k, s in enumerate(op.sprites):
            self.op_of[id(s)] = (op, k)


def active():
    return _ACTIVE


def note_sprite(s):
    """Called from Sprite.__init__: a symbolic sprite built directly from
    `dist.sample()` inside a traced initializer is its own one-sprite op."""
    t = _ACTIVE
    if t is None or getattr(t, 'suspend', False):
        return
    if s.is_symbolic:
        op = GenOp(None, 1, 1, False, [], 0, [s])
        # The sprite's own factor samples were taken (in the reference) when `dist.sample()` ran; direct draws made
        # after the first of them belong between / after them, not before the whole op.
        own = sorted((s.factors[k].seq, k) for k in s.sample_order)
        if own:
            late = [o for o in t.ops if isinstance(o, HDrawOp) and o.seq > own[0][0]]
            if late:
                for o in late:
                    t.ops.remove(o)
                merged = sorted([(q, ('factor', k)) for q, k in own] + [(o.seq, ('hdraw', o.index)) for o in late])
                op.draw_seq = [item for _, item in merged]
        t.add_op(op)


@contextlib.contextmanager
def tracing():
    global _ACTIVE
    t = Tracer()
    prev = _ACTIVE
    _ACTIVE = t
    real_randint = np.random.randint

    def fake_randint(low, high=None, size=None, dtype=int):
        if size is not None:
            raise NotImplementedError('randint(size=...) inside a traced state_initializer')
        if high is None:
            low, high = 0, low
        t.randint_calls.append((int(low), int(high)))
        return int(high) - 1
    np.random.randint = fake_randint

    # direct draws of the initializer become symbolic values the device re-draws at every reset (numpy's own
    # formulas over one uniform each, as tests/golden/make_golden.py defines them)
    def fake_uniform(low=0.0, high=1.0, size=None):
        if size is not None:
            raise NotImplementedError('np.random.uniform(size=...) inside a traced state_initializer')
        return low + (high - low) * t.hdraw()

    def fake_binomial(n, p, size=None):
        if n != 1:
            raise NotImplementedError('np.random.binomial(n != 1) inside a traced state_initializer')
        from . import _symbolic
        if size is None:
            return t.hdraw() < p
        count = int(np.prod(size))
        if np.ndim(size) > 1 or (np.ndim(size) == 1 and len(size) != 1):
            raise NotImplementedError('np.random.binomial with a multi-dimensional size')
        return _symbolic.SymVec([t.hdraw() < p for _ in range(count)])
    # Any other draw from numpy's global generator inside the initializer would be taken
    # once, at build time, and frozen into every episode: refuse it instead.
    blocked = {}

    def refuse(name):
        def _raise(*args, **kwargs):
            raise NotImplementedError(
                'np.random.%s inside a state_initializer is not lowered to the device sampler '
                '(sample through moog.state_initialization.distributions)' % name)
        return _raise
    for name in ('uniform', 'rand', 'randn', 'normal', 'random', 'random_sample', 'choice', 'shuffle',
                 'permutation', 'binomial'):
        blocked[name] = getattr(np.random, name)
        setattr(np.random, name, refuse(name))
    np.random.uniform = fake_uniform
    np.random.binomial = fake_binomial
    try:
        yield t
    finally:
        np.random.randint = real_randint
        for name, fn in blocked.items():
            setattr(np.random, name, fn)
        _ACTIVE = prev
