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


class Tracer(object):
    def __init__(self):
        self.ops = []               # GenOp, in randomness-consumption order
        self.op_of = {}             # id(sprite) -> (op, k)
        self.randint_calls = []
        self.maze = None            # the per-reset random maze of the initializer (maze_lib/_traced.py)

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
        t.add_op(GenOp(None, 1, 1, False, [], 0, [s]))


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
    try:
        yield t
    finally:
        np.random.randint = real_randint
        for name, fn in blocked.items():
            setattr(np.random, name, fn)
        _ACTIVE = prev
