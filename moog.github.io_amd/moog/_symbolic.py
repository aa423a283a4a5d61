"""Symbolic execution of the small Python callables configs pass around: sprite filters
(`sprite -> bool`), modifiers (`sprite -> None`, assigning attributes), pair conditions and
reward functions (`(sprite_0, sprite_1) -> bool / number`).

The reference calls these lambdas on live `Sprite` objects every step (e.g.
game_rules/modify_sprites.py:41-52, vanish.py:58-61, tasks/contact_reward.py:85-92).  The
engine instead runs each callable once at build time on *symbolic* sprites whose attributes
are expression nodes; arithmetic, comparisons and numpy ufuncs build an expression tree,
and Python control flow (`if`, `and`, `or`, `any(...)`, which call `bool()` on a symbolic
value) is handled by enumerating the execution paths: every `bool()` on a symbolic value
is a decision that is explored both ways, and the per-path results are merged into
`select(cond, a, b)` nodes.  The tree is then emitted as postfix code for the expression VM
(include/moog_engine.h MOOG_X_*), which the oracle and the HIP engine both interpret.

Scalar dtype semantics follow numpy 2 (NEP 50): Python scalars are weak, float32 samples
stay float32 through arithmetic with weak scalars; the VM tracks that tag per value.
"""
import numpy as np

from . import _abi

MAX_PATHS = 256

# attribute ids (MOOG_XA_*)
ATTRS = ('x', 'y', 'x_vel', 'y_vel', 'angle', 'angle_vel', 'mass', 'c0', 'c1', 'c2', 'opacity',
         'scale', 'aspect_ratio')
SETTABLE = ('x', 'y', 'x_vel', 'y_vel', 'angle', 'angle_vel', 'mass', 'c0', 'c1', 'c2', 'opacity')


class Unsupported(NotImplementedError):
    pass


# ---- expression nodes ----------------------------------------------------------------------
class Node(object):
    """('const', value, strong) | ('attr', sprite, attr) | (op, *children)"""

    def __init__(self, op, *args):
        self.op, self.args = op, args

    def key(self):
        return (self.op,) + tuple(a.key() if isinstance(a, Node) else a for a in self.args)


def const(v, strong=False):
    return Node('const', float(v), bool(strong))


def lift(v):
    if isinstance(v, Sym):
        return v.node
    if isinstance(v, Node):
        return v
    if isinstance(v, (bool, np.bool_)):
        return const(1.0 if v else 0.0)
    if isinstance(v, (int, float)):
        return const(v)
    if isinstance(v, np.generic):          # numpy scalars are strong-typed
        return const(float(v), strong=True)
    if isinstance(v, np.ndarray) and v.ndim == 0:
        return const(float(v), strong=True)
    raise Unsupported('cannot use %r in a lowered expression' % (v,))


_TRACER = None


def _has_live(node):
    if node.op in ('live', 'overlaps_slots', 'simstep'):
        return True
    return any(isinstance(a, Node) and _has_live(a) for a in node.args)


class Sym(object):
    """A symbolic scalar."""
    __array_priority__ = 1000

    def __init__(self, node):
        self.node = node

    def _bin(self, op, other, swap=False):
        a, b = self.node, lift(other)
        if swap:
            a, b = b, a
        return Sym(Node(op, a, b))

    __add__ = lambda s, o: s._bin('add', o)
    __radd__ = lambda s, o: s._bin('add', o, True)
    __sub__ = lambda s, o: s._bin('sub', o)
    __rsub__ = lambda s, o: s._bin('sub', o, True)
    __mul__ = lambda s, o: s._bin('mul', o)
    __rmul__ = lambda s, o: s._bin('mul', o, True)
    __truediv__ = lambda s, o: s._bin('div', o)
    __rtruediv__ = lambda s, o: s._bin('div', o, True)
    __mod__ = lambda s, o: s._bin('rem', o)
    __lt__ = lambda s, o: s._bin('lt', o)
    __le__ = lambda s, o: s._bin('le', o)
    __gt__ = lambda s, o: s._bin('gt', o)
    __ge__ = lambda s, o: s._bin('ge', o)
    __eq__ = lambda s, o: s._bin('eq', o)
    __ne__ = lambda s, o: s._bin('ne', o)
    __and__ = lambda s, o: s._bin('and', o)
    __rand__ = lambda s, o: s._bin('and', o, True)
    __or__ = lambda s, o: s._bin('or', o)
    __ror__ = lambda s, o: s._bin('or', o, True)
    __neg__ = lambda s: Sym(Node('neg', s.node))
    __abs__ = lambda s: Sym(Node('abs', s.node))
    __invert__ = lambda s: Sym(Node('not', s.node))
    __hash__ = None

    def __bool__(self):
        if self.node.op == 'const':
            return self.node.args[0] != 0
        if _TRACER is None:
            # `while not valid: angle = np.random.uniform(...)` inside a state_initializer (match_to_sample.py:33-43):
            # the rejection loop over the latest draw is the one branch on drawn values that is lowered
            from . import _trace
            if _trace.active() is None:
                raise Unsupported('bool() of a symbolic value outside a traced function')
            if _has_live(self.node):   # a test on the sprites as they are: the look-ahead loop of an initializer
                return _trace.active().sim_decide(self.node)
            return _trace.active().retry_decide(self.node)
        return _TRACER.decide(self.node)

    def __float__(self):
        raise Unsupported('float() of a symbolic value')

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        return _ufunc(ufunc, method, inputs, kwargs)


_UFUNCS = {
    'add': 'add', 'subtract': 'sub', 'multiply': 'mul', 'true_divide': 'div', 'divide': 'div',
    'remainder': 'rem', 'mod': 'rem', 'minimum': 'min', 'maximum': 'max',
    'less': 'lt', 'less_equal': 'le', 'greater': 'gt', 'greater_equal': 'ge',
    'equal': 'eq', 'not_equal': 'ne', 'logical_and': 'and', 'logical_or': 'or',
    'negative': 'neg', 'absolute': 'abs', 'fabs': 'abs', 'sqrt': 'sqrt', 'sin': 'sin', 'cos': 'cos',
    'floor': 'floor', 'logical_not': 'not', 'square': 'square', 'sign': 'sign',
}


def _elems(v):
    """Elements of a scalar-or-vector operand, and its length (None for scalars)."""
    if isinstance(v, SymVec):
        return list(v.items), len(v.items)
    if isinstance(v, (list, tuple)) or (isinstance(v, np.ndarray) and v.ndim == 1):
        return list(v), len(v)
    return [v], None


def _ufunc(ufunc, method, inputs, kwargs):
    if method != '__call__' or kwargs.get('out') is not None:
        raise Unsupported('numpy %s.%s on symbolic values' % (ufunc.__name__, method))
    name = _UFUNCS.get(ufunc.__name__)
    if name is None:
        raise Unsupported('numpy.%s on symbolic values' % ufunc.__name__)
    parts = [_elems(v) for v in inputs]
    n = max([ln for _, ln in parts if ln is not None] or [None], key=lambda z: -1 if z is None else z)
    out = []
    for i in range(n or 1):
        args = [lift(p[i] if ln is not None else p[0]) for p, ln in parts]
        if name == 'square':
            out.append(Sym(Node('mul', args[0], args[0])))
        else:
            out.append(Sym(Node(name, *args)))
    return SymVec(out) if n else out[0]


def _matmul(m, v):
    """np.matmul(constant matrix, symbolic vector) (match_to_sample.py:73: a rotation by 90 degrees): row sums in index
    order.  Exact whatever BLAS does when every row has at most one entry that is not 0 (checked)."""
    m = np.asarray(m, dtype=np.float64) if not isinstance(m, (SymVec, SymMat)) else None
    if m is None or m.ndim != 2 or not isinstance(v, SymVec) or m.shape[1] != len(v.items):
        raise Unsupported('np.matmul other than (constant matrix) @ (symbolic vector)')
    if any(np.count_nonzero(row) > 1 for row in m):
        raise Unsupported('np.matmul with a row of several non-zero entries (BLAS rounding is not restated)')
    out = []
    for row in m:
        acc = None
        for c, x in zip(row, v.items):
            t = Sym(const(float(c), True)) * x
            acc = t if acc is None else acc + t
        out.append(acc)
    return SymVec(out)


def sort_network(items, let):
    """np.sort of a short list that holds drawn values: compare-exchange steps whose outputs are kept as computed
    cells (`let`), so that every later use reads a value instead of re-deriving a tree of min / max."""
    vals = [v if isinstance(v, Sym) else Sym(lift(float(v))) for v in items]
    n = len(vals)
    for i in range(n):
        for j in range(n - 1 - i):
            a, b = vals[j], vals[j + 1]
            if a.node.op == 'const' and b.node.op == 'const':
                lo, hi = min(a.node.args[0], b.node.args[0]), max(a.node.args[0], b.node.args[0])
                vals[j], vals[j + 1] = Sym(const(lo, True)), Sym(const(hi, True))
            else:
                vals[j], vals[j + 1] = let(Node('min', a.node, b.node)), let(Node('max', a.node, b.node))
    return SymVec(vals)


class SymVec(object):
    """A symbolic 1-D array (position, velocity, or the result of elementwise ops)."""
    __array_priority__ = 1000

    def __init__(self, items):
        self.items = [i if isinstance(i, Sym) else Sym(lift(i)) for i in items]

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        return iter(self.items)

    def __getitem__(self, i):
        return self.items[i]

    def _bin(self, fn, other):
        o, ln = _elems(other)
        if ln is not None and ln != len(self.items):
            raise Unsupported('shape mismatch in a symbolic expression')
        return SymVec([fn(a, o[i] if ln is not None else o[0]) for i, a in enumerate(self.items)])

    __add__ = lambda s, o: s._bin(lambda a, b: a + b, o)
    __radd__ = lambda s, o: s._bin(lambda a, b: b + a, o)
    __sub__ = lambda s, o: s._bin(lambda a, b: a - b, o)
    __rsub__ = lambda s, o: s._bin(lambda a, b: b - a, o)
    __mul__ = lambda s, o: s._bin(lambda a, b: a * b, o)
    __rmul__ = lambda s, o: s._bin(lambda a, b: b * a, o)
    __truediv__ = lambda s, o: s._bin(lambda a, b: a / b, o)
    __rtruediv__ = lambda s, o: s._bin(lambda a, b: b / a, o)
    __mod__ = lambda s, o: s._bin(lambda a, b: a % b, o)
    __lt__ = lambda s, o: s._bin(lambda a, b: a < b, o)
    __le__ = lambda s, o: s._bin(lambda a, b: a <= b, o)
    __gt__ = lambda s, o: s._bin(lambda a, b: a > b, o)
    __ge__ = lambda s, o: s._bin(lambda a, b: a >= b, o)
    __eq__ = lambda s, o: s._bin(lambda a, b: a == b, o)
    __ne__ = lambda s, o: s._bin(lambda a, b: a != b, o)
    __and__ = lambda s, o: s._bin(lambda a, b: a & b, o)
    __or__ = lambda s, o: s._bin(lambda a, b: a | b, o)
    __neg__ = lambda s: SymVec([-a for a in s.items])
    __abs__ = lambda s: SymVec([abs(a) for a in s.items])
    __hash__ = None

    def __bool__(self):
        raise ValueError('The truth value of an array with more than one element is ambiguous.')

    def any(self):
        r = self.items[0]
        for a in self.items[1:]:
            r = r | a
        return r

    def all(self):
        r = self.items[0]
        for a in self.items[1:]:
            r = r & a
        return r

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if ufunc.__name__ == 'matmul' and method == '__call__' and len(inputs) == 2 and not kwargs:
            return _matmul(inputs[0], inputs[1])
        return _ufunc(ufunc, method, inputs, kwargs)

    def __array_function__(self, func, types, args, kwargs):
        name = getattr(func, '__name__', '')
        if name == 'norm' and len(args) == 1 and not kwargs:     # np.linalg.norm: sqrt(dot(x, x))
            its = args[0].items
            if len(its) == 2:   # float64 2-vectors: OpenBLAS ddot = fma(x1, x1, x0 * x0) (plain sums in float32)
                return Sym(Node('sqrt', Node('fma', its[1].node, its[1].node, (its[0] * its[0]).node)))
            acc = None
            for a in its:
                t = a * a
                acc = t if acc is None else acc + t
            return Sym(Node('sqrt', acc.node))
        if name in ('any', 'all') and len(args) == 1 and not kwargs:
            return getattr(args[0], name)()
        if name == 'dot' and len(args) == 2:
            a, la = _elems(args[0])
            b, lb = _elems(args[1])
            if la != lb or la is None:
                raise Unsupported('np.dot shapes')
            if la == 2:   # (as np.linalg.norm above)
                return Sym(Node('fma', lift(a[1]), lift(b[1]), (Sym(lift(a[0])) * Sym(lift(b[0]))).node))
            acc = None
            for x, y in zip(a, b):
                t = Sym(lift(x)) * Sym(lift(y))
                acc = t if acc is None else acc + t
            return acc
        if name == 'stack' and len(args) == 1 and kwargs.get('axis', 0) in (0, 1):
            cols = [_elems(a)[0] for a in args[0]]
            if len(set(len(c) for c in cols)) != 1:
                raise Unsupported('np.stack of symbolic vectors of different lengths')
            if kwargs.get('axis', 0) == 0:
                return SymMat([list(c) for c in cols])
            return SymMat([[c[i] for c in cols] for i in range(len(cols[0]))])
        if name == 'copy' and len(args) == 1:
            from . import _trace
            t = _trace.active()
            if t is not None and t.live and _TRACER is None:   # np.copy(sprite.position): the value NOW, kept in a cell
                return SymVec([t.let(a.node, tagged=True) if _has_live(a.node) else a for a in args[0].items])
            return SymVec(list(args[0].items))
        if name == 'matmul' and len(args) == 2:
            return _matmul(args[0], args[1])
        if name == 'clip' and len(args) == 3:
            return np.minimum(np.maximum(args[0], args[1]), args[2])
        raise Unsupported('numpy.%s on symbolic values' % name)


class SymMat(object):
    """A symbolic 2-D array (rows of symbolic scalars): the vertex arrays a state_initializer computes from its
    draws (parallelogram_catch.py:34-68).  Elementwise arithmetic with scalars, [n, 1] / [n, m] arrays and
    length-m rows, in place or not; rows index and iterate as SymVec."""
    __array_priority__ = 1000

    def __init__(self, rows):
        self.rows = [[i if isinstance(i, Sym) else Sym(lift(i)) for i in r] for r in rows]

    @property
    def shape(self):
        return (len(self.rows), len(self.rows[0]))

    def __len__(self):
        return len(self.rows)

    def __iter__(self):
        return iter(SymVec(r) for r in self.rows)

    def __getitem__(self, i):
        if isinstance(i, tuple) and len(i) == 2 and all(isinstance(k, (int, np.integer)) for k in i):
            return self.rows[i[0]][i[1]]
        if isinstance(i, (int, np.integer)):
            return SymVec(self.rows[i])
        if isinstance(i, slice):
            return SymMat(self.rows[i])
        raise Unsupported('indexing a symbolic matrix with %r' % (i,))

    def _other(self, o, r, c):
        if isinstance(o, SymMat):
            rr, cc = o.shape
            return o.rows[r if rr > 1 else 0][c if cc > 1 else 0]
        if isinstance(o, SymVec):
            return o.items[c if len(o.items) > 1 else 0]
        if isinstance(o, np.ndarray):
            if o.ndim == 0:
                return o[()]
            if o.ndim == 1:
                return o[c if o.shape[0] > 1 else 0]
            if o.ndim == 2:
                return o[r if o.shape[0] > 1 else 0, c if o.shape[1] > 1 else 0]
            raise Unsupported('broadcasting a symbolic matrix with a %d-d array' % o.ndim)
        return o

    def _bin(self, fn, other):
        return SymMat([[fn(a, self._other(other, r, c)) for c, a in enumerate(row)]
                       for r, row in enumerate(self.rows)])

    def _ibin(self, fn, other):
        self.rows = self._bin(fn, other).rows
        return self

    __add__ = lambda s, o: s._bin(lambda a, b: a + b, o)
    __radd__ = lambda s, o: s._bin(lambda a, b: b + a, o)
    __sub__ = lambda s, o: s._bin(lambda a, b: a - b, o)
    __rsub__ = lambda s, o: s._bin(lambda a, b: b - a, o)
    __mul__ = lambda s, o: s._bin(lambda a, b: a * b, o)
    __rmul__ = lambda s, o: s._bin(lambda a, b: b * a, o)
    __truediv__ = lambda s, o: s._bin(lambda a, b: a / b, o)
    __iadd__ = lambda s, o: s._ibin(lambda a, b: a + b, o)
    __isub__ = lambda s, o: s._ibin(lambda a, b: a - b, o)
    __imul__ = lambda s, o: s._ibin(lambda a, b: a * b, o)
    __itruediv__ = lambda s, o: s._ibin(lambda a, b: a / b, o)
    __neg__ = lambda s: SymMat([[-a for a in r] for r in s.rows])
    __hash__ = None

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        name = _UFUNCS.get(ufunc.__name__)
        if method != '__call__' or kwargs.get('out') is not None or name is None:
            raise Unsupported('numpy %s.%s on a symbolic matrix' % (ufunc.__name__, method))
        mats = [m for m in inputs if isinstance(m, SymMat)]
        nr, nc = mats[0].shape
        rows = []
        for r in range(nr):
            row = []
            for c in range(nc):
                args = [lift(m.rows[r][c] if isinstance(m, SymMat) else mats[0]._other(m, r, c)) for m in inputs]
                row.append(Sym(Node('mul', args[0], args[0])) if name == 'square' else Sym(Node(name, *args)))
            rows.append(row)
        return SymMat(rows)


class SymSprite(object):
    """A sprite whose factors are symbolic; attribute writes are recorded."""

    def __init__(self, index, first_of=None):
        object.__setattr__(self, '_index', index)
        object.__setattr__(self, '_written', {})
        object.__setattr__(self, '_first_of', first_of)   # layer name when this is state[L][0]

    def overlaps_sprite(self, other):
        """sprite.py:462-484 as a symbolic test (lowered to the engine's overlap routine):
        only a representative of a quantified layer against the first sprite of a layer."""
        if not isinstance(other, SymSprite):
            raise Unsupported('overlaps_sprite with a concrete sprite')
        a, b = self, other
        if a._first_of is not None and b._first_of is None:
            a, b = b, a          # overlap is symmetric (intersects_path both ways, filled)
        if a._first_of is not None or b._first_of is None:
            raise Unsupported('overlaps_sprite is lowered for (layer sprite, state[L][0]) only')
        return Sym(Node('overlaps', a._index, b._first_of))

    def _get(self, name):
        w = self._written
        return w[name] if name in w else Sym(Node('attr', self._index, name))

    def __getattr__(self, name):
        if name == 'metadata':   # constant per slot (the sprite's `metadata` factor): a table lookup on the device
            return _SymMetadata(self._index)
        if name == 'position':
            return SymVec([self._get('x'), self._get('y')])
        if name == 'velocity':
            return SymVec([self._get('x_vel'), self._get('y_vel')])
        if name == 'color':
            return (self._get('c0'), self._get('c1'), self._get('c2'))
        if name in ATTRS:
            return self._get(name)
        raise Unsupported('sprite.%s is not available to lowered functions' % name)

    def __setattr__(self, name, value):
        if name == 'position' or name == 'velocity':
            v, ln = _elems(value)
            if ln != 2:
                raise Unsupported('sprite.%s must be set to a length-2 value' % name)
            ks = ('x', 'y') if name == 'position' else ('x_vel', 'y_vel')
            for k, e in zip(ks, v):
                self._written[k] = Sym(lift(e))
            self._written['__vec_' + name] = True
            return
        if name not in SETTABLE:
            raise Unsupported('assigning sprite.%s is not lowered' % name)
        self._written[name] = Sym(lift(value))


class _SymMetadata(object):
    """`sprite.metadata` inside a traced function: `[key]` is the number / bool the config stored under that key
    for the sprite's slot (sprite.py:248-253: metadata is never touched by the engine)."""

    def __init__(self, index):
        self._index = index

    def __getitem__(self, key):
        return Sym(Node('meta', self._index, key))

    def get(self, key, default=None):
        raise Unsupported('sprite.metadata.get(...) in a lowered function (index it: metadata[key])')


class _Tracer(object):
    def __init__(self):
        self.forced = []
        self.trail = []

    def decide(self, node):
        i = len(self.trail)
        v = self.forced[i] if i < len(self.forced) else True
        self.trail.append((node, v))
        return v


def _explore(fn, n_sprites):
    """Runs fn on symbolic sprites along every execution path.  Returns a list of
    (decisions [(node, bool)], return value, [written dicts])."""
    global _TRACER
    paths = []
    forced = []
    while True:
        tr = _Tracer()
        tr.forced = list(forced)
        sprites = [SymSprite(i) for i in range(n_sprites)]
        prev, _TRACER = _TRACER, tr
        try:
            ret = fn(*sprites)
        finally:
            _TRACER = prev
        paths.append((list(tr.trail), ret, [dict(s._written) for s in sprites]))
        if len(paths) > MAX_PATHS:
            raise Unsupported('too many execution paths in a lowered function')
        # next path: flip the last decision that was taken as True
        trail = tr.trail
        k = len(trail) - 1
        while k >= 0 and trail[k][1] is False:
            k -= 1
        if k < 0:
            return paths
        forced = [v for _, v in trail[:k]] + [False]


def _merge(paths, leaf):
    """select-tree over the decision trail; `leaf(path)` gives the node at a path's end."""
    def build(group, depth):
        if len(group) == 1 and len(group[0][0]) <= depth:
            return leaf(group[0])
        cond = group[0][0][depth][0]
        t = [p for p in group if p[0][depth][1]]
        f = [p for p in group if not p[0][depth][1]]
        if not t or not f:
            return build(t or f, depth + 1)
        a, b = build(t, depth + 1), build(f, depth + 1)
        if a.key() == b.key():
            return a
        return Node('select', cond, a, b)
    return build(paths, 0)


def trace_value(fn, n_sprites):
    """Expression of `fn(*sprites)`'s return value (bool or number)."""
    paths = _explore(fn, n_sprites)

    def leaf(p):
        ret = p[1]
        if isinstance(ret, SymVec):
            raise Unsupported('lowered function returned an array')
        if ret is None:
            raise Unsupported('lowered function returned None')
        return lift(ret)
    return _merge(paths, leaf)


def trace_scalar_fn(fn):
    """Expression of `fn(x)` for one scalar argument x (a DistanceForce's force_fn(distance), distance_fn_force.py:36): the
    argument is the node ('arg',), a float64 as np.linalg.norm returns it; branches on it become selects."""
    global _TRACER
    paths, forced = [], []
    while True:
        tr = _Tracer()
        tr.forced = list(forced)
        prev, _TRACER = _TRACER, tr
        try:
            ret = fn(Sym(Node('arg')))
        finally:
            _TRACER = prev
        paths.append((list(tr.trail), ret, []))
        if len(paths) > MAX_PATHS:
            raise Unsupported('too many execution paths in a lowered function')
        trail = tr.trail
        k = len(trail) - 1
        while k >= 0 and trail[k][1] is False:
            k -= 1
        if k < 0:
            break
        forced = [v for _, v in trail[:k]] + [False]

    def leaf(p):
        if isinstance(p[1], SymVec) or p[1] is None:
            raise Unsupported('a lowered scalar function must return a number')
        return lift(p[1])
    return _merge(paths, leaf)


def trace_modifier(fn):
    """{attr: expression} of the attribute writes of `fn(sprite)`; and whether the velocity
    was assigned as a whole (a fresh ndarray in the reference)."""
    paths = _explore(fn, 1)
    attrs = []
    for p in paths:
        for k in p[2][0]:
            if k not in attrs and not k.startswith('__'):
                attrs.append(k)
    out = {}
    for a in attrs:
        out[a] = _merge(paths, lambda p, a=a: p[2][0][a].node if a in p[2][0]
                        else Node('attr', 0, a))
    for a, b in (('x_vel', 'y_vel'), ('y_vel', 'x_vel'), ('x', 'y'), ('y', 'x')):
        if a in out and b not in out:   # the device stores a pair as a whole: the other component keeps its value
            out[b] = Node('attr', 0, b)
    vec_vel = any('__vec_velocity' in p[2][0] for p in paths)
    return out, vec_vel


def trace_rule_step(step_fn):
    """Lowers the `step(state, meta_state)` of a config-local rule class that modifies the first sprite of one layer
    (`s = state[L][0]; s.attr = expr(s)`, e.g. multi_tracking_with_feature.py:66-74).  Returns (layer name,
    {attr: expression}, velocity assigned as a whole)."""
    global _TRACER
    st = _SymState(1)
    made = []
    real_first = _SymLayer.__getitem__

    def first(self, i):
        sp = real_first(self, i)
        made.append((self._name, sp))
        return sp
    _SymLayer.__getitem__ = first
    tr = _Tracer()
    prev, _TRACER = _TRACER, tr
    try:
        step_fn(st, _SymMeta())
    finally:
        _TRACER = prev
        _SymLayer.__getitem__ = real_first
    if tr.trail:
        raise Unsupported('a config-local rule that branches on the state')
    written = [(l, sp) for l, sp in made if sp._written]
    if len(written) != 1 or any(k == 'quant' for k, _ in st.uses):
        raise Unsupported('a config-local rule must modify the first sprite of exactly one layer')
    layer, sp = written[0]
    for node in [v.node for k, v in sp._written.items() if not k.startswith('__')]:
        if _sprites_of(node, set()) - {0}:
            raise Unsupported('a config-local rule that reads other sprites')
    mod = {k: v.node for k, v in sp._written.items() if not k.startswith('__')}
    return layer, mod, '__vec_velocity' in sp._written


class _ZSprite(SymSprite):
    """Representative i of layer `layer` in a traced rule step: its attributes are Node('zattr', layer, i, attr)."""

    def __init__(self, layer, i):
        SymSprite.__init__(self, i)
        object.__setattr__(self, '_layer', layer)

    def _get(self, name):
        w = self._written
        return w[name] if name in w else Sym(Node('zattr', self._layer, self._index, name))


class _ZLayer(object):
    def __init__(self, owner, name):
        self._owner, self._name = owner, name

    def __iter__(self):
        return iter(self._owner.sprites(self._name))

    def __len__(self):
        raise Unsupported('len(state[layer]) is not symbolic')

    def __getitem__(self, i):
        raise Unsupported('indexing a layer in a rule that loops over layers')


class _ZState(object):
    def __init__(self):
        self.layers = {}

    def sprites(self, name):
        if name not in self.layers:
            self.layers[name] = [_ZSprite(name, i) for i in range(2)]
        return self.layers[name]

    def __getitem__(self, name):
        return _ZLayer(self, name)


def _zmap(node, fn):
    if node.op == 'zattr':
        return fn(node)
    if node.op in ('const', 'rdraw', 'phase_is', 'meta_num'):
        return node
    return Node(node.op, *[_zmap(a, fn) if isinstance(a, Node) else a for a in node.args])


def trace_rule_zip(step_fn):
    """Lowers the `step(state, meta_state)` of a config-local rule that (a) takes draws from np.random.uniform /
    randint and (b) loops over one layer, or over several layers in lock step (`for t, c in zip(state[A], state[B])`),
    assigning attributes computed from the draws and from the attributes of the sprites of that iteration
    (match_to_sample.py:66-79).  The step runs once on two representatives per layer; the second must repeat the first
    with the index shifted, which is what tells the lock-step loop from nested loops.  Returns (number of draws,
    [(layer, {attr: node}, velocity assigned as a whole)]) -- nodes over 'attr' (the sprite itself), 'zipattr' (its
    partner in another layer), 'rdraw'."""
    global _TRACER
    st = _ZState()
    draws = []

    def rdraw():
        draws.append(len(draws))
        return Sym(Node('rdraw', draws[-1]))

    def fake_uniform(low=0.0, high=1.0, size=None):
        if size is not None:
            raise Unsupported('np.random.uniform(size=...) in a rule step')
        return low + (high - low) * rdraw()

    def fake_randint(low, high=None, size=None, dtype=int):
        if size is not None:
            raise Unsupported('np.random.randint(size=...) in a rule step')
        if high is None:
            low, high = 0, low
        return int(low) + Sym(Node('floor', (rdraw() * float(int(high) - int(low))).node))   # a + int(u * (b - a))
    saved = (np.random.uniform, np.random.randint)
    np.random.uniform, np.random.randint = fake_uniform, fake_randint
    tr = _Tracer()
    prev, _TRACER = _TRACER, tr
    try:
        step_fn(st, _SymMeta())
    finally:
        _TRACER = prev
        np.random.uniform, np.random.randint = saved
    if tr.trail:
        raise Unsupported('a config-local rule that branches on the state or on its draws')
    out, written_attrs = [], set()
    for layer, reps in st.layers.items():
        w0 = {k: v.node for k, v in reps[0]._written.items() if not k.startswith('__')}
        w1 = {k: v.node for k, v in reps[1]._written.items() if not k.startswith('__')}
        if not w0 and not w1:
            continue
        shifted = {k: _zmap(n, lambda z: Node('zattr', z.args[0], 1 - z.args[1], z.args[2])) for k, n in w1.items()}
        if set(w0) != set(w1) or any(w0[k].key() != shifted[k].key() for k in w0):
            raise Unsupported('a config-local rule whose loop does not treat every sprite of a layer alike')
        mod = {}
        for k, n in w0.items():
            def own(z, _layer=layer):
                if z.args[1] != 0:
                    raise Unsupported('a config-local rule that reads the sprites of another iteration (nested loops)')
                return Node('attr', 0, z.args[2]) if z.args[0] == _layer else Node('zipattr', z.args[0], z.args[2])
            mod[k] = _zmap(n, own)
            written_attrs.add(k)
        for a, b in (('x_vel', 'y_vel'), ('y_vel', 'x_vel'), ('x', 'y'), ('y', 'x')):
            if a in mod and b not in mod:
                mod[b] = Node('attr', 0, b)
        out.append((layer, mod, '__vec_velocity' in reps[0]._written))
    if not out:
        raise Unsupported('a config-local rule that modifies no sprite')

    def reads(n, acc):
        if n.op in ('attr',):
            acc.add(n.args[1])
        elif n.op == 'zipattr':
            acc.add(n.args[1])
        for a in n.args:
            if isinstance(a, Node):
                reads(a, acc)
        return acc
    read_attrs = set()
    for _, mod, _v in out:
        for k, n in mod.items():
            if not (n.op == 'attr' and n.args[1] == k):
                reads(n, read_attrs)
    if read_attrs & written_attrs:   # a later iteration / layer would see the new value: order-dependent
        raise Unsupported('a config-local rule that reads an attribute it also assigns')
    return len(draws), out


# ---- state-level conditions ---------------------------------------------------------------------
class _SymLayer(object):
    """`state[layer]` inside a traced condition.  Iteration yields two representative sprites
    (enough to tell `all(...)` from `any(...)` in the explored paths); `[0]` is the layer's
    first sprite; `len()` is not symbolic (Python requires an int)."""

    def __init__(self, owner, name):
        self._owner, self._name = owner, name

    def __iter__(self):
        self._owner.note('quant', self._name)
        return iter([SymSprite(i) for i in range(self._owner.reps)])

    def __getitem__(self, i):
        if i != 0:
            raise Unsupported('only state[layer][0] is lowered')
        self._owner.note('first', self._name)
        return SymSprite(0, first_of=self._name)

    def __len__(self):
        raise Unsupported('len(state[layer]) is not symbolic')


class _SymState(object):
    def __init__(self, reps=2):
        self.uses = []
        self.reps = reps

    def note(self, kind, layer):
        if (kind, layer) not in self.uses:
            self.uses.append((kind, layer))

    def __getitem__(self, name):
        return _SymLayer(self, name)


class _SymMetaValue(object):
    """`meta_state[key]` inside a traced condition: `== 'phase name'` / `!=` test the value a PhaseSequence
    publishes (task_phases.py:126-127,140-141); in arithmetic and numeric comparisons it is the count a Fixation
    rule publishes under that key (fixation.py:47-58), i.e. that rule's state scalar."""

    def __init__(self, key):
        self._key = key

    def _num(self):
        return Sym(Node('meta_num', self._key))

    def __eq__(self, other):
        if isinstance(other, str):
            return Sym(Node('phase_is', self._key, other))
        return self._num() == other

    def __ne__(self, other):
        return Sym(Node('not', self.__eq__(other).node))

    __lt__ = lambda s, o: s._num() < o
    __le__ = lambda s, o: s._num() <= o
    __gt__ = lambda s, o: s._num() > o
    __ge__ = lambda s, o: s._num() >= o
    __add__ = lambda s, o: s._num() + o
    __radd__ = lambda s, o: o + s._num()
    __sub__ = lambda s, o: s._num() - o
    __rsub__ = lambda s, o: o - s._num()
    __mul__ = lambda s, o: s._num() * o
    __rmul__ = lambda s, o: o * s._num()
    __hash__ = None


class _SymMeta(object):
    def __getitem__(self, key):
        return _SymMetaValue(key)


class _FixedMetadata(object):
    def __init__(self, where):
        self._where = where

    def __getitem__(self, key):
        return Sym(Node('lmeta', self._where[0], self._where[1], key))


class _FixedSprite(object):
    """`state[layer][k]` in a state-level task function that names its sprites by position (bounce_box_contact_
    prediction.py:123-133): attributes, overlap tests with other such sprites, metadata."""

    def __init__(self, layer, k):
        self._where = (layer, int(k))

    def overlaps_sprite(self, other):
        if not isinstance(other, _FixedSprite):
            raise Unsupported('overlaps_sprite between a positional sprite and a quantified one')
        return Sym(Node('overlaps_slots', self._where, other._where))

    def __getattr__(self, name):
        if name == 'metadata':
            return _FixedMetadata(self._where)
        if name == 'position':
            return SymVec([Sym(Node('live', self._where, k)) for k in ('x', 'y')])
        if name == 'velocity':
            return SymVec([Sym(Node('live', self._where, k)) for k in ('x_vel', 'y_vel')])
        if name in ATTRS:
            return Sym(Node('live', self._where, name))
        raise Unsupported('sprite.%s is not available to lowered functions' % name)


class _FixedLayer(object):
    def __init__(self, name):
        self._name = name

    def __getitem__(self, i):
        if not isinstance(i, (int, np.integer)) or i < 0:
            raise Unsupported('state[layer][%r]' % (i,))
        return _FixedSprite(self._name, i)

    def __iter__(self):
        raise Unsupported('iterating a layer in a function that also names sprites by position')

    def __len__(self):
        raise Unsupported('len(state[layer]) is not symbolic')


class _FixedState(object):
    def __getitem__(self, name):
        return _FixedLayer(name)


def trace_state_fixed(fn, with_meta=False):
    """Expression of a state-level function (`condition(state)`, `reward_fn(state)`) that only looks at sprites named
    by position, `state[layer][k]`: their attributes, overlap tests between them, their metadata.  Leaves carry
    (layer, k); the compiler turns them into slots of layers whose size never changes."""
    global _TRACER
    paths, forced = [], []
    while True:
        tr = _Tracer()
        tr.forced = list(forced)
        prev, _TRACER = _TRACER, tr
        try:
            ret = fn(_FixedState(), _SymMeta()) if with_meta else fn(_FixedState())
        finally:
            _TRACER = prev
        if ret is None or isinstance(ret, (SymVec, SymMat)):
            raise Unsupported('a state-level function must return a number or a bool')
        paths.append((list(tr.trail), ret, []))
        if len(paths) > MAX_PATHS:
            raise Unsupported('too many execution paths in a lowered state function')
        trail = tr.trail
        k = len(trail) - 1
        while k >= 0 and trail[k][1] is False:
            k -= 1
        if k < 0:
            return _merge(paths, lambda p: lift(p[1]))
        forced = [v for _, v in trail[:k]] + [False]


def trace_index_filter(fn, layer):
    """The per-sprite predicate of `fn(state) -> indices into state[layer]` when that is a filter over the layer's own
    sprites, `[i for i, s in enumerate(state[layer]) if pred(s)]` (a config-local subclass of game_rules.Vanish,
    vanish.py:9-39).  Traced over two representative sprites: sprite 0 is in the result exactly where the predicate holds
    for it, and sprite 1's membership must be the same expression with the sprites exchanged (checked on random values
    of the leaves) -- indices that depend on the other sprites, on other layers or on the list's order are refused."""
    global _TRACER
    paths, forced, uses = [], [], []
    while True:
        tr = _Tracer()
        tr.forced = list(forced)
        st = _SymState(2)
        prev, _TRACER = _TRACER, tr
        try:
            ret = fn(st)
            ret = [int(i) for i in ret]
        finally:
            _TRACER = prev
        for u in st.uses:
            if u not in uses:
                uses.append(u)
        if any(i not in (0, 1) for i in ret) or sorted(set(ret)) != ret:
            raise Unsupported('_get_vanish_inds must return increasing indices of the layer\'s sprites')
        paths.append((list(tr.trail), ret, []))
        if len(paths) > MAX_PATHS:
            raise Unsupported('too many execution paths in _get_vanish_inds')
        trail = tr.trail
        k = len(trail) - 1
        while k >= 0 and trail[k][1] is False:
            k -= 1
        if k < 0:
            break
        forced = [v for _, v in trail[:k]] + [False]
    if uses != [('quant', layer)]:
        raise Unsupported('_get_vanish_inds must iterate over its own layer and nothing else (it used %s)' % (uses,))
    n0 = _merge(paths, lambda p: lift(float(0 in p[1])))
    n1 = _merge(paths, lambda p: lift(float(1 in p[1])))
    rs = np.random.RandomState(0)
    for _ in range(256):
        env = _RandomLeaves(rs)
        if _evaluate(n1, env) != _evaluate(_substitute(_substitute(_substitute(n0, 0, 2), 1, 0), 2, 1), env):
            raise Unsupported('_get_vanish_inds is not a per-sprite filter of its layer')
    seen = set()
    _sprites_of(n0, seen)
    if 1 in seen:
        raise Unsupported('_get_vanish_inds: whether a sprite vanishes depends on another sprite of the layer')
    return n0


def _substitute(node, old, new):
    if node.op == 'attr':
        return Node('attr', new if node.args[0] == old else node.args[0], node.args[1])
    if node.op == 'overlaps':
        return Node('overlaps', new if node.args[0] == old else node.args[0], node.args[1])
    if node.op in ('const', 'phase_is', 'meta_num', 'hdraw', 'slotattr', 'selffac', 'meta', 'rdraw', 'zattr', 'zipattr',
                   'pstate', 'live', 'overlaps_slots', 'lmeta', 'simstep', 'hdrawt'):
        return node
    return Node(node.op, *[_substitute(a, old, new) if isinstance(a, Node) else a for a in node.args])


def _sprites_of(node, acc):
    if node.op in ('attr', 'overlaps'):
        acc.add(node.args[0])
    elif node.op == 'meta':
        acc.add(node.args[0])
    elif node.op not in ('const', 'phase_is', 'meta_num', 'hdraw', 'slotattr', 'selffac', 'rdraw', 'zattr', 'zipattr',
                         'pstate', 'live', 'overlaps_slots', 'lmeta', 'simstep', 'hdrawt'):
        for a in node.args:
            if isinstance(a, Node):
                _sprites_of(a, acc)
    return acc


class _RandomLeaves(object):
    """Random values for the leaves of an expression (consistent within one assignment)."""

    def __init__(self, rs):
        self.rs, self.vals = rs, {}

    def get(self, key, boolean):
        if key not in self.vals:
            self.vals[key] = float(self.rs.randint(2)) if boolean else float(self.rs.choice(
                [0., 0.3, 0.5, 0.6, 1., -1., 2.5]))
        return self.vals[key]


def _evaluate(node, env):
    """Python-float value of an expression tree (used to check algebraic properties)."""
    op, a = node.op, node.args
    if op == 'const':
        return a[0]
    if op == 'attr':
        return env.get(node.key(), False)
    if op in ('overlaps', 'phase_is'):
        return env.get(node.key(), True)
    if op in ('meta_num', 'hdraw', 'selffac', 'slotattr'):
        return env.get(node.key(), False)
    v = [_evaluate(x, env) for x in a]
    if op == 'select':
        return v[1] if v[0] != 0 else v[2]
    table = {
        'add': lambda: v[0] + v[1], 'sub': lambda: v[0] - v[1], 'mul': lambda: v[0] * v[1],
        'div': lambda: v[0] / v[1] if v[1] else float('inf'), 'rem': lambda: np.remainder(v[0], v[1]),
        'min': lambda: min(v), 'max': lambda: max(v), 'lt': lambda: float(v[0] < v[1]),
        'le': lambda: float(v[0] <= v[1]), 'gt': lambda: float(v[0] > v[1]), 'ge': lambda: float(v[0] >= v[1]),
        'eq': lambda: float(v[0] == v[1]), 'ne': lambda: float(v[0] != v[1]),
        'and': lambda: float(v[0] != 0 and v[1] != 0), 'or': lambda: float(v[0] != 0 or v[1] != 0),
        'neg': lambda: -v[0], 'abs': lambda: abs(v[0]), 'sqrt': lambda: abs(v[0]) ** 0.5,
        'sin': lambda: np.sin(v[0]), 'cos': lambda: np.cos(v[0]), 'floor': lambda: np.floor(v[0]),
        'not': lambda: float(v[0] == 0), 'sign': lambda: np.sign(v[0]),
    }
    return float(table[op]())


def trace_state_condition(fn, with_meta=False):
    """Lowers `condition(state)` / `condition(state, meta_state)` of the forms
        all(pred(s) for s in state[L])  /  any(...)      -> ('all' | 'any', L, pred expression)
        expr(state[L][0])                                 -> ('first', L, expression)
    where pred / expr only read sprite attributes."""
    def explore(reps):
        global _TRACER
        paths, forced, uses = [], [], []
        while True:
            tr = _Tracer()
            tr.forced = list(forced)
            st = _SymState(reps)
            prev, _TRACER = _TRACER, tr
            try:
                ret = fn(st, _SymMeta()) if with_meta else fn(st)
            finally:
                _TRACER = prev
            for u in st.uses:
                if u not in uses:
                    uses.append(u)
            paths.append((list(tr.trail), ret, []))
            if len(paths) > MAX_PATHS:
                raise Unsupported('too many execution paths in a lowered condition')
            trail = tr.trail
            k = len(trail) - 1
            while k >= 0 and trail[k][1] is False:
                k -= 1
            if k < 0:
                return paths, uses
            forced = [v for _, v in trail[:k]] + [False]

    paths, uses = explore(2)
    if not uses:   # reads only the meta-state (e.g. the current phase)
        return 'plain', None, _merge(paths, lambda p: lift(p[1]))
    quant = [l for k, l in uses if k == 'quant']
    if len(quant) > 1 or (not quant and len(uses) != 1):
        raise Unsupported('condition must iterate over at most one layer')
    if quant and not all(isinstance(p[1], (bool, np.bool_)) for p in paths):
        # an accumulated number, e.g. `n += s.overlaps_sprite(agent)` in a loop (cleanup.py:181-190):
        # the per-sprite term comes from a one-sprite layer; that the two-sprite result is the sum
        # of the two terms is checked on random assignments of the attribute / overlap values
        one, _ = explore(1)
        g = _merge(one, lambda p: lift(p[1]))
        h = _merge(paths, lambda p: lift(p[1]))
        rs = np.random.RandomState(0)
        for _ in range(256):
            env = _RandomLeaves(rs)
            if abs(_evaluate(h, env) - (_evaluate(g, env) + _evaluate(_substitute(g, 0, 1), env))) > 1e-12:
                raise Unsupported('condition is not a sum of per-sprite terms over the layer')
        return 'count', quant[0], g
    kind, layer = ('quant', quant[0]) if quant else uses[0]
    if quant and len(uses) > 1:
        raise Unsupported('all / any conditions may read one layer only')
    if kind == 'first':
        return 'first', layer, _merge(paths, lambda p: lift(p[1]))
    # quantifier: every decision must be the same predicate on one of the two representatives
    preds = {}
    for trail, ret, _ in paths:
        if isinstance(ret, Sym):
            raise Unsupported('quantified condition must return a bool')
        for node, _v in trail:
            who = _sprites_of(node, set())
            if len(who) != 1:
                raise Unsupported('predicate must read one sprite')
            preds.setdefault(next(iter(who)), node)
    if set(preds) != {0, 1} or _substitute(preds[1], 1, 0).key() != preds[0].key():
        raise Unsupported('condition is not all(...) / any(...) of one predicate over the layer')
    table = {}
    for trail, ret, _ in paths:
        val = {next(iter(_sprites_of(n, set()))): v for n, v in trail}
        for a in (True, False):
            for b in (True, False):
                if val.get(0, a) == a and val.get(1, b) == b:
                    table[(a, b)] = bool(ret)
    if table == {(True, True): True, (True, False): False, (False, True): False, (False, False): False}:
        return 'all', layer, preds[0]
    if table == {(True, True): True, (True, False): True, (False, True): True, (False, False): False}:
        return 'any', layer, preds[0]
    raise Unsupported('condition is neither all(...) nor any(...) over the layer')


# ---- emission ---------------------------------------------------------------------------------
_BIN = {'add': 'ADD', 'sub': 'SUB', 'mul': 'MUL', 'div': 'DIV', 'rem': 'REM', 'min': 'MIN', 'max': 'MAX',
        'lt': 'LT', 'le': 'LE', 'gt': 'GT', 'ge': 'GE', 'eq': 'EQ', 'ne': 'NE', 'and': 'AND', 'or': 'OR'}
_UN = {'neg': 'NEG', 'abs': 'ABS', 'sqrt': 'SQRT', 'sin': 'SIN', 'cos': 'COS', 'floor': 'FLOOR',
       'not': 'NOT', 'sign': 'SIGN'}


def emit(node, out, resolver=None):
    """Postfix code (list of instruction dicts) for an expression tree.  `resolver(key, name)`
    maps a meta-state phase test to (rule index, phase index)."""
    if node.op == 'phase_is':
        if resolver is None:
            raise Unsupported('meta_state phase test outside a config with a PhaseSequence')
        rule, idx = resolver(node.args[0], node.args[1])
        out.append(dict(op=_abi.MOOG_X_RULE_STATE, a=rule))
        out.append(dict(op=_abi.MOOG_X_CONST, x=float(idx), b=0))
        out.append(dict(op=_abi.MOOG_X_EQ))
        return out
    if node.op == 'meta_num':   # the count a Fixation rule publishes under this meta-state key
        if resolver is None:
            raise Unsupported('meta_state value outside a config with a Fixation rule')
        out.append(dict(op=_abi.MOOG_X_RULE_STATE, a=resolver(node.args[0], None)))
        return out
    if node.op == 'overlaps':   # (layer sprite, state[L][0]); resolver(None, L) gives L's index
        if resolver is None:
            raise Unsupported('overlap test outside a state condition')
        out.append(dict(op=_abi.MOOG_X_OVERLAPS_FIRST, a=resolver(None, node.args[1]), b=int(node.args[0])))
        return out
    if node.op == 'arg':   # the scalar argument of a traced one-argument function (trace_scalar_fn)
        out.append(dict(op=_abi.MOOG_X_ARG))
        return out
    if node.op == 'const':
        out.append(dict(op=_abi.MOOG_X_CONST, x=node.args[0], b=int(node.args[1])))
    elif node.op == 'attr':
        out.append(dict(op=_abi.MOOG_X_ATTR, a=ATTRS.index(node.args[1]), b=int(node.args[0])))
    elif node.op == 'selffac':    # a factor of the sprite being created, as just sampled (DependentDistribution)
        out.append(dict(op=_abi.MOOG_X_FACTOR, a=_abi.FACTOR_NAMES.index(node.args[0])))
    elif node.op == 'hdraw':      # a direct np.random draw of the state_initializer (reset-time expressions)
        out.append(dict(op=_abi.MOOG_X_HDRAW, a=int(node.args[0])))
    elif node.op == 'live':       # attribute of a sprite as it is now (not its recipe); resolver('slot', sprite) -> its slot
        if resolver is None:
            raise Unsupported('a live sprite attribute outside a state_initializer')
        out.append(dict(op=_abi.MOOG_X_SLOT_ATTR, a=ATTRS.index(node.args[1]), b=int(resolver('slot', node.args[0]))))
    elif node.op == 'simstep':    # the loop counter of the initializer's look-ahead; resolver('simstep', None) -> its cell
        if resolver is None:
            raise Unsupported('a look-ahead loop counter outside a state_initializer')
        out.append(dict(op=_abi.MOOG_X_HDRAW, a=int(resolver('simstep', None))))
    elif node.op == 'overlaps_slots':   # sprite_a.overlaps_sprite(sprite_b) for two fixed sprites
        if resolver is None:
            raise Unsupported('an overlap test between fixed sprites outside a state_initializer / state function')
        out.append(dict(op=_abi.MOOG_X_OVERLAPS_SLOTS, a=int(resolver('slot', node.args[0])),
                        b=int(resolver('slot', node.args[1]))))
    elif node.op == 'lmeta':      # metadata[key] of the sprite at (layer, k): resolver('lmeta', ...) -> a node (constant or
        if resolver is None:      # a value the initializer's look-ahead selected)
            raise Unsupported('sprite metadata outside a task function')
        emit(resolver('lmeta', node.args), out, resolver)
    elif node.op == 'hdrawt':     # a computed cell that carries its numpy dtype in the next cell (np.copy of a live attribute)
        out.append(dict(op=_abi.MOOG_X_HDRAW_T, a=int(node.args[0])))
    elif node.op == 'pstate':     # a number the initializer keeps across episodes; resolver('pstate', name) -> its slot
        if resolver is None:
            raise Unsupported('persistent initializer state outside a state_initializer')
        out.append(dict(op=_abi.MOOG_X_RULE_STATE, a=int(resolver('pstate', node.args[0]))))
    elif node.op == 'meta':       # sprite.metadata[key]: resolver('meta', key) gives the per-slot table in program.cand
        if resolver is None:
            raise Unsupported('sprite.metadata outside a rule / task function')
        how, what = resolver('meta', (node.args[0], node.args[1]))
        if how == 'table':
            out.append(dict(op=_abi.MOOG_X_SLOT_CONST, a=int(what), b=int(node.args[0])))
        else:   # the sprite on that side is one fixed sprite: its metadata value itself (a constant, or what the
            emit(what, out, resolver)   # initializer's look-ahead selected)
    elif node.op == 'slotattr':   # a factor of an earlier sprite; resolver('slot', sprite) gives its slot
        if resolver is None:
            raise Unsupported('sprite factor reference outside a state_initializer')
        out.append(dict(op=_abi.MOOG_X_SLOT_ATTR, a=ATTRS.index(node.args[1]), b=int(resolver('slot', node.args[0]))))
    elif node.op == 'rdraw':      # draw k of the config-local rule this code belongs to; resolver('rdraw', k) -> (rule, which scalar)
        if resolver is None:
            raise Unsupported('a step-time draw outside a rule')
        rule, which = resolver('rdraw', node.args[0])
        out.append(dict(op=_abi.MOOG_X_RULE_STATE2 if which else _abi.MOOG_X_RULE_STATE, a=int(rule)))
    elif node.op == 'zipattr':    # attribute of the sprite at the same list index in another layer; resolver(None, L) -> L's index
        if resolver is None:
            raise Unsupported('a zipped sprite outside a rule')
        out.append(dict(op=_abi.MOOG_X_ZIP_ATTR, a=ATTRS.index(node.args[1]), b=int(resolver(None, node.args[0]))))
    elif node.op in ('select', 'fma'):
        for a in node.args:
            emit(a, out, resolver)
        out.append(dict(op=_abi.MOOG_X_SELECT if node.op == 'select' else _abi.MOOG_X_FMA))
    elif node.op in _BIN:
        emit(node.args[0], out, resolver)
        emit(node.args[1], out, resolver)
        out.append(dict(op=getattr(_abi, 'MOOG_X_' + _BIN[node.op])))
    elif node.op in _UN:
        emit(node.args[0], out, resolver)
        out.append(dict(op=getattr(_abi, 'MOOG_X_' + _UN[node.op])))
    else:
        raise Unsupported('expression op %r' % (node.op,))
    return out


def depth(code):
    """Maximum value-stack depth of a postfix program."""
    d = m = 0
    for ins in code:
        op = ins['op']
        if op in (_abi.MOOG_X_CONST, _abi.MOOG_X_ATTR, _abi.MOOG_X_RULE_STATE, _abi.MOOG_X_OVERLAPS_FIRST,
                  _abi.MOOG_X_HDRAW, _abi.MOOG_X_SLOT_ATTR, _abi.MOOG_X_FACTOR, _abi.MOOG_X_RULE_STATE2,
                  _abi.MOOG_X_SLOT_CONST, _abi.MOOG_X_ZIP_ATTR, _abi.MOOG_X_HDRAW_T, _abi.MOOG_X_OVERLAPS_SLOTS,
                  _abi.MOOG_X_ARG):
            d += 1
        elif op in (_abi.MOOG_X_SELECT, _abi.MOOG_X_FMA):
            d -= 2
        elif op in (_abi.MOOG_X_STORE, _abi.MOOG_X_STORE_VERT):
            d -= 1
        elif op in (_abi.MOOG_X_NEG, _abi.MOOG_X_ABS, _abi.MOOG_X_SQRT, _abi.MOOG_X_SIN, _abi.MOOG_X_COS,
                    _abi.MOOG_X_FLOOR, _abi.MOOG_X_NOT, _abi.MOOG_X_SIGN):
            pass
        else:
            d -= 1
        m = max(m, d)
    return m


def is_constant(node):
    """The expression reads nothing (constants and arithmetic on them only)."""
    if node.op == 'const':
        return True
    if node.op in ('attr', 'overlaps', 'phase_is', 'meta_num', 'hdraw', 'slotattr', 'selffac'):
        return False
    return all(is_constant(a) for a in node.args if isinstance(a, Node))


def uses_attr(code, names):
    ids = [ATTRS.index(n) for n in names]
    return any(ins['op'] == _abi.MOOG_X_ATTR and ins['a'] in ids for ins in code) or \
        any(ins['op'] == _abi.MOOG_X_STORE and ins['a'] in ids for ins in code)
