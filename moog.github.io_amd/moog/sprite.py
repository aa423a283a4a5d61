"""Sprite factor records (reference: moog/sprite.py:229-675).

In the reference a Sprite owns matplotlib paths and is mutated by Python loops.
Here a `Sprite` is only the *recipe* for one sprite slot: the 14 factors of
`Sprite.FACTOR_NAMES` (sprite.py:237-253).  Geometry (centred path, inertia,
world vertices) is derived on the device at reset from the shape table built by
`shape_record()` below, which restates sprite.py:329-409 for one raw shape.
"""
import numpy as np

FACTOR_NAMES = ('x', 'y', 'shape', 'angle', 'scale', 'aspect_ratio', 'c0', 'c1', 'c2',
                'opacity', 'x_vel', 'y_vel', 'angle_vel', 'mass', 'metadata')

_DEFAULTS = dict(x=0.5, y=0.5, shape='square', angle=0., scale=1., aspect_ratio=1., c0=0, c1=0,
                 c2=0, opacity=255, x_vel=0., y_vel=0., angle_vel=0., mass=1., metadata=None)


class SymbolicFactor(object):
    """A factor value that is drawn on the device at every reset."""

    def __init__(self, dist):
        self.dist = dist  # distributions.Continuous / Discrete
        from . import _trace
        t = _trace.active()
        self.seq = t.next_seq() if t is not None else 0   # when the reference would have drawn it
        self.owner = None                                  # (sprite, factor name) that carries the draw

    def __deepcopy__(self, memo):
        # `copy.deepcopy(factors)` of a sampled dict (match_to_sample.py:128) copies VALUES in the reference: the copy is
        # the same draw, not a new one
        return self

    __copy__ = lambda self: self


class ExprFactor(SymbolicFactor):
    """A factor value the initializer computed from its own np.random draws / from other sprites' factors: an
    expression tree (moog/_symbolic.py) evaluated on the device at every reset; takes no draw itself."""

    def __init__(self, node):
        SymbolicFactor.__init__(self, None)
        self.node = node


class ExprShape(SymbolicFactor):
    """A raw shape whose vertex coordinates are expressions (parallelogram_catch.py:34-43)."""

    def __init__(self, rows):
        SymbolicFactor.__init__(self, None)
        self.rows = rows   # [[node_x, node_y], ...]


class Sprite(object):
    """Recipe for one sprite (same constructor as sprite.py:261-276).

    Factor values may be numbers (static) or `SymbolicFactor`s produced by a
    distribution's `.sample()` while an environment is tracing its
    state_initializer.
    """
    FACTOR_NAMES = FACTOR_NAMES

    def __init__(self, **factors):
        unknown = set(factors) - set(_DEFAULTS)
        if unknown:
            raise TypeError('unknown sprite factors: %s' % sorted(unknown))
        self.factors = dict(_DEFAULTS)
        from . import _trace
        # built while no initializer is being traced = built once by the config: ONE Python object for all episodes
        # of the reference, so what rules do to it outlives resets (_compiler: slot_persist)
        self.built_outside = _trace.active() is None
        self.factors.update(self._adopt(factors))
        self.sample_order = [k for k in factors if isinstance(self.factors[k], SymbolicFactor) and
                             not isinstance(self.factors[k], (ExprFactor, ExprShape))]
        from . import _trace
        _trace.note_sprite(self)

    def _adopt(self, factors):
        """Symbolic values as factor records: expressions become ExprFactor / ExprShape; a sampled factor that
        another sprite already carries (`Sprite(x=other.x)`, multi_tracking_with_feature.py:137-141) becomes a
        reference to that sprite's value instead of a second draw."""
        from . import _symbolic
        out = {}
        for k, v in factors.items():
            if isinstance(v, _symbolic.Sym):
                v = ExprFactor(v.node)
            elif k == 'shape' and isinstance(v, _symbolic.SymMat):
                v = ExprShape([[_symbolic.lift(c) for c in row] for row in v.rows])
            elif isinstance(v, SymbolicFactor) and not isinstance(v, (ExprFactor, ExprShape)) \
                    and getattr(v, 'cell', None) is None:
                if v.owner is None:
                    v.owner = (self, k)
                elif v.owner[0] is not self:
                    v = ExprFactor(_symbolic.Node('slotattr', v.owner[0], v.owner[1]))
            out[k] = v
        return out

    @property
    def is_symbolic(self):
        return any(isinstance(v, SymbolicFactor) for v in self.factors.values())

    def overlaps_sprite(self, other):
        """Inside a traced initializer: the overlap test of the two sprites as they are at that point (the look-ahead
        loops of bounce_box_contact_prediction.py:42 / red_green.py:102-103)."""
        from . import _trace, _symbolic
        t = _trace.active()
        if t is None or not isinstance(other, Sprite):
            raise NotImplementedError('sprite.overlaps_sprite outside a traced state_initializer')
        t.go_live()
        return _symbolic.Sym(_symbolic.Node('overlaps_slots', self, other))

    def __setattr__(self, name, value):
        if name in ('position', 'velocity') and 'factors' in self.__dict__:
            # put back / moved after construction inside the initializer (bounce_box_contact_prediction.py:117-119)
            from . import _trace, _symbolic
            t = _trace.active()
            v, ln = _symbolic._elems(value)
            if t is None or ln != 2:
                raise NotImplementedError('assigning sprite.%s after construction outside a traced state_initializer' % name)
            t.go_live()
            keys = ('x', 'y') if name == 'position' else ('x_vel', 'y_vel')
            t.add_op(_trace.StoreOp(self, {k: _symbolic.lift(e) for k, e in zip(keys, v)}, name == 'velocity'))
            return
        # `sprite.mass = ...` after construction (e.g. predators_arena.py:88-89 inside its
        # state_initializer) is host logic the recipe cannot carry: refuse instead of ignoring it
        if name == 'metadata' and 'factors' in self.__dict__:
            # a label the config hangs on a sprite (match_to_sample.py:120-124): constant per slot, read by traced
            # reward functions / filters as `s.metadata[key]`
            self.factors['metadata'] = value
            return
        if name in FACTOR_NAMES and name != 'shape' and 'factors' in self.__dict__:
            # `s.mass = value` right after construction inside the initializer (predators_arena.py:95-96) is the same
            # recipe as Sprite(mass=value) -- unless the factor was sampled: the reference drew it and then dropped it
            from . import _symbolic, _trace
            if isinstance(value, (int, float, np.integer, np.floating, _symbolic.Sym)) and \
                    not isinstance(value, bool) and name not in self.sample_order:
                t = _trace.active()
                mine = t.op_of.get(id(self)) if t is not None else None
                if mine is not None and len(mine[0].sprites) > 1 and not isinstance(value, _symbolic.Sym) \
                        and name in ('c0', 'c1', 'c2', 'opacity'):
                    # ONE sprite of a generator call is repainted (red_green.py:161-167: two of the obstacles become
                    # red and green): a store right after the call, the recipe stays the generator's.  (Other factors
                    # set this way must get the same value on every sprite of the call: checked by the compiler.)
                    t.add_op(_trace.StoreOp(self, {name: _symbolic.lift(value)}, False))
                    return
                self.factors[name] = self._adopt({name: value})[name]
                return
            raise NotImplementedError(
                'assigning sprite.%s after construction is not lowered; pass it to Sprite(...) or '
                'through the factor distribution' % name)
        object.__setattr__(self, name, value)

    def __getattr__(self, name):
        f = self.__dict__.get('factors')
        if f is not None and name in ('position', 'velocity') or (f is not None and name in f and name != 'metadata'
                                                                   and name != 'shape'):
            # once the initializer looks at the sprites as they ARE (it read a position, tested an overlap, stepped the
            # physics), attribute reads are live values of the sprite's slot, no longer its recipe
            from . import _trace, _symbolic
            t = _trace.active()
            if t is not None and name in ('position', 'velocity'):
                t.go_live()
            if t is not None and t.live and not getattr(t, 'suspend', False):
                if name == 'position':
                    return _symbolic.SymVec([_symbolic.Sym(_symbolic.Node('live', self, k)) for k in ('x', 'y')])
                if name == 'velocity':
                    return _symbolic.SymVec([_symbolic.Sym(_symbolic.Node('live', self, k)) for k in ('x_vel', 'y_vel')])
                if name in _symbolic.ATTRS:
                    return _symbolic.Sym(_symbolic.Node('live', self, name))
        if f is not None and name in f:
            v = f[name]
            # a symbolic factor read back by the initializer (`Sprite(x=other.x)`, `1. - other.y`) is a value to
            # compute with: the expression itself, or a reference to this sprite's sampled value
            if isinstance(v, ExprFactor):
                from . import _symbolic
                return _symbolic.Sym(v.node)
            if isinstance(v, SymbolicFactor) and not isinstance(v, ExprShape) and getattr(v, 'cell', None) is None:
                from . import _symbolic
                return _symbolic.Sym(_symbolic.Node('slotattr', self, name))
            return v
        if name in ('position', 'velocity', 'vertices', 'contains_point', 'color', 'path',
                    'moment_of_inertia', 'max_radius', 'update_pos_from_vel'):
            # e.g. bounce_box_contact_prediction.py:113-121 / red_green.py:86-106 step the physics inside the
            # state_initializer to label or reject a trial
            raise NotImplementedError(
                'sprite.%s inside a state_initializer: a Sprite here is a recipe (its factors), live sprites exist on '
                'the device only; initialisers that simulate the episode on the host are not lowered' % name)
        raise AttributeError(name)


def _cross(a, b):
    return a[0] * b[1] - a[1] * b[0]


def shape_record(raw):
    """Centroid / inertia / centred CCW path of a raw shape (sprite.py:360-401).

    Returns (centred_vertices [n,2], centroid [2], inertia_over_area [2]).
    """
    raw = np.asarray(raw, dtype=np.float64)
    n = raw.shape[0]
    inertia = np.array([0., 0.])
    area = 0.
    centroid = np.array([0., 0.])
    for i in range(n):
        v0 = raw[i]
        v1 = raw[(i + 1) % n]
        cr = _cross(v0, v1)
        inertia += (1. / 12.) * cr * (v0 * v0 + v1 * v1 + v0 * v1)
        tri_area = cr / 2.
        area += tri_area
        centroid += ((v0 + v1) / 3.) * tri_area
    centroid /= area
    path = raw
    if area < 0:
        path = raw[::-1]
        inertia *= -1.
        area *= -1.
    # Affine2D().translate(-centroid): (1*x + 0*y) + (-c)
    neg = -1 * centroid
    centred = np.stack([(1.0 * path[:, 0] + 0.0 * path[:, 1]) + neg[0],
                        (0.0 * path[:, 0] + 1.0 * path[:, 1]) + neg[1]], axis=1)
    inertia -= area * np.square(centroid)
    return centred, centroid, inertia / area


# ---- module functions of the reference's sprite.py (config-time, host side) ------------------------------------------
_EPSILON_INTERPOLATION = 1e-8   # sprite.py:35


def update_sprite(sprite, **factors):
    """sprite.update_sprite (sprite.py:51-105) for a sprite RECIPE: the new factors replace the recipe's (the engine
    builds the live sprite from them, so "without resetting the shape unless necessary" has nothing to save here).
    What a recipe cannot carry (a factor of a sprite that is already live on the device) is refused by
    Sprite.__setattr__ with the reason."""
    unknown = set(factors) - set(FACTOR_NAMES)
    if unknown:
        raise TypeError('unknown sprite factors: %s' % sorted(unknown))
    for k, v in factors.items():
        if k == 'shape':
            sprite.factors['shape'] = sprite._adopt({'shape': v})['shape']
        else:
            setattr(sprite, k, v)


def _cross_2d(a, b):
    return a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]


def segment_crossing_coefficients(start_0, end_0, start_1, end_1):
    """sprite.py:108-164: for every pair of a segment of set 0 ([N, 2] starts / ends) and one of set 1 ([M, 2]) the
    coefficients A, B of start_0 + A * delta_0 = start_1 + B * delta_1 (both [N, M]); the denominator carries the
    reference's 1e-8 so that parallel segments do not divide by zero.  Host numpy (the engine's collision code has its
    own restatement in csrc/moog_device.h)."""
    start_0, end_0 = np.asarray(start_0, float), np.asarray(end_0, float)
    start_1, end_1 = np.asarray(start_1, float), np.asarray(end_1, float)
    d0 = (end_0 - start_0)[:, None]
    d1 = (end_1 - start_1)[None]
    rel = start_1[None] - start_0[:, None]
    den = _cross_2d(d0, d1) + _EPSILON_INTERPOLATION
    return _cross_2d(rel, d1) / den, _cross_2d(rel, d0) / den


def segment_crossings(start_0, end_0, start_1, end_1):
    """sprite.py:166-199: the crossing points ([K, 2]) of all pairs of segments whose A and B lie strictly inside
    (0, 1), and their segment indices ([K, 2])."""
    start_0, end_0 = np.asarray(start_0, float), np.asarray(end_0, float)
    A, B = segment_crossing_coefficients(start_0, end_0, start_1, end_1)
    inds = np.argwhere((A > 0) & (A < 1) & (B > 0) & (B < 1))
    pts = np.array([start_0[i] + A[i, j] * (end_0[i] - start_0[i]) for i, j in inds])
    return pts, inds


def sprite_edge_crossings(sprite_0, sprite_1):
    """sprite.py:202-224, for anything with closed `path.vertices` (first vertex repeated at the end) -- the reference's
    live sprites; a recipe has no path (Sprite.__getattr__ says so)."""
    v0, v1 = np.asarray(sprite_0.path.vertices, float), np.asarray(sprite_1.path.vertices, float)
    return segment_crossings(v0[:-1], v0[1:], v1[:-1], v1[1:])
