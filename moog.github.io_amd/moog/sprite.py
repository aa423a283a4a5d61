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
        self.factors.update(factors)
        self.sample_order = [k for k in factors if isinstance(factors[k], SymbolicFactor)]
        from . import _trace
        _trace.note_sprite(self)

    @property
    def is_symbolic(self):
        return bool(self.sample_order)

    def __setattr__(self, name, value):
        # `sprite.mass = ...` after construction (e.g. predators_arena.py:88-89 inside its
        # state_initializer) is host logic the recipe cannot carry: refuse instead of ignoring it
        if name in FACTOR_NAMES and 'factors' in self.__dict__:
            raise NotImplementedError(
                'assigning sprite.%s after construction is not lowered; pass it to Sprite(...) or '
                'through the factor distribution' % name)
        object.__setattr__(self, name, value)

    def __getattr__(self, name):
        f = self.__dict__.get('factors')
        if f is not None and name in f:
            return f[name]
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
