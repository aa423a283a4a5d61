"""Factor distributions (reference: moog/state_initialization/distributions.py).

The device sampler runs `Continuous` (:78-116), `Discrete` (:119-156) and `Product`
(:251-330) directly, and `Mixture` (:159-204), `Intersection` (:207-258), `SetMinus`
(:304-352), `Selection` (:355-405) and `Discrete(probs=...)` through a distribution
program (`_distcode.py` -> `moog_dinstr_t`) that draws uniforms in the order the
reference's recursive `.sample()` does.  While an environment traces its
state_initializer `.sample()` returns symbolic factors; outside tracing it samples with
numpy (same algorithm as the reference) so configs that draw at build time keep working.
"""
import numpy as np

from .. import _trace
from ..sprite import SymbolicFactor


class AbstractDistribution(object):
    def _get_rng(self, rng=None):
        return np.random if rng is None else rng


class Continuous(AbstractDistribution):
    """Uniform on [minval, maxval), cast to `dtype` (distributions.py:81-100)."""

    def __init__(self, key, minval, maxval, dtype='float32'):
        self.key, self.minval, self.maxval, self.dtype = key, minval, maxval, dtype

    def sample(self, rng=None):
        if _trace.active() is not None:
            return {self.key: SymbolicFactor(self)}
        out = self._get_rng(rng).uniform(low=self.minval, high=self.maxval)
        return {self.key: np.asarray(out, dtype=self.dtype)}

    def contains(self, spec):
        return spec[self.key] >= self.minval and spec[self.key] < self.maxval

    @property
    def keys(self):
        return set([self.key])


class Discrete(AbstractDistribution):
    """Uniform choice among candidates (distributions.py:122-140)."""

    def __init__(self, key, candidates, probs=None):
        self.key, self.candidates, self.probs = key, candidates, probs

    def sample(self, rng=None):
        if _trace.active() is not None:
            if len(self.candidates) == 1 and self.probs is None:
                return {self.key: self.candidates[0]}
            return {self.key: SymbolicFactor(self)}
        idx = self._get_rng(rng).choice(len(self.candidates), p=self.probs)
        return {self.key: self.candidates[idx]}

    def contains(self, spec):
        return spec[self.key] in self.candidates

    @property
    def keys(self):
        return set([self.key])


class Product(AbstractDistribution):
    """Product of components with disjoint keys plus constants (distributions.py:254-301)."""

    def __init__(self, components, **constants):
        self.components = list(components) + [Discrete(k, [v]) for k, v in constants.items()]
        keys = [k for c in self.components for k in c.keys]
        if len(set(keys)) < len(keys):
            raise ValueError('All components must have different keys.')
        self._keys = set(keys)

    def sample(self, rng=None):
        out = {}
        for c in self.components:
            out.update(c.sample(rng=rng))
        return out

    def contains(self, spec):
        return all(c.contains(spec) for c in self.components)

    @property
    def keys(self):
        return self._keys


_MAX_TRIES = int(1e5)   # distributions.py:43


class _Composite(AbstractDistribution):
    """Distributions lowered as a whole: while tracing every key is a symbolic factor
    bound to this node (the enclosing generator's root distribution is what gets
    compiled)."""

    def _symbolic(self):
        return {k: SymbolicFactor(self) for k in self.keys}


class Mixture(_Composite):
    """Mixture of components with identical key sets (distributions.py:162-184)."""

    def __init__(self, components, probs=None):
        self.components = list(components)
        self.probs = (np.ones(len(self.components)) / len(self.components) if probs is None
                      else np.array(probs))
        self._keys = self.components[0].keys
        for c in self.components[1:]:
            if c.keys != self._keys:
                raise ValueError('All components must have the same key sets. However detected '
                                 'key sets {} and {}'.format(self._keys, c.keys))

    def sample(self, rng=None):
        if _trace.active() is not None:
            return self._symbolic()
        rng = self._get_rng(rng)
        return self.components[rng.choice(len(self.components), p=self.probs)].sample(rng=rng)

    def contains(self, spec):
        return any(c.contains(spec) for c in self.components)

    @property
    def keys(self):
        return self._keys


class Intersection(_Composite):
    """Samples components[index_for_sampling], rejects unless every component contains
    the sample (distributions.py:210-249)."""

    def __init__(self, components, index_for_sampling=0):
        self.components = list(components)
        self.index_for_sampling = index_for_sampling
        self._keys = self.components[0].keys
        for c in self.components[1:]:
            if c.keys != self._keys:
                raise ValueError('All components must have the same key sets. However detected '
                                 'key sets {} and {}'.format(self._keys, c.keys))

    def sample(self, rng=None):
        if _trace.active() is not None:
            return self._symbolic()
        rng = self._get_rng(rng)
        for _ in range(_MAX_TRIES):
            sample = self.components[self.index_for_sampling].sample(rng=rng)
            if all(c.contains(sample) for c in self.components):
                return sample
        raise ValueError('Maximum number of tried exceeded when trying to sample from Intersection.')

    def contains(self, spec):
        return all(c.contains(spec) for c in self.components)

    @property
    def keys(self):
        return self._keys


class SetMinus(_Composite):
    """Samples `base`, rejects samples contained in `hold_out` (distributions.py:307-343)."""

    def __init__(self, base, hold_out):
        self.base, self.hold_out = base, hold_out
        self._keys = base.keys
        if not hold_out.keys.issubset(self._keys):
            raise ValueError('Keys {} of hold_out is not a subset of keys {} of SetMinus base '
                             'distribution.'.format(hold_out.keys, base.keys))

    def sample(self, rng=None):
        if _trace.active() is not None:
            return self._symbolic()
        rng = self._get_rng(rng)
        for _ in range(_MAX_TRIES):
            sample = self.base.sample(rng=rng)
            if not self.hold_out.contains(sample):
                return sample
        raise ValueError('Maximum number of tried exceeded when trying to sample from SetMinus.')

    def contains(self, spec):
        return self.base.contains(spec) and not self.hold_out.contains(spec)

    @property
    def keys(self):
        return self._keys


class Selection(_Composite):
    """Samples `base`, keeps samples contained in `filtering` (distributions.py:358-397)."""

    def __init__(self, base, filtering):
        self.base, self.filtering = base, filtering
        self._keys = base.keys
        if not filtering.keys.issubset(self._keys):
            raise ValueError('Keys {} of filtering is not a subset of keys {} of Selection base '
                             'distribution.'.format(filtering.keys, base.keys))

    def sample(self, rng=None):
        if _trace.active() is not None:
            return self._symbolic()
        rng = self._get_rng(rng)
        for _ in range(_MAX_TRIES):
            sample = self.base.sample(rng=rng)
            if self.filtering.contains(sample):
                return sample
        raise ValueError('Maximum number of tried exceeded when trying to sample from Selection.')

    def contains(self, spec):
        return self.base.contains(spec) and self.filtering.contains(spec)

    @property
    def keys(self):
        return self._keys


class DependentDistribution(AbstractDistribution):
    """Some factors are a deterministic function of others (distributions.py:420-475):
    `dependent_fn(sample of independent_distrib) -> {key: value}`.  Inside a traced initializer / generator the
    function runs once on symbolic values ("this sprite's own factor k"); its results become expressions the device
    evaluates after the independent factors are drawn (MOOG_X_FACTOR), with numpy's float32 promotion."""

    def __init__(self, independent_distrib, dependent_fn, dependent_fn_keys):
        self._independent_distrib = independent_distrib
        self._dependent_fn = dependent_fn
        self._dependent_fn_keys = list(dependent_fn_keys)
        if not set(independent_distrib.keys).isdisjoint(set(dependent_fn_keys)):
            raise ValueError('independent_distrib keys {} and dependent_fn keys {} are not disjoint.'.format(
                independent_distrib.keys, dependent_fn_keys))

    def sample(self, rng=None):
        rng = self._get_rng(rng)
        sample = self._independent_distrib.sample(rng=rng)
        if _trace.active() is None:
            sample.update(self._dependent_fn(sample))
            return sample
        from .. import _symbolic
        from ..sprite import ExprFactor, ExprShape
        view = {}
        for k, v in sample.items():
            if isinstance(v, ExprFactor):
                view[k] = _symbolic.Sym(v.node)
            elif isinstance(v, SymbolicFactor) and not isinstance(v, ExprShape):
                if k == 'shape':
                    raise NotImplementedError('a dependent_fn over a sampled shape')
                view[k] = _symbolic.Sym(_symbolic.Node('selffac', k))
            else:
                view[k] = v
        sample.update(self._dependent_fn(view))
        return sample

    def contains(self, spec):
        ok = self._independent_distrib.contains(spec)
        dep = self._dependent_fn({k: spec[k] for k in self._independent_distrib.keys})
        for k in self._dependent_fn_keys:
            ok &= spec[k] == dep[k]
        return ok

    @property
    def keys(self):
        return set(self._independent_distrib.keys).union(self._dependent_fn_keys)

    def to_str(self, indent):
        return indent * '  ' + '<DependentDistribution: independent_distrib={}, dependent_fn={}>'.format(
            self._independent_distrib, self._dependent_fn)


def is_flat(dist):
    """True when the flat per-factor sampler (moog_factor_t) covers `dist`."""
    if isinstance(dist, Continuous):
        return True
    if isinstance(dist, Discrete):
        return dist.probs is None
    if isinstance(dist, Product):
        return all(is_flat(c) for c in dist.components)
    if isinstance(dist, DependentDistribution):   # its dependent factors are expressions of the op's own draws
        return is_flat(dist._independent_distrib)
    if type(dist).__module__ != __name__:   # a distribution class of the config's own (red_green.py:31-64): its sample()
        return True                         # ran on symbolic draws, its factors are expressions of the op's draws
    return False
