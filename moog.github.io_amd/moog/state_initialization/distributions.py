"""Factor distributions (reference: moog/state_initialization/distributions.py).

Supported on the device sampler: `Continuous` (:78-116), `Discrete` (:119-156)
and `Product` (:251-330).  While an environment traces its state_initializer
`.sample()` returns symbolic factors; outside tracing it samples with numpy so
configs that draw constants at build time keep working.
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
            if self.probs is not None:
                raise NotImplementedError('Discrete(probs=...) is not lowered to the device sampler')
            if len(self.candidates) == 1:
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
