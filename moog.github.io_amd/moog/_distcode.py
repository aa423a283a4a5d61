"""Lowers a factor-distribution tree to a distribution program (`moog_dinstr_t`,
include/moog_engine.h): straight-line sampling code with jumps for Mixture components
and rejection loops for Intersection / SetMinus / Selection, plus postfix predicate code
for the `contains()` tests.  Uniforms are consumed in the order of the reference's
recursive `.sample()` (distributions.py:95-100,137-140,176-180,236-249,296-301,337-351,
383-397)."""
import numpy as np

from . import _abi
from .state_initialization import distributions as distribs


class _Emitter(object):
    def __init__(self, program, cand_n, factor_value):
        self.P = program
        self.cand_n = cand_n
        self.code = []                 # dicts of instruction fields, absolute pcs assigned last
        self.factor_value = factor_value   # (factor name, python value) -> float (interns shapes)
        self.keys = set()

    # -- tables ---------------------------------------------------------------------
    def put_cands(self, values):
        off = self.cand_n
        if off + len(values) > _abi.MOOG_MAX_CAND:
            raise ValueError('too many Discrete candidates / probabilities')
        for v in values:
            self.P.cand[self.cand_n] = float(v)
            self.cand_n += 1
        return off

    def emit(self, op, **f):
        self.code.append(dict(op=op, **f))
        return len(self.code) - 1

    @staticmethod
    def fac(key):
        if key not in _abi.FACTOR_NAMES:
            raise NotImplementedError('factor %r is not a device factor' % (key,))
        return _abi.FACTOR_NAMES.index(key)

    # -- sampling code ---------------------------------------------------------------
    def sample(self, d, depth):
        if isinstance(d, distribs.Continuous):
            if str(d.dtype) not in ('float32', 'float64'):
                raise NotImplementedError('Continuous dtype %r' % (d.dtype,))
            self.keys.add(d.key)
            self.emit(_abi.MOOG_D_CONT, a=self.fac(d.key), b=int(str(d.dtype) == 'float32'),
                      x=float(d.minval), y=float(d.maxval))
        elif isinstance(d, distribs.Discrete):
            self.keys.add(d.key)
            vals = [self.factor_value(d.key, c) for c in d.candidates]
            if len(vals) == 1 and d.probs is None:   # rng.choice(1) draws nothing
                self.emit(_abi.MOOG_D_CONST, a=self.fac(d.key), x=vals[0])
            elif d.probs is None:
                self.emit(_abi.MOOG_D_DISC, a=self.fac(d.key), b=len(vals), c=self.put_cands(vals))
            else:
                c = self.put_cands(vals)
                self.emit(_abi.MOOG_D_DISCP, a=self.fac(d.key), b=len(vals), c=c,
                          d=self.put_cands(np.asarray(d.probs, dtype=np.float64)))
        elif isinstance(d, distribs.Product):
            for c in d.components:
                self.sample(c, depth)
        elif isinstance(d, distribs.Mixture):
            n = len(d.components)
            self.emit(_abi.MOOG_D_CHOICE, b=n, d=self.put_cands(np.asarray(d.probs, np.float64)))
            jumps = [self.emit(_abi.MOOG_D_JUMP, a=-1) for _ in range(n)]
            ends = []
            for j, c in zip(jumps, d.components):
                self.code[j]['a'] = len(self.code)
                self.sample(c, depth)
                ends.append(self.emit(_abi.MOOG_D_JUMP, a=-1))
            for j in ends:
                self.code[j]['a'] = len(self.code)
        elif isinstance(d, (distribs.Intersection, distribs.SetMinus, distribs.Selection)):
            if depth >= 2:
                raise NotImplementedError('rejection-sampling distributions nested deeper than 2')
            self.emit(_abi.MOOG_D_LOOP, a=depth)
            start = len(self.code)
            if isinstance(d, distribs.Intersection):
                self.sample(d.components[d.index_for_sampling], depth + 1)
                pred, accept = ('and', list(d.components)), 1
            elif isinstance(d, distribs.SetMinus):
                self.sample(d.base, depth + 1)
                pred, accept = d.hold_out, 0
            else:
                self.sample(d.base, depth + 1)
                pred, accept = d.filtering, 1
            self.emit(_abi.MOOG_D_TEST, a=depth, d=accept, x=float(start), pred=pred)
        else:
            raise NotImplementedError('distribution %r is not lowered' % (type(d).__name__,))

    # -- predicate code ----------------------------------------------------------------
    def pred(self, d, out):
        if isinstance(d, tuple) and d[0] == 'and':
            for c in d[1]:
                self.pred(c, out)
            out.append(dict(op=_abi.MOOG_P_AND, b=len(d[1])))
        elif isinstance(d, distribs.Continuous):
            out.append(dict(op=_abi.MOOG_P_RANGE, a=self.fac(d.key), x=float(d.minval), y=float(d.maxval)))
        elif isinstance(d, distribs.Discrete):
            vals = [self.factor_value(d.key, c) for c in d.candidates]
            out.append(dict(op=_abi.MOOG_P_SET, a=self.fac(d.key), b=len(vals), c=self.put_cands(vals)))
        elif isinstance(d, (distribs.Product, distribs.Intersection)):
            for c in d.components:
                self.pred(c, out)
            out.append(dict(op=_abi.MOOG_P_AND, b=len(d.components)))
        elif isinstance(d, distribs.Mixture):
            for c in d.components:
                self.pred(c, out)
            out.append(dict(op=_abi.MOOG_P_OR, b=len(d.components)))
        elif isinstance(d, distribs.SetMinus):
            self.pred(d.base, out)
            self.pred(d.hold_out, out)
            out.append(dict(op=_abi.MOOG_P_NOT))
            out.append(dict(op=_abi.MOOG_P_AND, b=2))
        elif isinstance(d, distribs.Selection):
            self.pred(d.base, out)
            self.pred(d.filtering, out)
            out.append(dict(op=_abi.MOOG_P_AND, b=2))
        else:
            raise NotImplementedError('contains() of %r is not lowered' % (type(d).__name__,))


def lower(program, dist, cand_n, factor_value):
    """Appends the program of `dist` to program.dcode.  Returns (code_off, keys, cand_n)."""
    em = _Emitter(program, cand_n, factor_value)
    em.sample(dist, 0)
    em.emit(_abi.MOOG_D_END)
    # predicates go after END; TEST instructions get their offsets
    for ins in list(em.code):
        if ins['op'] == _abi.MOOG_D_TEST:
            out = []
            em.pred(ins.pop('pred'), out)
            if len(out) > 30:
                raise NotImplementedError('contains() predicate too long')
            ins['c'], ins['b'] = len(em.code), len(out)
            em.code.extend(out)
    base = program.n_dcode
    if base + len(em.code) > _abi.MOOG_MAX_DCODE:
        raise ValueError('distribution programs too long (max %d instructions)' % _abi.MOOG_MAX_DCODE)
    for i, ins in enumerate(em.code):
        I = program.dcode[base + i]
        I.op = ins['op']
        I.a, I.b, I.c, I.d = ins.get('a', 0), ins.get('b', 0), ins.get('c', 0), ins.get('d', 0)
        I.x, I.y = ins.get('x', 0.0), ins.get('y', 0.0)
        # pcs inside the program are relative to its start: make them absolute
        if I.op == _abi.MOOG_D_JUMP:
            I.a += base
        elif I.op == _abi.MOOG_D_TEST:
            I.c += base
            I.x = float(int(ins['x']) + base)
    program.n_dcode = base + len(em.code)
    return base, em.keys, em.cand_n
