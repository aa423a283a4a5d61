"""Game rules (reference: moog/game_rules/__init__.py:3-23).

Lowered to the device: VanishOnContact (vanish.py:63-86), VanishByFilter (:42-61),
ChangeLayer (change_layer.py), CreateSprites (create_sprites.py), TimedRule / DelayedRule /
TemporaryRule (timing.py), ConditionalRule (conditional.py), ModifySprites whose
modifier is the torus position wrap (modify_sprites.py:8-52 with
chase_avoid_torus.py:144-149), Portal (portal.py:11-76) and config-local rules
that register a lowering (`register_lowering`), e.g. functional_maze.py's
`Booster` (:23-78), recognised by class name and attributes.
"""
import numpy as np

from .. import _abi


class AbstractRule(object):
    """abstract_rule.py:6-34"""

    def reset(self, state, meta_state):
        pass

    def step(self, state, meta_state):
        raise NotImplementedError


class VanishOnContact(AbstractRule):
    def __init__(self, vanishing_layer, contacting_layer):
        self._layer = vanishing_layer
        self._contacting_layer = contacting_layer


class ModifyMetaState(AbstractRule):
    """modify_meta_state.py:7-24: calls `meta_state_modifier(meta_state)` every step.  The
    meta-state is a host-side Python object, so this rule runs on the host; it never
    touches sprites and is skipped by the device lowering."""

    host_side = True

    def __init__(self, meta_state_modifier):
        self._meta_state_modifier = meta_state_modifier

    def step(self, state, meta_state):
        del state
        self._meta_state_modifier(meta_state)


class UpdateMetaStateValue(AbstractRule):
    """modify_meta_state.py:28-48: meta_state[key] = value every step it runs (typically a one-time rule of a Phase).
    Host side, like ModifyMetaState."""

    host_side = True

    def __init__(self, key, value):
        self._key = key
        self._value = value

    def step(self, state, meta_state):
        del state
        meta_state[self._key] = self._value


class _ProbeSprite(object):
    def __init__(self, pos):
        self.position = np.array(pos, dtype=float)


class ModifySprites(AbstractRule):
    def __init__(self, layers, modifier, sample_one=False, filter_fn=None):
        if isinstance(layers, str):
            layers = [layers]
        self._layers = list(layers)
        self._modifier = modifier
        self._sample_one = sample_one
        self._filter_fn = filter_fn

    def classify(self):
        """The torus wrap `position = remainder(position, 1)` over whole layers
        (chase_avoid_torus.py:144-149) keeps its dedicated kernel path; everything else is
        traced symbolically.  Returns (kind, filter kind, filter node, modifier dict, vec_vel)."""
        from .. import _symbolic
        if not self._sample_one and self._filter_fn is None:
            try:
                pts = [(1.25, -0.25), (0.5, 0.75), (-3.5, 2.0), (0.999, 1e-3)]
                ok = True
                for p in pts:
                    s = _ProbeSprite(p)
                    self._modifier(s)
                    ok = ok and list(s.__dict__) == ['position'] and np.array_equal(
                        np.asarray(s.position), np.remainder(np.array(p), 1))
                if ok:
                    return _abi.MOOG_RULE_TORUS_WRAP, _abi.MOOG_FILTER_ALWAYS, None, None, False
            except Exception:  # pylint: disable=broad-except
                pass
        fk, fnode = _classify_filter(self._filter_fn)
        mod, vec_vel = _symbolic.trace_modifier(self._modifier)
        return _abi.MOOG_RULE_MODIFY_SPRITES, fk, fnode, mod, vec_vel


class ModifyOnContact(AbstractRule):
    """contact_rules.py:58-141: modifier_0 on the sprites of layers_0 that pass filter_0 and
    touch a sprite of layers_1 (other than themselves); then the same with the roles swapped."""

    def __init__(self, layers_0, layers_1, modifier_0=None, modifier_1=None, filter_0=None,
                 filter_1=None):
        if not isinstance(layers_0, (list, tuple)):
            layers_0 = (layers_0,)
        if not isinstance(layers_1, (list, tuple)):
            layers_1 = (layers_1,)
        self._layers_0, self._layers_1 = tuple(layers_0), tuple(layers_1)
        self._modifier_0, self._modifier_1 = modifier_0, modifier_1
        self._filter_0, self._filter_1 = filter_0, filter_1


class Portal(AbstractRule):
    def __init__(self, teleporting_layer, portal_layer):
        self._teleporting_layer = teleporting_layer
        self._portal_layer = portal_layer


class _RuleDraws(AbstractRule):
    """Lowering product: the np.random draws a config-local rule takes when stepped (first, second)."""

    def __init__(self, first, n):
        self.first, self.n = first, n


class _ModifyTraced(AbstractRule):
    """Lowering product: the attribute writes of a traced config-local rule to every sprite of `layer`."""

    def __init__(self, layer, mod, vec, draws):
        self.layer, self.mod, self.vec, self.draws = layer, mod, vec, draws


def expand_local_rule(r):
    """A config-local rule class whose step() draws from np.random and loops over layers (match_to_sample.py:53-79)
    becomes [draws ...] + one traced modifier per assigned layer; every other rule stays itself."""
    known = (VanishOnContact, VanishByFilter, ChangeLayer, ModifyOnContact, CreateSprites, KeepNearCenter, Fixation,
             Phase, PhaseSequence, TimedRule, ConditionalRule, ModifySprites, Portal, _RuleDraws, _ModifyTraced)
    if isinstance(r, known) or lookup_lowering(r) is not None or getattr(r, 'host_side', False):
        return [r]
    if isinstance(r, Vanish) and type(r).step is Vanish.step:
        # a config-local vanishing rule (vanish.py:9-39): its index function, traced, is a filter over the layer
        if getattr(r, '_moog_expanded', None) is None:
            from .. import _symbolic
            try:
                node = _symbolic.trace_index_filter(r._get_vanish_inds, r._layer)
            except _symbolic.Unsupported as e:
                raise NotImplementedError('rule %s: %s' % (type(r).__name__, e))
            low = VanishByFilter(r._layer)
            low._traced_filter = node
            r._moog_expanded = [low]
        return r._moog_expanded
    if getattr(r, '_moog_expanded', None) is not None:
        return r._moog_expanded
    r._moog_expanded = _expand(r)
    return r._moog_expanded


def _expand(r):
    from .. import _symbolic
    try:
        _symbolic.trace_rule_step(r.step)
        return [r]               # the simple form (first sprite of one layer): lowered where the rule table is filled
    except (NotImplementedError, AttributeError, TypeError):
        pass
    try:
        n, mods = _symbolic.trace_rule_zip(r.step)
    except (NotImplementedError, AttributeError, TypeError) as exc:
        raise NotImplementedError('game rule %r has no device lowering (see game_rules.register_lowering): %s'
                                  % (type(r).__name__, exc))
    draws = [_RuleDraws(k, min(2, n - k)) for k in range(0, n, 2)]
    return draws + [_ModifyTraced(layer, mod, vec, draws) for layer, mod, vec in mods]


class KeepNearCenter(AbstractRule):
    """re_center.py:11-58: when the first sprite of agent_layer strays more than a grid cell
    from (0.5, 0.5), every sprite of layers_to_center (and the agent) is shifted back by one
    cell."""

    def __init__(self, agent_layer, layers_to_center, grid_x, grid_y=None):
        self._agent_layer = agent_layer
        if agent_layer not in set(layers_to_center):
            layers_to_center = list(layers_to_center) + [agent_layer]
        self._layers_to_center = list(layers_to_center)
        self._grid_cell = (float(grid_x), float(grid_x if grid_y is None else grid_y))


class _NoAttributes(object):
    """Probe argument: a filter that returns a constant without looking at the sprite."""

    def __getattr__(self, name):
        raise _TouchedSprite(name)


class _TouchedSprite(Exception):
    pass


def _classify_filter(filter_fn):
    """(MOOG_FILTER_*, expression node or None) of a `sprite -> bool` function.  None and
    functions that ignore their argument and return True (e.g. `lambda _: True`) are ALWAYS;
    anything else is traced symbolically (moog/_symbolic.py)."""
    if filter_fn is None:
        return _abi.MOOG_FILTER_ALWAYS, None
    try:
        if filter_fn(_NoAttributes()) is True:
            return _abi.MOOG_FILTER_ALWAYS, None
    except Exception:  # pylint: disable=broad-except
        pass
    from .. import _symbolic
    return _abi.MOOG_FILTER_EXPR, _symbolic.trace_value(filter_fn, 1)


class Vanish(AbstractRule):
    """vanish.py:9-39, the base of the vanishing rules: a subclass names the sprites of `layer` to remove through
    `_get_vanish_inds(state)`.  On the device sprites are named by a filter over their own attributes, so a config-local
    subclass is lowered when its `_get_vanish_inds` is the reference's own pattern -- the indices of the sprites of the
    layer that pass a test (`[i for i, s in enumerate(state[layer]) if test(s)]`): it is traced once over a probe layer
    (game_rules.expand_local_rule) into a VanishByFilter.  Anything else (indices that depend on other layers, on the
    order of the list) is refused at construction of the environment with the reason."""

    def __init__(self, layer):
        self._layer = layer

    def _get_vanish_inds(self, state):
        raise NotImplementedError

    def step(self, state, meta_state):
        raise RuntimeError('game rules are stepped by the engine, not on the host')


class VanishByFilter(AbstractRule):
    """vanish.py:42-61: every sprite of `layer` for which filter_fn is True is popped."""

    def __init__(self, layer, filter_fn=None):
        self._layer = layer
        self._filter_fn = filter_fn


class ChangeLayer(AbstractRule):
    """change_layer.py:16-46: sprites of old_layer passing filter_fn are popped and appended
    to new_layer, in order."""

    def __init__(self, old_layer, new_layer, filter_fn=None):
        self._old_layer, self._new_layer = old_layer, new_layer
        self._filter_fn = filter_fn


class CreateSprites(AbstractRule):
    """create_sprites.py:10-37: `generator(without_overlapping=<sprites of those layers>)`
    at rule time; the new sprites are appended to `layer`."""

    def __init__(self, layer, generator, without_overlapping=()):
        self._layer = layer
        self._generator = generator
        if isinstance(without_overlapping, str):
            without_overlapping = (without_overlapping,)
        self._without_overlapping = tuple(without_overlapping)


def _probe_randint(fn, patterns=None):
    """Calls fn() twice with np.random.randint replaced -- once returning its low bound, once its highest value -- and
    returns ([(low, high) of every draw], value at the low bounds, value at the high bounds).  `patterns` (a list of tuples
    of 0 / 1, one entry per draw: low bound / highest value; draws beyond a tuple's end take the low bound): fn() is called
    once per pattern instead and the values come back as a list -- how a caller checks which draw a value depends on.

    The reference calls fn() at every reset (timing.py:47): a callable may only be lowered when everything random in it
    is seen here.  So every other entry point of np.random and of the `random` module raises while fn runs, and a
    callable that got at a generator behind the probe's back (`from numpy.random import randint`, a RandomState of its
    own is fine only if it is deterministic) is caught by the global generators' state having moved, or by the two
    calls disagreeing without a probed draw.  The host generators are left exactly as they were."""
    import random as py_random
    real = np.random.randint
    np_state, py_state = np.random.get_state(), py_random.getstate()

    def refuse(name):
        def _refuse(*args, **kwargs):
            raise NotImplementedError('a rule interval that draws from %s is not lowered (np.random.randint is)' % name)
        return _refuse
    np_names = [n for n in ('rand', 'randn', 'random', 'random_sample', 'ranf', 'sample', 'uniform', 'normal', 'choice',
                            'permutation', 'shuffle', 'random_integers', 'binomial', 'poisson', 'exponential', 'beta',
                            'gamma', 'geometric', 'triangular', 'standard_normal', 'bytes') if hasattr(np.random, n)]
    py_names = [n for n in ('random', 'uniform', 'randint', 'randrange', 'choice', 'choices', 'sample', 'shuffle', 'gauss',
                            'normalvariate', 'triangular', 'betavariate', 'expovariate', 'getrandbits') if hasattr(py_random, n)]
    saved = [(np.random, n, getattr(np.random, n)) for n in np_names] + [(py_random, n, getattr(py_random, n)) for n in py_names]
    out = []
    try:
        for mod, n, _ in saved:
            setattr(mod, n, refuse(('np.random.' if mod is np.random else 'random.') + n))
        for pattern in (patterns if patterns is not None else ((0,) * 64, (1,) * 64)):
            seen = []

            def probe(low, high=None, size=None, dtype=int, _seen=seen, _pat=pattern):
                if high is None:
                    low, high = 0, low
                if size is not None:
                    raise NotImplementedError('np.random.randint with a size in a rule\'s interval')
                hi_pick = len(_seen) < len(_pat) and bool(_pat[len(_seen)])
                _seen.append((int(low), int(high)))
                return int(high) - 1 if hi_pick else int(low)
            np.random.randint = probe
            try:
                value = fn()
            finally:
                np.random.randint = real
            out.append((seen, value))
    finally:
        for mod, n, f in saved:
            setattr(mod, n, f)
        moved = (not all(np.array_equal(a, b) for a, b in zip(np_state, np.random.get_state()))) or py_state != py_random.getstate()
        np.random.set_state(np_state)
        py_random.setstate(py_state)
    if moved:
        raise NotImplementedError('a rule interval that draws from the host\'s random generators other than through '
                                  'np.random.randint is not lowered')
    if any(o[0] != out[0][0] for o in out[1:]):
        raise NotImplementedError('a rule interval whose draws depend on each other')
    if patterns is not None:
        return out[0][0], [o[1] for o in out]
    if not out[0][0]:
        v0, v1 = np.asarray(out[0][1], float), np.asarray(out[1][1], float)
        if v0.shape != v1.shape or not np.array_equal(v0, v1):
            raise NotImplementedError('a rule interval that changes from call to call without an np.random.randint draw '
                                      'is not lowered')
    return out[0][0], out[0][1], out[1][1]


class TimedRule(AbstractRule):
    """timing.py:18-59: steps `rules` while _steps_until_start <= 0 < _steps_until_stop;
    both count down once per call.  A callable `step_interval` (timing.py:28-31,47) is drawn whenever the rule is reset;
    the forms that are lowered: ONE np.random.randint draw -- a random start with a fixed width (stop - start does not
    depend on the draw; DelayedRule(lambda: np.random.randint(a, b), ...)) or a fixed start with a random stop
    (TemporaryRule(lambda: np.random.randint(a, b), ...)) -- and TWO: a random start and a random duration
    (DelayedRule(lambda: np.random.randint(a, b), ..., duration=lambda: np.random.randint(c, d)), timing.py:84-86).  The
    draws are made on the device, in the reference's order."""

    def __init__(self, step_interval, rules):
        self._random = None   # (op, p0, p1, p2) of MOOG_RULE_TIMED
        if callable(step_interval):
            draws, lo, hi = _probe_randint(step_interval)
            lo = (float(lo[0]), float(lo[1]))
            hi = (float(hi[0]), float(hi[1]))
            if not draws:
                step_interval = lo
            elif len(draws) == 2:
                # timing.py:84-86 DelayedRule(start=<callable>, duration=<callable>): the start is drawn first, then the duration;
                # lowered when the start is the first draw itself and stop - start the second (checked on all four corners)
                (a, b), (c, d) = draws
                _, corners = _probe_randint(step_interval, patterns=((0, 0), (1, 1), (1, 0), (0, 1)))
                want = [(a, a + c), (b - 1, b - 1 + d - 1), (b - 1, b - 1 + c), (a, a + d - 1)]
                if [(float(v[0]), float(v[1])) for v in corners] != [(float(x), float(y)) for x, y in want]:
                    raise NotImplementedError('a callable step_interval with two np.random.randint draws other than '
                                              '(randint, randint + randint) is not lowered')
                self._random = (3, float(a), float(c), float(b), int(d))   # start = randint(a, b), stop = start + randint(c, d)
                step_interval = lo
            elif len(draws) != 1:
                raise NotImplementedError('a callable step_interval with more than two np.random.randint draws is not lowered')
            else:
                a, b = draws[0]
                if lo[0] == a and hi[0] == b - 1 and (lo[1] - lo[0] == hi[1] - hi[0] or (np.isinf(lo[1]) and np.isinf(hi[1]))):
                    self._random = (1, float(a), lo[1] - lo[0], float(b))          # start = randint(a, b), width fixed
                elif lo[0] == hi[0] and lo[1] == a and hi[1] == b - 1:
                    self._random = (2, lo[0], float(a), float(b))                  # stop = randint(a, b), start fixed
                else:
                    raise NotImplementedError('a callable step_interval other than (randint, randint + c) / (c, randint) is not lowered')
                step_interval = lo
        self._step_interval = (float(step_interval[0]), float(step_interval[1]))
        if not isinstance(rules, (list, tuple)):
            rules = (rules,)
        self._rules = tuple(rules)


class DelayedRule(TimedRule):
    """timing.py:62-90"""

    def __init__(self, steps_until_start, rules, duration=np.inf):
        if callable(steps_until_start) or callable(duration):
            start = steps_until_start if callable(steps_until_start) else (lambda: steps_until_start)
            dur = duration if callable(duration) else (lambda: duration)

            def interval():   # timing.py:84-86: the start is drawn first
                t0 = start()
                return (t0, t0 + dur())
            super(DelayedRule, self).__init__(interval, rules)
        else:
            super(DelayedRule, self).__init__((steps_until_start, steps_until_start + duration), rules)


class TemporaryRule(TimedRule):
    """timing.py:93-109"""

    def __init__(self, steps_until_stop, rules):
        if callable(steps_until_stop):
            super(TemporaryRule, self).__init__(lambda: (0, steps_until_stop()), rules)
        else:
            super(TemporaryRule, self).__init__((0, steps_until_stop), rules)


class Phase(AbstractRule):
    """task_phases.py:17-98: one-time rules on the phase's first step, continual rules every
    step, until `duration` steps have passed or end_condition(state[, meta_state]) holds."""

    def __init__(self, one_time_rules=(), continual_rules=(), end_condition=None, duration=np.inf,
                 name=''):
        if not isinstance(one_time_rules, (list, tuple)):
            one_time_rules = (one_time_rules,)
        if not isinstance(continual_rules, (list, tuple)):
            continual_rules = (continual_rules,)
        self._one_time_rules, self._continual_rules = tuple(one_time_rules), tuple(continual_rules)
        self._end_condition = end_condition
        self._random_duration = None
        if callable(duration):
            # task_phases.py:72 draws `duration()` whenever the phase is reset; the one form the configs use,
            # `lambda: np.random.randint(lo, hi)` (multi_tracking_with_feature.py:233), is drawn on the device
            real, seen = np.random.randint, []

            def probe(low, high=None, size=None, dtype=int):
                if high is None:
                    low, high = 0, low
                seen.append((int(low), int(high), size))
                return int(low)
            np.random.randint = probe
            try:
                value = duration()
            finally:
                np.random.randint = real
            if len(seen) != 1 or seen[0][2] is not None or value != seen[0][0]:
                raise NotImplementedError('a callable Phase duration other than np.random.randint(lo, hi)')
            self._random_duration = seen[0][:2]
            duration = seen[0][0]
        self._duration = float(duration)
        self._name = name

    @property
    def name(self):
        return self._name

    @property
    def _rules(self):   # children in the order the device walks them
        return self._one_time_rules + self._continual_rules


class PhaseSequence(AbstractRule):
    """task_phases.py:101-141: the phases one after the other.  With
    meta_state_phase_name_key the reference publishes the current phase's name in the
    meta-state; conditions that test it (`meta_state[key] == 'name'`) are lowered to a test of
    this rule's phase index."""

    def __init__(self, *single_phases, meta_state_phase_name_key=None):
        for ph in single_phases:
            if not isinstance(ph, Phase):
                raise TypeError('PhaseSequence takes Phase instances')
        self._phases = tuple(single_phases)
        self._meta_state_key = meta_state_phase_name_key

    @property
    def _rules(self):
        return self._phases


class Fixation(AbstractRule):
    """fixation.py:17-58: counts the consecutive steps during which the first sprite of `agent_layer` stays within
    `fixation_threshold` of the first sprite of `fixation_layer`.  The reference keeps the count in
    meta_state[meta_state_fixation_key]; here it is the rule's state scalar on the device, and traced conditions that
    read `meta_state[key]` (e.g. `>= 15`, multi_tracking_with_feature.py:205-206) read that scalar."""

    def __init__(self, agent_layer, fixation_layer, fixation_threshold=0.1,
                 meta_state_fixation_key='fixation_duration'):
        self._agent_layer = agent_layer
        self._fixation_layer = fixation_layer
        self._fixation_threshold = fixation_threshold
        self._meta_state_fixation_key = meta_state_fixation_key


class _ContactCounter(object):
    """contact_rules.get_contact_counter / get_contact_indices (:15-56): a `state -> int`
    condition for ConditionalRule, evaluated on the device (number of overlapping pairs)."""

    def __init__(self, layer_0, layer_1):
        self.layer_0, self.layer_1 = layer_0, layer_1

    def __call__(self, state):
        raise RuntimeError('contact counters are evaluated by the engine, not on the host')


def get_contact_counter(layer_0, layer_1):
    return _ContactCounter(layer_0, layer_1)


class _ContactIndices(_ContactCounter):
    """contact_rules.get_contact_indices (:15-37): `state -> [(i_0, i_1), ...]`.  As a ConditionalRule condition only its
    truth value matters (an empty list is false), which is the contact counter's: lowered the same way.  The list
    itself exists on the device only (RawState observers read sprites, not index pairs)."""


def get_contact_indices(layer_0, layer_1):
    return _ContactIndices(layer_0, layer_1)


class _BinomialProbe(object):
    def __init__(self, n, p):
        self.n, self.p = n, p


class ConditionalRule(AbstractRule):
    """conditional.py:16-63: steps `rules` condition(state) times.  Lowered conditions:
    `lambda state: np.random.binomial(1, p)` (first_person_predators_prey.py:181,189),
    recognised by running the condition once with np.random.binomial recording its
    arguments."""

    def __init__(self, condition, rules):
        self._condition = condition
        if not isinstance(rules, (list, tuple)):
            rules = [rules]
        self._rules = tuple(rules)

    def classify(self):
        return classify_condition(self._condition)


def classify_condition(condition):
    """Lowers a `state[, meta_state] -> bool / int` condition (ConditionalRule, Phase end
    conditions).  Returns (MOOG_RCOND_*, p0, layers (l0, l1) or None, expression node or None)."""
    import inspect
    from .. import _symbolic
    if isinstance(condition, _ContactCounter):
        return _abi.MOOG_RCOND_CONTACT_COUNT, 0., (condition.layer_0, condition.layer_1), None
    nargs = len(inspect.signature(condition).parameters)
    real = np.random.binomial
    np.random.binomial = lambda n, p, size=None: _BinomialProbe(n, p)
    try:
        out = condition(*([None] * nargs))
    except Exception:  # pylint: disable=broad-except
        out = None
    finally:
        np.random.binomial = real
    if isinstance(out, _BinomialProbe):
        if out.n == 1:
            return _abi.MOOG_RCOND_BERNOULLI, float(out.p), None, None
        raise NotImplementedError('condition is np.random.binomial(n, p) with n != 1')
    # reads the state: all / any over a layer, the layer's first sprite, or the current phase
    kind, layer, node = _symbolic.trace_state_condition(condition, with_meta=(nargs == 2))
    code = {'all': _abi.MOOG_RCOND_ALL_EXPR, 'any': _abi.MOOG_RCOND_ANY_EXPR,
            'first': _abi.MOOG_RCOND_FIRST_EXPR, 'plain': _abi.MOOG_RCOND_STATE_EXPR,
            'count': _abi.MOOG_RCOND_COUNT_EXPR}[kind]
    return code, 0., ((layer, layer) if layer is not None else None), node


# ---- lowering registry for config-local rule classes --------------------------
_LOWERINGS = {}


def register_lowering(class_name, fn):
    """fn(rule, layer_index) -> dict(kind=..., l0=..., l1=..., p0=..., p1=..., p2=...)"""
    _LOWERINGS[class_name] = fn


def lookup_lowering(rule):
    for klass in type(rule).__mro__:
        if klass.__name__ in _LOWERINGS:
            return _LOWERINGS[klass.__name__]
    return None


def _lower_booster(rule, layer_index):
    # functional_maze.py:23-78
    return dict(kind=_abi.MOOG_RULE_BOOSTER,
                l0=layer_index(rule._agent_layer), l1=layer_index(rule._booster_layer),
                p0=float(rule._mass_multiplier), p1=float(rule._c2_multiplier),
                p2=float(rule.boost_duration))


register_lowering('Booster', _lower_booster)

def _submodules(**modules):
    """The reference keeps one class per file (`from moog.physics import collisions`, `moog.game_rules.vanish.Vanish`);
    here a package is one file, and those module paths are aliases that hold the same objects."""
    import sys
    import types
    for name, names in modules.items():
        m = types.ModuleType(__name__ + '.' + name)
        m.__doc__ = 'Alias module: the reference\'s moog/%s/%s.py (names defined in %s).' % (__name__.split('.')[-1], name, __name__)
        for n in names:
            setattr(m, n, globals()[n])
        sys.modules[m.__name__] = m
        globals()[name] = m

_submodules(abstract_rule=('AbstractRule',), change_layer=('ChangeLayer',), conditional=('ConditionalRule',), contact_rules=('get_contact_counter', 'get_contact_indices', 'ModifyOnContact'), create_sprites=('CreateSprites',), fixation=('Fixation',), modify_meta_state=('ModifyMetaState', 'UpdateMetaStateValue'), modify_sprites=('ModifySprites',), portal=('Portal',), re_center=('KeepNearCenter',), task_phases=('Phase', 'PhaseSequence'), timing=('DelayedRule', 'TemporaryRule', 'TimedRule'), vanish=('Vanish', 'VanishByFilter', 'VanishOnContact'))
