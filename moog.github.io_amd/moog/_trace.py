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


class HExprOp(GenOp):
    """A value the initializer computed from its draws and uses more than once (the outputs of a sorting network over
    drawn values, match_to_sample.py:46): evaluated once per reset into direct-draw slot `index`, read like a draw."""

    def __init__(self, index, node, seq, tagged=False):
        GenOp.__init__(self, None, 0, 0, False, [], 0, [])
        self.index, self.node, self.seq, self.tagged = index, node, seq, tagged
        from . import _abi
        self.cell = (_abi.MOOG_CELL_HEXPR, index)


class PStateOp(GenOp):
    """A number the initializer's object keeps ACROSS episodes and updates at every reset but the first
    (predators_arena.py:56,88-94: the auto-curriculum's predator mass): lives in a per-env slot that resets never clear;
    `node` is the update over Node('pstate', name), `init` the value of the first episode."""

    def __init__(self, name, init, node):
        GenOp.__init__(self, None, 0, 0, False, [], 0, [])
        self.name, self.init, self.node = name, float(init), node
        from . import _abi
        self.cell = (_abi.MOOG_CELL_PSTATE, -1)   # (the slot's rule index is filled in by the compiler)


class SimOp(GenOp):
    """The initializer runs the episode's physics forward to learn something about the trial (bounce_box_contact_
    prediction.py:40-50: `while True: if <test>: return ...; physics.step(state)`).  `node`: 0 while the loop goes on,
    k > 0 when it leaves through its k-th exit (filled in by the compiler from all explored paths); the exit taken goes
    to direct-draw slot `index`."""

    def __init__(self, index, seq):
        GenOp.__init__(self, None, 0, 0, False, [], 0, [])
        self.index, self.seq, self.node = index, seq, None
        from . import _abi
        self.cell = (_abi.MOOG_CELL_SIMULATE, index)


class StoreOp(GenOp):
    """`sprite.position = ...` / `sprite.velocity = ...` on an already built sprite (bounce_box_contact_prediction.py:
    117-119: the targets are put back where they started once the look-ahead is done)."""

    def __init__(self, sprite, stores, vec):
        GenOp.__init__(self, None, 0, 0, False, [], 0, [])
        self.sprite, self.stores, self.vec = sprite, stores, vec
        from . import _abi
        self.cell = (_abi.MOOG_CELL_STORE, -1)


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


class Restarted(Exception):
    """The traced initializer called itself (`return state_initializer()`): the path being explored starts over."""


class Tracer(object):
    def __init__(self):
        self.init_code = None         # code object of the initializer being traced (recursion = it starts over)
        self.len_observed = set()     # fail_gracefully ops whose result's length / truth the initializer asked for
        self.short_op = None          # the op this run pretends came out one sprite short
        self.collected, self.replay, self.replay_i = [], None, 0
        self.ops = []               # GenOp, in randomness-consumption order
        self.op_of = {}             # id(sprite) -> (op, k)
        self.randint_calls = []
        self.n_hdraws = 0
        self.alias = {}             # id(placeholder of a later sample_generator alternative) -> the first alternative's
        self.alias_keep = []        # (keeps those placeholders alive: ids must stay unique)
        self.maze = None            # the per-reset random maze of the initializer (maze_lib/_traced.py)
        self.seq = 0                # order of sampling events (deferred factor samples and direct draws)
        # rejection loops over a direct draw (`while not ok: a = np.random.uniform(..); ok = test(a)`, match_to_sample.py:
        # 33-43): the draw they redo, the accept conditions seen so far, the condition being probed
        self.retry_op = None
        self.retry_confirmed = []
        self.retry_probe = None
        self.retry_reuse = False
        # a look-ahead loop that steps the physics inside the initializer (SimOp): explored path by path
        self.live = False           # sprites are read as they ARE (their slots), no longer as their recipes
        self.sim_op = None
        self.sim_forced = []        # answers of the first pass through the loop body (the path this run explores)
        self.sim_exit_plan = None   # answers that leave the loop (a known exit), used after this run's first physics step
        self.sim_trail = []         # (node, answer) of the first pass
        self.sim_steps = 0
        self.sim_pass_i = 0
        self.sim_seen = {}          # answers given in the current pass, by test

    def next_seq(self):
        self.seq += 1
        return self.seq

    def hdraw(self):
        """A new direct draw: a symbolic uniform in [0, 1)."""
        from . import _abi, _symbolic
        self.check_restart()
        if getattr(self, 'suspend', False) == 'collect':   # a draw of a config-local distribution inside generate_sprites
            if self.replay is not None:
                if self.replay_i >= len(self.replay):
                    raise NotImplementedError('a distribution that takes a varying number of draws')
                k = self.replay[self.replay_i][0]
                self.replay_i += 1
                return _symbolic.Sym(_symbolic.Node('hdraw', k))
            if self.n_hdraws >= _abi.MOOG_MAX_HDRAWS:
                raise NotImplementedError('more than %d direct np.random draws per reset' % _abi.MOOG_MAX_HDRAWS)
            k = self.n_hdraws
            self.n_hdraws += 1
            self.collected.append((k, self.next_seq()))
            return _symbolic.Sym(_symbolic.Node('hdraw', k))
        if getattr(self, 'suspend', False):
            raise NotImplementedError('np.random calls inside a distribution sampled by generate_sprites')
        if self.retry_reuse:   # the loop body of a rejection loop runs again: the same draw, taken anew on the device
            self.retry_reuse = False
            return _symbolic.Sym(_symbolic.Node('hdraw', self.retry_op.index))
        if self.retry_probe is not None:
            raise _symbolic.Unsupported('a rejection loop that takes more than one np.random draw per try')
        if self.n_hdraws >= _abi.MOOG_MAX_HDRAWS:
            raise NotImplementedError('more than %d direct np.random draws per reset' % _abi.MOOG_MAX_HDRAWS)
        k = self.n_hdraws
        self.n_hdraws += 1
        op = HDrawOp(k, self.next_seq())
        op.accept = []
        self.add_op(op)
        self.retry_op, self.retry_confirmed, self.retry_probe = op, [], None
        return _symbolic.Sym(_symbolic.Node('hdraw', k))

    def let(self, node, tagged=False):
        """A computed value kept in a direct-draw slot (evaluated once per reset, in op order).  tagged: a copy of a live
        sprite attribute (np.copy(sprite.velocity)) keeps its numpy dtype in a second cell -- a float32 velocity put back
        later is a float32 array again."""
        from . import _abi, _symbolic
        if self.retry_probe is not None:
            raise _symbolic.Unsupported('a rejection loop with side effects')
        n = 2 if tagged else 1
        if self.n_hdraws + n > _abi.MOOG_MAX_HDRAWS:
            raise NotImplementedError('more than %d direct np.random draws / computed values per reset' % _abi.MOOG_MAX_HDRAWS)
        k = self.n_hdraws
        self.n_hdraws += n
        self.add_op(HExprOp(k, node, self.next_seq(), tagged))
        return _symbolic.Sym(_symbolic.Node('hdrawt' if tagged else 'hdraw', k))

    def retry_decide(self, node):
        """bool() of a value computed from the LATEST direct draw inside the initializer.  The only control flow that is
        lowered is the rejection loop: when the test fails the code draws again and tests again.  Probed once: the
        first bool() answers False; the code must then ask for a draw (handed the same symbolic draw again) and
        arrive at the structurally same test, which answers True and becomes an accept condition of that draw -- the
        device redraws until all of a draw's conditions hold (one uniform per try, as in the reference)."""
        from . import _symbolic
        op = self.retry_op
        found = set()

        def walk(n):
            if n.op == 'hdraw':
                found.add(n.args[0])
            for a in n.args:
                if isinstance(a, _symbolic.Node):
                    walk(a)
        walk(node)
        if op is None or self.ops[-1] is not op or op.index not in found or max(found) != op.index:
            raise _symbolic.Unsupported('branching on a value drawn at reset time other than the accept test of a '
                                        'rejection loop over the latest np.random draw')
        key = node.key()
        if key in self.retry_confirmed:
            return True
        if self.retry_probe is not None:
            if key != self.retry_probe or self.retry_reuse:
                raise _symbolic.Unsupported('branching on a drawn value that is not a rejection loop (the failed test '
                                            'was not repeated on a fresh draw)')
            self.retry_probe = None
            self.retry_confirmed.append(key)
            op.accept.append(node)
            return True
        self.retry_probe, self.retry_reuse = key, True
        return False

    def check_restart(self):
        """Raises Restarted when the traced initializer has called itself (`return state_initializer()`, red_green.py:
        155,203): two frames of its code on the stack.  Called where the initializer does something traceable."""
        if self.init_code is None:
            return
        import sys
        f, depth = sys._getframe(1), 0
        while f is not None:
            if f.f_code is self.init_code:
                depth += 1
                if depth > 1:
                    raise Restarted()
            f = f.f_back

    def go_live(self):
        if self.retry_probe is not None:
            from . import _symbolic
            raise _symbolic.Unsupported('a rejection loop with side effects')
        self.live = True

    def sim_decide(self, node):
        """bool() of a value read off the live sprites inside the initializer: a test of the look-ahead loop."""
        from . import _abi, _symbolic
        if self.sim_op is None:
            if self.n_hdraws + 2 > _abi.MOOG_MAX_HDRAWS:
                raise NotImplementedError('more than %d direct np.random draws / computed values per reset' % _abi.MOOG_MAX_HDRAWS)
            self.sim_op = SimOp(self.n_hdraws, self.next_seq())   # two cells: the exit taken, the loop counter
            self.n_hdraws += 2
            self.add_op(self.sim_op)
        elif self.ops[-1] is not self.sim_op:
            raise _symbolic.Unsupported('tests on live sprites after the look-ahead loop of a state_initializer')
        key = node.key()
        if key in self.sim_seen:   # the same test again within one pass (`0 if red_overlap else 1`, red_green.py:110)
            return self.sim_seen[key]
        if self.sim_steps == 0:   # the path this run explores
            i = len(self.sim_trail)
            v = self.sim_forced[i] if i < len(self.sim_forced) else True
            self.sim_trail.append((node, v))
        else:                     # a later pass: leave through a known exit
            plan = self.sim_exit_plan
            if plan is None or self.sim_pass_i >= len(plan):
                raise _symbolic.Unsupported('a look-ahead loop whose first branch does not leave the loop')
            v = plan[self.sim_pass_i]
            self.sim_pass_i += 1
        self.sim_seen[key] = v
        return v

    def sim_step(self):
        """physics.step(state) inside the initializer."""
        from . import _symbolic
        self.go_live()
        if self.sim_op is None or self.ops[-1] is not self.sim_op:
            raise _symbolic.Unsupported('physics.step(state) in a state_initializer outside a look-ahead loop with an exit test')
        self.sim_steps += 1
        self.sim_pass_i = 0
        self.sim_seen = {}
        if self.sim_steps > 2:
            raise _symbolic.Unsupported('a look-ahead loop that the known exit does not leave')

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
        if self.retry_probe is not None:
            from . import _symbolic
            raise _symbolic.Unsupported('a rejection loop with side effects')
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
    t.check_restart()
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


class _SimRange(object):
    """`for step in range(n)` around the look-ahead of an initializer (red_green.py:101): the counter is symbolic (the
    device counts physics steps in a cell), and 'the range is used up' is one more exit test of the loop."""

    def __init__(self, t, n):
        self.t, self.n = t, int(n)

    def __iter__(self):
        return self

    def __next__(self):
        from . import _symbolic
        step = _symbolic.Sym(_symbolic.Node('simstep'))
        if bool(step >= self.n):
            raise StopIteration
        return step


def traced_range(*args):
    """Stands in for the builtin `range` in the config module while its initializer is traced: once the initializer
    looks at live sprites, a one-argument range is the look-ahead's step loop."""
    t = _ACTIVE
    if t is not None and t.live and len(args) == 1 and t.sim_steps == 0 and (t.sim_op is None or t.ops[-1] is t.sim_op) \
            and not getattr(t, 'suspend', False):
        return _SimRange(t, args[0])
    return range(*args)


@contextlib.contextmanager
def tracing(sim_forced=(), sim_exit_plan=None, short_op=None):
    global _ACTIVE
    t = Tracer()
    t.sim_forced, t.sim_exit_plan, t.short_op = list(sim_forced), sim_exit_plan, short_op
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
    real_sort = np.sort

    def fake_sort(a, *args, **kwargs):   # np.sort of a list with drawn values: a sorting network over computed cells
        from . import _symbolic
        items = list(a.items) if isinstance(a, _symbolic.SymVec) else (list(a) if isinstance(a, (list, tuple)) else None)
        if items is None or not any(isinstance(v, _symbolic.Sym) for v in items):
            return real_sort(a, *args, **kwargs)
        if args or kwargs:
            raise NotImplementedError('np.sort(..., axis / kind / order) of drawn values')
        return _symbolic.sort_network(items, t.let)
    np.sort = fake_sort
    try:
        yield t
        if t.retry_probe is not None:
            from . import _symbolic
            raise _symbolic.Unsupported('branching on a drawn value that is not a rejection loop')
    finally:
        np.sort = real_sort
        np.random.randint = real_randint
        for name, fn in blocked.items():
            setattr(np.random, name, fn)
        _ACTIVE = prev
