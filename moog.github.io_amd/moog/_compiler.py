"""Lowering of a MOOG config dict to the engine's `moog_program_t`.

Input: the kwargs of `moog.environment.Environment.__init__`
(reference environment.py:28-80): state_initializer, physics, task,
action_space, observers, game_rules.  Output: a filled `_abi.Program` plus the
layer-name table.  The state_initializer closure is traced once (see _trace.py);
every other component is a parameter record.
"""
import collections

import numpy as np

from . import _abi
from . import _distcode
from . import _symbolic
from . import _trace
from . import action_spaces
from . import game_rules as rules_lib
from . import observers as observers_lib
from . import physics as physics_lib
from . import shapes as shapes_lib
from . import sprite as sprite_lib
from . import tasks as tasks_lib
from .observers import polygon_modifiers
from .state_initialization import distributions as distribs

Compiled = collections.namedtuple(
    'Compiled', ['program', 'layer_names', 'layer_slots', 'observer_key', 'layout', 'shape_names', 'rule_ref_index',
                 'pstate_slots', 'dynamic_meta', 'color_fn', 'layer_n_init'])


class _ShapeTable(object):
    def __init__(self, program):
        self.p = program
        self.entries = []   # (key, id)
        self.nverts_total = 0

    def intern(self, shape):
        """Returns shape id for a named shape or a raw [n,2] vertex array."""
        if isinstance(shape, str):
            if shape not in shapes_lib.SHAPES:
                raise ValueError('unknown shape name %r' % (shape,))
            key = ('name', shape)
            raw = shapes_lib.SHAPES[shape]
        else:
            raw = np.asarray(shape, dtype=np.float64)
            key = ('raw', raw.shape, raw.tobytes())
        for k, sid in self.entries:
            if k == key:
                return sid
        sid = len(self.entries)
        if sid >= _abi.MOOG_MAX_SHAPES:
            raise ValueError('too many distinct shapes (max %d)' % _abi.MOOG_MAX_SHAPES)
        centred, centroid, inertia = sprite_lib.shape_record(raw)
        n = centred.shape[0]
        if self.nverts_total + n > _abi.MOOG_MAX_SHAPE_VERTS:
            raise ValueError('shape table overflow')
        rec = self.p.shapes[sid]
        rec.nverts = n
        rec.voff = self.nverts_total
        rec.is_circle = int(isinstance(shape, str) and shape == 'circle')
        rec.centroid[0], rec.centroid[1] = centroid
        rec.inertia[0], rec.inertia[1] = inertia
        for k in range(n):
            self.p.shape_verts[self.nverts_total + k][0] = centred[k, 0]
            self.p.shape_verts[self.nverts_total + k][1] = centred[k, 1]
        self.nverts_total += n
        self.entries.append((key, sid))
        self.p.n_shapes = len(self.entries)
        return sid

    def nverts(self, sid):
        return self.p.shapes[sid].nverts


def _as_list(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


def _fill_layers(dst, names, layer_index):
    names = _as_list(names)
    if len(names) > _abi.MOOG_MAX_LAYERS:
        raise ValueError('too many layers in one component')
    for i, n in enumerate(names):
        dst[i] = layer_index(n)
    return len(names)


_REF_COUNTER, _REF_INDEX = [0], {}


def _flatten_rules(rules, parent=-1, depth=0, out=None):
    """Pre-order list of (rule, parent index) over TimedRule / ConditionalRule nesting."""
    out = [] if out is None else out
    expanded = [(k, r) for r0 in rules for k, r in enumerate(rules_lib.expand_local_rule(r0))]
    for k, r in expanded:
        if getattr(r, 'host_side', False):
            continue
        if k == 0:   # (position of the config's own rule in a pre-order walk of its rule objects: fixtures use it)
            _REF_COUNTER[0] += 1
        _REF_INDEX[id(r)] = _REF_COUNTER[0] - 1 if k == 0 else -1
        out.append((r, parent))
        if isinstance(r, (rules_lib.TimedRule, rules_lib.ConditionalRule, rules_lib.Phase,
                          rules_lib.PhaseSequence)):
            if depth >= 2:
                raise NotImplementedError('rule combinators nested deeper than 2')
            _flatten_rules(r._rules, len(out) - 1, depth + 1, out)
    return out


def _reject_f32_velocity(state, layers, what):
    """The maze components restate numpy's float64 arithmetic only: velocities sampled by Continuous are float32
    arrays in the reference (distributions.py:81,99)."""
    from .state_initialization import distributions as distribs
    if not isinstance(layers, (list, tuple)):
        layers = [layers]
    for name in layers:
        for sp in state[name]:
            for k in ('x_vel', 'y_vel'):
                f = sp.factors[k]
                if isinstance(f, sprite_lib.SymbolicFactor) and isinstance(f.dist, distribs.Continuous):
                    raise NotImplementedError('%s over sprites whose velocity is sampled by Continuous' % what)


def _trace_initializer(state_initializer):
    """Traces the state_initializer.  An initializer that steps the physics in a loop to look ahead (bounce_box_contact_
    prediction.py:40-50,113-119; red_green.py:92-116,193-203) is run once per path through the loop body -- the tracer
    answers the body's tests on live sprites from a forced list, as moog/_symbolic.py explores config lambdas -- and the
    paths are merged: the loop becomes one SimOp whose expression says which exit (if any) the current state takes; what
    the exits hand to the rest of the initializer (sprite metadata) becomes a value selected by the exit taken; an exit
    after which the initializer calls itself again is a restart of the whole reset.  Generators that may come out short
    (fail_gracefully) and whose result the initializer measures are probed the same way."""
    glob = getattr(state_initializer, '__globals__', None)
    had_range = glob is not None and 'range' in glob
    old_range = glob.get('range') if had_range else None
    if glob is not None:
        glob['range'] = _trace.traced_range

    code = getattr(getattr(state_initializer, '__func__', state_initializer), '__code__', None)

    def run(**kw):
        with _trace.tracing(**kw) as t:
            t.init_code = code
            try:
                st = state_initializer()
            except _trace.Restarted:
                st = None
        return t, st
    try:
        tr, state = run()
        if tr.sim_op is None and state is None:
            raise NotImplementedError('a state_initializer that always starts over')
        if tr.sim_op is not None:
            tr, state = _explore_look_ahead(run, tr, state)
        # generators whose short result the initializer reacts to
        for key in sorted(tr.len_observed):
            t2, st2 = run(short_op=key)
            if st2 is None:
                tr.ops[key].restart_if_short = True
            elif [type(op) for op in t2.ops] != [type(op) for op in tr.ops]:
                raise NotImplementedError('a state_initializer that builds a different state when a generator comes out short')
        return tr, state
    finally:
        if glob is not None:
            if had_range:
                glob['range'] = old_range
            else:
                glob.pop('range', None)


def _explore_look_ahead(run, tr, state):
    def index_sprites(st):
        return {id(sp): (name, i) for name, sprites in st.items() for i, sp in enumerate(sprites)}

    def canon(node, where):   # sprite objects of a run -> (layer, index)
        args = []
        for a in node.args:
            if isinstance(a, _symbolic.Node):
                args.append(canon(a, where))
            elif isinstance(a, sprite_lib.Sprite):
                if id(a) not in where:
                    raise NotImplementedError('a look-ahead test on a sprite that is not in the returned state')
                args.append(where[id(a)])
            else:
                args.append(a)
        return _symbolic.Node(node.op, *args)

    # A run that starts over has no state of its own to name its sprites by; its tests are named through the sprite
    # order of the run's generation ops instead, which every run shares with the base run.
    def op_sprites(t):
        return [sp for op in t.ops for sp in op.sprites]

    paths, exits, forced, plan = [], [], [], None
    cur_tr, cur_state = tr, state
    raw = []
    while True:
        raw.append((cur_tr, cur_state, cur_tr.sim_steps == 0))
        trail_answers = [v for _, v in cur_tr.sim_trail]
        if cur_tr.sim_steps == 0 and plan is None:
            plan = list(trail_answers)
        if len(raw) > _symbolic.MAX_PATHS:
            raise NotImplementedError('too many paths through the look-ahead loop of the state_initializer')
        k = len(trail_answers) - 1
        while k >= 0 and trail_answers[k] is False:
            k -= 1
        if k < 0:
            break
        forced = trail_answers[:k] + [False]
        cur_tr, cur_state = run(sim_forced=forced, sim_exit_plan=plan)
        if cur_tr.sim_op is None:
            raise NotImplementedError('a state_initializer whose look-ahead loop is not always reached')
    finished = [(t, st) for t, st, is_exit in raw if is_exit and st is not None]
    if not finished:
        raise NotImplementedError('a state_initializer whose look-ahead never lets it finish')
    base_tr, base_state = finished[0]
    base_where = index_sprites(base_state)
    base_ops = op_sprites(base_tr)
    statics = {id(sp): sp for sprites in base_state.values() for sp in sprites}
    restart_mask = 0
    n_exit = 0
    exit_of = {}
    for t, st, is_exit in raw:
        mine = op_sprites(t)
        if len(mine) != len(base_ops) and st is not None:
            raise NotImplementedError('a state_initializer that builds different states after different outcomes of its look-ahead')
        # sprites of this run -> (layer, index) of the base run's state: generated ones by generation order, the others
        # (built outside the initializer / static) are the same objects in every run
        where = {}
        for a, b in zip(mine, base_ops):
            if id(b) in base_where:
                where[id(a)] = base_where[id(b)]
        for sid in statics:
            where.setdefault(sid, base_where[sid])
        if st is not None:
            for name, sprites in st.items():
                for i, sp in enumerate(sprites):
                    where.setdefault(id(sp), (name, i))
        trail = [(canon(n, where), v) for n, v in t.sim_trail]
        idx = 0
        if is_exit:
            n_exit += 1
            idx = n_exit
            exit_of[id(t)] = idx
            if st is None:
                restart_mask |= 1 << idx
        paths.append((trail, idx, None))
    if n_exit > 30:
        raise NotImplementedError('a look-ahead loop with more than 30 exits')
    cell = base_tr.sim_op.index

    def uncanon(node):   # (layer, index) -> the sprite objects of the run the program is built from
        return _symbolic.Node(node.op, *[uncanon(a) if isinstance(a, _symbolic.Node) else
                                         (base_state[a[0]][a[1]] if isinstance(a, tuple) else a) for a in node.args])
    base_tr.sim_op.node = uncanon(_symbolic._merge(paths, lambda p: _symbolic.const(float(p[1]))))
    base_tr.sim_op.restart_mask = restart_mask
    # what the exits hand on: metadata values that differ from exit to exit become a selection by the exit taken
    for other_tr, other_state in finished[1:]:
        if [type(op) for op in other_tr.ops] != [type(op) for op in base_tr.ops] or \
                [(n, len(v)) for n, v in other_state.items()] != [(n, len(v)) for n, v in base_state.items()]:
            raise NotImplementedError('a state_initializer that builds different states after different outcomes of its look-ahead')
    for name, sprites in base_state.items():
        for i, sp in enumerate(sprites):
            mds = [st[name][i].factors.get('metadata') for _, st in finished]
            if not any(isinstance(m, dict) for m in mds):
                continue
            if not all(isinstance(m, dict) and set(m) == set(mds[0]) for m in mds):
                raise NotImplementedError('sprite metadata whose keys depend on the outcome of the look-ahead')
            for key in mds[0]:
                vals = [m[key] for m in mds]
                if all(v == vals[0] for v in vals):
                    continue
                if not all(isinstance(v, (bool, int, float, np.integer, np.floating, np.bool_)) for v in vals):
                    raise NotImplementedError('sprite.metadata[%r] set to non-numbers by the look-ahead' % (key,))
                node = _symbolic.const(float(vals[-1]))
                for j in range(len(vals) - 2, -1, -1):
                    test = _symbolic.Node('eq', _symbolic.Node('hdraw', cell),
                                          _symbolic.const(float(exit_of[id(finished[j][0])])))
                    node = _symbolic.Node('select', test, _symbolic.const(float(vals[j])), node)
                sp.factors['metadata'][key] = _symbolic.Sym(node)
                if not hasattr(base_tr, 'dynamic_meta'):
                    base_tr.dynamic_meta = []
                base_tr.dynamic_meta.append((sp, key, cell, {exit_of[id(finished[j][0])]: float(vals[j])
                                                             for j in range(len(vals))}))
    return base_tr, base_state


def _trace_persistent_state(state_initializer, meta_state_initializer, game_rules, first, saved):
    """An initializer that is a method of an object which keeps numbers ACROSS episodes (predators_arena.py:29-106: an
    auto-curriculum adapts `self._mass` at every reset after the first, once the meta-state exists).  The first trace
    saw the first episode.  Here the meta-state initializer is called as the environment would have (environment.py:
    86-88), one more initializer run on plain numbers shows which attributes change, and the initializer is traced
    again with those attributes symbolic: their new values are the per-reset updates, and the factors computed from
    them read a per-env slot that resets never clear.  Returns (tracer, state) of that second trace, with one PStateOp
    per such attribute in front of the ops, or None when the initializer keeps nothing."""
    owner = getattr(state_initializer, '__self__', None)
    if owner is None or saved is None:
        return None
    if meta_state_initializer is None:
        for k, v in saved.items():
            setattr(owner, k, v)
        return None
    numeric = {k: v for k, v in saved.items() if isinstance(v, float)}
    if not numeric:
        return None
    try:
        meta_state_initializer()
        with _trace.tracing():
            state_initializer()
        changed = [k for k, v in numeric.items() if vars(owner).get(k) != v]
        if not changed:
            return None
        if any(getattr(r, 'host_side', False) for r, _ in _flatten_host_rules(game_rules)):
            raise NotImplementedError('a state_initializer that keeps numbers across episodes next to rules that '
                                      'modify the meta-state on the host')
        for k in changed:
            setattr(owner, k, _symbolic.Sym(_symbolic.Node('pstate', k)))
        with _trace.tracing() as tr:
            state = state_initializer()
        updates = {}
        for k in changed:
            v = vars(owner).get(k)
            if not isinstance(v, _symbolic.Sym):
                raise NotImplementedError('persistent initializer attribute %r is replaced by a non-numeric value' % k)
            updates[k] = v.node
        first_state = getattr(first, 'state_shape', None)
        if first_state is not None and first_state != [(n, len(v)) for n, v in state.items()]:
            raise NotImplementedError('a state_initializer whose later episodes build a different state than the first')
        # The sprites read the attribute AFTER its update (predators_arena.py:88-96), i.e. the slot's content once the
        # PStateOp has run: the update's expression inside a factor is that slot; the old value is not available there.
        keys = {updates[k].key(): k for k in changed}

        def after(node):
            if node.key() in keys:
                return _symbolic.Node('pstate', keys[node.key()])
            if node.op == 'pstate':
                raise NotImplementedError('a sprite factor computed from the value a persistent attribute had BEFORE '
                                          'this reset\'s update')
            return _symbolic.Node(node.op, *[after(a) if isinstance(a, _symbolic.Node) else a for a in node.args])
        for sprites in state.values():
            for sp in sprites:
                for fname, val in sp.factors.items():
                    if isinstance(val, sprite_lib.ExprFactor):
                        val.node = after(val.node)
        tr.ops[:0] = [_trace.PStateOp(k, numeric[k], updates[k]) for k in changed]
        return tr, state
    finally:
        for k in list(vars(owner)):
            if k not in saved:
                delattr(owner, k)
        for k, v in saved.items():
            setattr(owner, k, v)


def _flatten_host_rules(rules, out=None):
    out = [] if out is None else out
    for r in (rules if isinstance(rules, (list, tuple)) else (rules,)):
        out.append((r, None))
        kids = getattr(r, '_rules', None)
        if kids:
            _flatten_host_rules(list(kids), out)
    return out


def compile_config(state_initializer, physics, task, action_space, observers, game_rules=(),
                   meta_state_initializer=None, layer_capacity=None, keep_sprite_factors=False):
    # meta_state lives on the host (environment.py keeps it for `ModifyMetaState`)
    P = _abi.Program()
    P.abi_version = _abi.MOOG_ABI_VERSION
    if keep_sprite_factors:   # scale / aspect_ratio per sprite in the records (LoggingEnvironment)
        P.sprite_factors = 1
    shapes = _ShapeTable(P)

    # ---- trace the state initializer (environment.py:86) -----------------------
    owner = getattr(state_initializer, '__self__', None)
    owner_vars = dict(vars(owner)) if hasattr(owner, '__dict__') else None   # (tracing must leave the config's objects as they were)
    tr, state = _trace_initializer(state_initializer)
    tr.state_shape = [(n, len(v)) for n, v in state.items()] if isinstance(state, dict) else None
    persistent = _trace_persistent_state(state_initializer, meta_state_initializer, game_rules, tr, owner_vars)
    if persistent is not None:
        tr, state = persistent
    if not isinstance(state, dict):
        raise TypeError('state_initializer must return an OrderedDict of sprite lists')
    layer_names = list(state.keys())
    if len(layer_names) > _abi.MOOG_MAX_LAYERS:
        raise ValueError('too many layers (max %d)' % _abi.MOOG_MAX_LAYERS)

    def layer_index(name):
        if name not in layer_names:
            raise KeyError('layer %r is not a key of the environment state' % (name,))
        return layer_names.index(name)

    # Layers that rules append to (CreateSprites, ChangeLayer's new_layer) behave like the
    # reference's Python lists: live sprites stay packed at the front of the layer's slots
    # in list order, and the layer gets spare slots (`layer_capacity`, default +8).
    _REF_COUNTER[0] = 0
    _REF_INDEX.clear()
    flat_rules = _flatten_rules(tuple(game_rules))
    rule_ref_index = [_REF_INDEX[id(r)] for r, _ in flat_rules]
    dynamic = []
    for r, _ in flat_rules:
        if isinstance(r, rules_lib.CreateSprites):
            dynamic.append(r._layer)
        elif isinstance(r, rules_lib.ChangeLayer):
            dynamic.append(r._new_layer)
    layer_capacity = dict(layer_capacity or {})
    for name in layer_capacity:
        layer_index(name)

    # sprite_generators.shuffle swaps its sprites through a spare slot behind them
    shuffle_ops = [op for op in tr.ops if isinstance(op, _trace.ShuffleOp)]
    for op in shuffle_ops:
        owners = [n for n in layer_names if state[n] and state[n][-1] is op.members[-1]]
        if len(owners) != 1 or [id(x) for x in state[owners[0]][-len(op.members):]] != [id(x) for x in op.members]:
            raise NotImplementedError('shuffled sprites must be the last sprites of one layer, in generation order')
        layer_capacity[owners[0]] = max(int(layer_capacity.get(owners[0], 0)), len(state[owners[0]]) + 1)

    # slots: layer order, list order
    slot_of = {}
    slot_sprite = []
    layer_n_init = {}
    P.n_layers = len(layer_names)
    for li, name in enumerate(layer_names):
        P.layer_slot0[li] = len(slot_sprite)
        for s in state[name]:
            if id(s) in slot_of:
                raise ValueError('the same sprite object appears twice in the state')
            slot_of[id(s)] = len(slot_sprite)
            P.slot_layer[len(slot_sprite)] = li
            slot_sprite.append(s)
        n_init = len(slot_sprite) - P.layer_slot0[li]
        layer_n_init[name] = n_init   # (slots of the layer that have a recipe: the floor of any later re-sizing)
        cap = n_init
        if name in dynamic:
            P.layer_dynamic[li] = 1
            cap = int(layer_capacity.get(name, n_init + 8))
        elif name in layer_capacity:
            cap = int(layer_capacity[name])
        if cap < n_init:
            raise ValueError('layer_capacity[%r] is below its initial sprite count' % (name,))
        for _ in range(cap - n_init):   # spare slots: no recipe, dead after reset
            P.slot_layer[len(slot_sprite)] = li
            slot_sprite.append(None)
        P.layer_nslots[li] = len(slot_sprite) - P.layer_slot0[li]
    for alias_id, target in tr.alias.items():   # later alternatives of a sample_generator share the first one's slots
        if id(target) in slot_of:
            slot_of[alias_id] = slot_of[id(target)]
    S = len(slot_sprite)
    if S > _abi.MOOG_MAX_SLOTS:
        raise ValueError('too many sprites (max %d)' % _abi.MOOG_MAX_SLOTS)
    P.n_slots = S

    # ---- generation ops: traced ops in draw order, statics as 1-sprite ops ------
    ops = []            # (GenOp-like, [sprites])
    op_index_of_sprite = {}
    traced_ids = set()
    for op in tr.ops:
        for s in op.sprites:
            traced_ids.add(id(s))
    pending_static = [s for s in slot_sprite if s is not None and id(s) not in traced_ids]
    # statics consume no randomness: put them first so that they can be avoided
    for s in pending_static:
        if s.is_symbolic:
            raise ValueError('symbolic sprite created outside a traced generator')
        ops.append((None, [s]))
    for op in tr.ops:
        ops.append((op, op.sprites))
    n_reset_ops = len(ops)
    runtime_op_of_rule = {}
    for ri, (r, _) in enumerate(flat_rules):   # CreateSprites generators run at rule time
        if isinstance(r, rules_lib.CreateSprites):
            with _trace.tracing() as tr2:
                r._generator(without_overlapping=[])
            if len(tr2.ops) != 1:
                raise NotImplementedError('CreateSprites generator must be one generate_sprites()')
            runtime_op_of_rule[ri] = len(ops)
            ops.append((tr2.ops[0], tr2.ops[0].sprites))
    if len(ops) > _abi.MOOG_MAX_OPS:
        raise ValueError('too many sprite generation ops (max %d)' % _abi.MOOG_MAX_OPS)
    for oi, (_, sprites) in enumerate(ops):
        for s in sprites:
            op_index_of_sprite[id(s)] = oi
    P.n_ops = len(ops)

    cand_n = 0
    vcap = [0] * S
    op_max_nv = {}

    def put_code(code):
        """Appends postfix expression code (plus X_END) to program.dcode; returns its offset."""
        if _symbolic.depth(code) > _abi.MOOG_X_STACK:
            raise NotImplementedError('expression too deep for the device evaluator')
        if _symbolic.uses_attr(code, ('scale', 'aspect_ratio', 'mass', 'c0', 'c1', 'c2')):
            P.sprite_factors = 1
        code = list(code) + [dict(op=_abi.MOOG_X_END)]
        base = P.n_dcode
        if base + len(code) > _abi.MOOG_MAX_DCODE:
            raise ValueError('expression / distribution code too long')
        for i, ins in enumerate(code):
            I = P.dcode[base + i]
            I.op, I.a, I.b, I.x = ins['op'], ins.get('a', 0), ins.get('b', 0), ins.get('x', 0.0)
        P.n_dcode = base + len(code)
        return base

    # reset-time expressions: factors the initializer computed from its own np.random draws (_trace.Tracer.hdraw)
    P.n_hdraws = tr.n_hdraws
    shape_code = {}   # id(ExprShape rows source) -> code offset (one shape object shared by several sprites)
    # the per-reset random maze (maze_lib/_traced.py): its matrix lives in the env records
    from .maze_lib import _traced as traced_maze
    wall_shape0 = None
    coord_tables = {}   # tuple(table) -> offset in program.cand
    if tr.maze is not None:
        if tr.maze['flip'] is None:
            raise NotImplementedError('a random maze matrix that no Maze(...) wraps')
        P.maze.size, P.maze.random = tr.maze['ambient'], 1
        P.maze.gen_size, P.maze.flip = tr.maze['gen_size'], int(tr.maze['flip'])
        if 'walls' in tr.maze:   # the wall squares of all cells, column-major like Maze.to_sprites (maze.py:98-111)
            n = P.maze.size
            v = np.linspace(0., 1., n + 1)
            for x in range(n):
                for y in range(n):
                    sid = shapes.intern(np.array([[v[x], v[y]], [v[x], v[y + 1]], [v[x + 1], v[y + 1]],
                                                  [v[x + 1], v[y]]]))
                    if wall_shape0 is None:
                        wall_shape0 = sid
                    if sid != wall_shape0 + x * n + y:
                        raise NotImplementedError('maze wall squares that coincide with other shapes of the config')
    # numbers the initializer keeps across episodes (_trace_persistent_state): state-slot rules behind the config's rules
    pstate_names = [op.name for op in tr.ops if isinstance(op, _trace.PStateOp)]
    pstate_slot = {name: len(flat_rules) + k for k, name in enumerate(pstate_names)}
    # sprites built outside the initializer keep what rules did to them across resets (program.born_rule)
    persist_slots = [sl for sl, sp in enumerate(slot_sprite) if sp is not None and getattr(sp, 'built_outside', False)
                     and id(sp) not in traced_ids]
    born_rule = len(flat_rules) + len(pstate_names) if persist_slots else -1
    for sl in persist_slots:
        P.slot_persist[sl] = 1
    P.born_rule = born_rule + 1

    def live_resolver(key, ref):   # sprites named by the initializer's look-ahead / put-back code -> their slots
        if key == 'simstep':
            return [op.index for op in tr.ops if isinstance(op, _trace.SimOp)][0] + 1
        if key != 'slot' or id(ref) not in slot_of:
            raise NotImplementedError('the state_initializer looks at a sprite that is not in the returned state')
        return slot_of[id(ref)]

    def pstate_resolver(key, name):
        if key != 'pstate':
            raise NotImplementedError('persistent initializer state computed from %r' % (key,))
        return pstate_slot[name]
    for oi, (op, sprites) in enumerate(ops):
        G = P.ops[oi]
        runtime = oi >= n_reset_ops
        G.runtime = int(runtime)
        if getattr(op, 'cond', None) is not None:
            G.cond_hdraw, G.cond_value = 1 + op.cond[0], op.cond[1]
        if isinstance(op, _trace.ChoiceOp):
            G.cell_sel, G.cell_arg, G.count_max = _abi.MOOG_CELL_CHOICE, op.index, op.n
            G.code_off = -1
            G.factors[0].cand_off = -1
            if op.p is not None:
                if len(op.p) != op.n or cand_n + op.n > _abi.MOOG_MAX_CAND:
                    raise ValueError('sample_generator probabilities')
                cdf = np.cumsum(np.asarray(op.p, dtype=np.float64))
                cdf /= cdf[-1]
                G.factors[0].cand_off = cand_n
                for v in cdf:
                    P.cand[cand_n] = float(v)
                    cand_n += 1
            op_max_nv[oi] = 0
            continue
        if isinstance(op, _trace.ShuffleOp):
            G.cell_sel, G.cell_arg = _abi.MOOG_CELL_SHUFFLE, len(op.members)
            G.slot0 = slot_of[id(op.members[0])]
            G.code_off = -1
            op_max_nv[oi] = 0
            continue
        if not sprites:   # MOOG_CELL_GENERATE / MOOG_CELL_SAMPLE / HDRAW / HEXPR: randomness or a computed value, no sprite
            G.cell_sel, G.cell_arg = op.cell
            G.code_off = -1
            if isinstance(op, _trace.SimOp):
                if getattr(tr, 'sim_physics', physics) is not physics:
                    raise NotImplementedError('the state_initializer steps a physics object other than the environment\'s')
                G.code_off = put_code(_symbolic.emit(op.node, [], live_resolver))
                G.count_max = 100000
                G.max_tries = int(getattr(op, 'restart_mask', 0))   # exits after which the initializer starts over
            elif isinstance(op, _trace.StoreOp):
                G.cell_arg = live_resolver('slot', op.sprite)
                code = []
                for attr, n in op.stores.items():
                    _symbolic.emit(n, code, live_resolver)
                    code.append(dict(op=_abi.MOOG_X_STORE, a=_symbolic.ATTRS.index(attr)))
                G.code_off = put_code(code)
            elif isinstance(op, _trace.PStateOp):
                G.cell_arg = pstate_slot[op.name]
                G.code_off = put_code(_symbolic.emit(op.node, [], pstate_resolver))
                G.factors[0].a = op.init
                P.rule_state2 = 1
            elif isinstance(op, _trace.HExprOp):
                G.code_off = put_code(_symbolic.emit(op.node, [], live_resolver))
                G.count_min = int(bool(op.tagged))   # the value's numpy dtype tag goes to the next cell
            elif getattr(op, 'accept', None):   # the accept test(s) of a rejection loop over this draw
                node = op.accept[0]
                for extra in op.accept[1:]:
                    node = _symbolic.Node('and', node, extra)
                G.code_off = put_code(_symbolic.emit(node, [], None))
            op_max_nv[oi] = 0
            continue
        cell = traced_maze.cell_of(sprites[0])
        if cell is not None:
            if len(sprites) != 1 or runtime:
                raise NotImplementedError('generators over maze cells')
            G.cell_sel, G.cell_arg = cell
        if runtime:
            slots = []
            if op.avoid:
                raise ValueError('CreateSprites passes without_overlapping itself')
        else:
            for s in sprites:
                if id(s) not in slot_of:
                    raise ValueError('a generated sprite is missing from the returned state')
            slots = [slot_of[id(s)] for s in sprites]
            if slots != list(range(slots[0], slots[0] + len(slots))):
                # the call's sprites end up in other places of the state than in call order (red_green.py:157-183: the
                # first two obstacles become the layers 'red' and 'green' behind the rest): still one contiguous run of
                # slots, filled through a table
                lo = min(slots)
                if sorted(slots) != list(range(lo, lo + len(slots))) or cell is not None:
                    raise ValueError('sprites of one generator call must occupy one contiguous run of slots')
                if cand_n + len(slots) > _abi.MOOG_MAX_CAND:
                    raise ValueError('too many Discrete candidates')
                G.cell_arg = 1 + cand_n
                for sl in slots:
                    P.cand[cand_n] = float(sl - lo)
                    cand_n += 1
                slots = list(range(lo, lo + len(slots)))
            G.slot0 = slots[0]
        G.count_max = len(sprites)
        G.count_min = len(sprites) if op is None else op.count_min
        G.disjoint = 0 if op is None else int(op.disjoint)
        G.max_tries = 0 if op is None else int(op.max_tries)
        G.fail_gracefully = int(bool(getattr(op, 'fail_gracefully', False)))
        if getattr(op, 'restart_if_short', False):
            G.fail_gracefully = 2   # ... and the initializer starts over when the call came out short (red_green.py:152-155)
        avoid = 0
        if op is not None and not runtime:
            for a in op.avoid:
                if id(a) not in op_index_of_sprite:
                    raise ValueError('without_overlapping refers to a sprite that is not in the state')
                aj = op_index_of_sprite[id(a)]
                if aj >= oi:
                    raise ValueError('without_overlapping refers to a later generator')
                if aj >= 64:
                    raise NotImplementedError('without_overlapping over more than the first 64 generation ops')
                avoid |= (1 << aj)
        G.avoid_ops = avoid
        proto = sprites[0]
        for other in sprites[1:]:   # one recipe per call: a factor assigned after the call must be the same for all its sprites
            for fname in _abi.FACTOR_NAMES:
                a, b = proto.factors[fname], other.factors[fname]
                if isinstance(a, sprite_lib.ExprFactor) != isinstance(b, sprite_lib.ExprFactor) or \
                        (isinstance(a, sprite_lib.ExprFactor) and a.node.key() != b.node.key()) or \
                        (isinstance(a, (int, float, np.integer, np.floating)) and
                         isinstance(b, (int, float, np.integer, np.floating)) and float(a) != float(b)):
                    raise NotImplementedError('sprites of one generator call were given different values of %r after the call'
                                              % (fname,))
        order = [k for k in proto.sample_order   # factors read off a maze cell take no draw
                 if not isinstance(proto.factors[k], (traced_maze.CellShape, traced_maze.CellIndex))]
        G.n_sampled = len(order)
        draw_seq = getattr(op, 'draw_seq', None)   # factor samples interleaved with direct draws (_trace.note_sprite)
        max_nv = 0
        G.code_off = -1
        tree_keys = set()
        if op is not None and op.dist is not None and not distribs.is_flat(op.dist):
            # Mixture / Intersection / SetMinus / Selection / Discrete(probs): the whole
            # factor distribution of this generator becomes a distribution program
            nv_box = [0]

            def factor_value(key, v):
                if key == 'shape':
                    sid = shapes.intern(v)
                    nv_box[0] = max(nv_box[0], shapes.nverts(sid))
                    return float(sid)
                return float(v)
            G.code_off, tree_keys, cand_n = _distcode.lower(P, op.dist, cand_n, factor_value)
            max_nv = nv_box[0]
            order = []
            G.n_sampled = 0
        for fi, fname in enumerate(_abi.FACTOR_NAMES):
            F = G.factors[fi]
            val = proto.factors[fname]
            if fname in tree_keys:
                F.kind = _abi.MOOG_DIST_TREE
                continue
            if isinstance(val, sprite_lib.ExprFactor):
                if fname in ('shape', 'opacity'):
                    raise NotImplementedError('a computed value as factor %r' % (fname,))

                def slot_resolver(key, ref, _oi=oi):
                    if key == 'pstate':
                        return pstate_slot[ref]
                    if key != 'slot' or id(ref) not in slot_of:
                        raise NotImplementedError('a factor copied from a sprite that is not in the state')
                    if op_index_of_sprite[id(ref)] >= _oi:
                        raise NotImplementedError('a factor copied from a sprite that is created later')
                    return slot_of[id(ref)]
                F.kind, F.cand_off = _abi.MOOG_DIST_EXPR, put_code(_symbolic.emit(val.node, [], slot_resolver))
            elif isinstance(val, sprite_lib.ExprShape):
                if fname != 'shape':
                    raise NotImplementedError('a computed polygon as factor %r' % (fname,))
                nv = len(val.rows)
                if nv < 3 or nv > 64 or any(len(r) != 2 for r in val.rows):
                    raise NotImplementedError('computed shapes need 3 .. 64 vertices of two coordinates')
                key = tuple(n.key() for r in val.rows for n in r)
                if key not in shape_code:
                    code = []
                    for vi, row in enumerate(val.rows):
                        for ci, node in enumerate(row):
                            _symbolic.emit(node, code, None)
                            code.append(dict(op=_abi.MOOG_X_STORE_VERT, a=2 * vi + ci))
                    shape_code[key] = put_code(code)
                F.kind, F.n_cand, F.cand_off = _abi.MOOG_DIST_EXPR_SHAPE, nv, shape_code[key]
                max_nv = max(max_nv, nv)
            elif isinstance(val, traced_maze.CellShape):
                if fname != 'shape':
                    raise NotImplementedError('a maze wall square as factor %r' % (fname,))
                F.kind, F.a = _abi.MOOG_DIST_MAZE_SHAPE, float(wall_shape0)
                max_nv = max(max_nv, 4)
            elif isinstance(val, traced_maze.CellIndex):
                if fname in ('shape', 'opacity'):
                    raise NotImplementedError('a maze cell index as factor %r' % (fname,))
                table = tuple(val.table(P.maze.size))
                if table not in coord_tables:
                    if cand_n + len(table) > _abi.MOOG_MAX_CAND:
                        raise ValueError('too many Discrete candidates')
                    coord_tables[table] = cand_n
                    for c in table:
                        P.cand[cand_n] = c
                        cand_n += 1
                F.kind, F.n_cand, F.cand_off = _abi.MOOG_DIST_MAZE_COORD, int(val.axis), coord_tables[table]
            elif isinstance(val, sprite_lib.SymbolicFactor):
                d = val.dist
                if isinstance(d, distribs.Continuous):
                    F.kind = _abi.MOOG_DIST_CONTINUOUS
                    F.a, F.b = float(d.minval), float(d.maxval)
                    if str(d.dtype) not in ('float32', 'float64'):
                        raise NotImplementedError('Continuous dtype %r' % (d.dtype,))
                    F.f32 = int(str(d.dtype) == 'float32')
                elif isinstance(d, distribs.Discrete):
                    F.kind = _abi.MOOG_DIST_DISCRETE
                    F.n_cand = len(d.candidates)
                    F.cand_off = cand_n
                    if cand_n + F.n_cand > _abi.MOOG_MAX_CAND:
                        raise ValueError('too many Discrete candidates')
                    for c in d.candidates:
                        if fname == 'shape':
                            sid = shapes.intern(c)
                            max_nv = max(max_nv, shapes.nverts(sid))
                            P.cand[cand_n] = float(sid)
                        else:
                            P.cand[cand_n] = float(c)
                        cand_n += 1
                else:
                    raise NotImplementedError(
                        'distribution %r is only lowered inside a generate_sprites() generator'
                        % (type(d).__name__,))
            else:
                F.kind = _abi.MOOG_DIST_CONST
                if fname == 'shape':
                    sid = shapes.intern(val)
                    max_nv = max(max_nv, shapes.nverts(sid))
                    F.a = float(sid)
                else:
                    F.a = float(val)
        if draw_seq is not None and order:
            entries = [(_abi.FACTOR_NAMES.index(v) if kind == 'factor' else _abi.MOOG_NUM_FACTORS + v)
                       for kind, v in draw_seq if kind == 'hdraw' or v in order]
        else:
            entries = [_abi.FACTOR_NAMES.index(fname) for fname in order]
        if len(entries) > _abi.MOOG_MAX_OP_DRAWS:
            raise NotImplementedError('more than %d draws in one sprite' % _abi.MOOG_MAX_OP_DRAWS)
        G.n_sampled = len(entries)
        for k, ent in enumerate(entries):
            G.sample_order[k] = ent
        # where each factor's uniform sits among the draws of one sample (the device evaluates the factors in lanes)
        for fi in range(len(_abi.FACTOR_NAMES)):
            G.factors[fi].draw_pos = -1
        pos = 0
        for ent in entries:
            if ent >= _abi.MOOG_NUM_FACTORS:
                pos += 1
            elif G.factors[ent].kind in (_abi.MOOG_DIST_CONTINUOUS, _abi.MOOG_DIST_DISCRETE):
                G.factors[ent].draw_pos = pos
                pos += 1
        G.n_draws = pos
        # a constant one-sprite op (no draw, no rejection test, no computed factor): the engine may build a run of such
        # ops side by side, one per lane (bit 1 of `disjoint`; the oracle builds them one after the other)
        if (not runtime and len(sprites) == 1 and G.count_min == 1 and pos == 0 and not G.avoid_ops and not G.disjoint
                and G.code_off < 0 and not G.cond_hdraw and not G.fail_gracefully and
                G.cell_sel in (_abi.MOOG_CELL_NONE, _abi.MOOG_CELL_SAMPLED, _abi.MOOG_CELL_OPEN_RANK, _abi.MOOG_CELL_WALL_RANK)
                and not (G.cell_sel == _abi.MOOG_CELL_NONE and G.cell_arg) and
                all(G.factors[fi].kind in (_abi.MOOG_DIST_CONST, _abi.MOOG_DIST_MAZE_COORD, _abi.MOOG_DIST_MAZE_SHAPE)
                    for fi in range(len(_abi.FACTOR_NAMES)))):
            G.disjoint = 2
        for sl in slots:
            vcap[sl] = max(vcap[sl], max_nv)   # (sample_generator alternatives share slots)
        op_max_nv[oi] = max_nv
    P.n_cand = cand_n
    # every slot of a dynamic layer can hold any sprite that may end up in that layer
    layer_nv = [max([vcap[sl] for sl in range(P.layer_slot0[li], P.layer_slot0[li] + P.layer_nslots[li])]
                    or [0]) for li in range(P.n_layers)]
    for _ in range(P.n_layers):   # ChangeLayer chains: propagate to a fixed point
        for ri, (r, _p) in enumerate(flat_rules):
            if isinstance(r, rules_lib.CreateSprites):
                li = layer_index(r._layer)
                layer_nv[li] = max(layer_nv[li], op_max_nv[runtime_op_of_rule[ri]])
            elif isinstance(r, rules_lib.ChangeLayer):
                lo, ln = layer_index(r._old_layer), layer_index(r._new_layer)
                layer_nv[ln] = max(layer_nv[ln], layer_nv[lo])
    for li in range(P.n_layers):
        if P.layer_dynamic[li]:
            for sl in range(P.layer_slot0[li], P.layer_slot0[li] + P.layer_nslots[li]):
                vcap[sl] = layer_nv[li]
    for op in shuffle_ops:   # any of the shuffled sprites can end up in any of their slots (and in the spare)
        s0 = slot_of[id(op.members[0])]
        rng_ = range(s0, s0 + len(op.members) + 1)
        top = max(vcap[sl] for sl in rng_)
        for sl in rng_:
            vcap[sl] = top
    voff = 0
    for sl in range(S):
        P.slot_voff[sl] = voff
        P.slot_vcap[sl] = vcap[sl]
        voff += vcap[sl]
    P.n_total_verts = voff

    # ---- physics (physics.py:15, :88-117) -------------------------------------------
    if not isinstance(physics, physics_lib.Physics):
        raise NotImplementedError('physics must be a moog.physics.Physics instance')
    P.updates_per_env_step = int(physics.updates_per_env_step)
    if len(physics._forces) > _abi.MOOG_MAX_FORCES:
        raise ValueError('too many forces')
    maze_layers = set()   # wall layers the maze walks / MazePhysics infer their maze from
    det_walks = []        # DeterministicMazeWalk forces: (force record, flattened velocity table)
    for fi, entry in enumerate(physics._forces):
        force, args = entry[0], entry[1:]
        F = P.forces[fi]
        F.n_a = _fill_layers(F.layers_a, args[0], layer_index)
        F.n_b = _fill_layers(F.layers_b, args[1], layer_index) if len(args) > 1 else 0
        if len(args) > 2:
            raise NotImplementedError('forces over more than two sprites')
        pair = False
        if isinstance(force, physics_lib.Drag):
            F.kind, F.p0 = _abi.MOOG_FORCE_DRAG, force._coeff_friction
        elif isinstance(force, physics_lib.KineticFriction):
            F.kind, F.p0 = _abi.MOOG_FORCE_KINETIC_FRICTION, force._coeff_friction
        elif isinstance(force, physics_lib.DownGravity):
            F.kind, F.p0 = _abi.MOOG_FORCE_DOWN_GRAVITY, force._g
        elif isinstance(force, physics_lib.RandomForce):
            F.kind, F.p0 = _abi.MOOG_FORCE_RANDOM, force._max_force_magnitude
        elif isinstance(force, physics_lib.Gravity):
            F.kind, F.p0, F.symmetric, pair = _abi.MOOG_FORCE_GRAVITY, force._g, int(force._symmetric), True
        elif isinstance(force, physics_lib.DistanceForce):
            fn = force._force_fn
            pair = True
            F.symmetric = int(force._symmetric)
            if force._force_node is not None:   # any scalar function of the distance, traced
                F.kind = _abi.MOOG_FORCE_DISTANCE_EXPR
                F.i0 = put_code(_symbolic.emit(force._force_node, []))
            elif fn.kind == 'linear':
                F.kind = _abi.MOOG_FORCE_DISTANCE_LINEAR
                F.p0, F.p1 = fn.params['zero_intercept'], fn.params['slope']
                F.i0, F.i1 = int(fn.params['apply_distant_force']), int(fn.params['apply_nearby_force'])
            else:
                F.kind = _abi.MOOG_FORCE_DISTANCE_SPRING
                F.p0, F.p1 = fn.params['spring_constant'], fn.params['equilibrium']
        elif isinstance(force, physics_lib.RandomMazeWalk):
            F.kind, F.p0 = _abi.MOOG_FORCE_MAZE_WALK, force._speed
            F.i0 = (int(bool(force._prevent_backtracking)) | (int(bool(force._allow_wall_backtracking)) << 1) |
                    (int(bool(force._only_turn_at_wall)) << 2))
            maze_layers.add(force._maze_layer)
            _reject_f32_velocity(state, args[0], 'RandomMazeWalk')
        elif isinstance(force, physics_lib.DeterministicMazeWalk):
            F.kind, F.p0 = _abi.MOOG_FORCE_MAZE_WALK_DET, force._speed
            table = []
            for v in force._step_velocities:
                v = np.asarray(v, dtype=np.float64).reshape(-1)
                if v.shape != (2,):
                    raise ValueError('DeterministicMazeWalk: step_velocities must be 2-vectors')
                table += [float(v[0]), float(v[1])]
            det_walks.append((F, table))
            maze_layers.add(force._maze_layer)
            _reject_f32_velocity(state, args[0], 'DeterministicMazeWalk')
        elif isinstance(force, physics_lib.Collision):
            pair = True
            F.kind = _abi.MOOG_FORCE_COLLISION
            F.p0 = force._elasticity
            F.symmetric = int(force._symmetric)
            F.i0 = int(force._update_angle_vel)
            F.i1 = int(force._max_recursion_depth)
        else:
            raise NotImplementedError('force %r is not lowered' % (type(force).__name__,))
        if pair != (F.n_b > 0):
            raise ValueError('%s applied to the wrong number of layer arguments' % type(force).__name__)
    P.n_forces = len(physics._forces)
    if len(physics._corrective_physics) > _abi.MOOG_MAX_CORRECTIVE:
        raise ValueError('too many corrective physics entries')
    for ci, c in enumerate(physics._corrective_physics):
        C = P.corrective[ci]
        if isinstance(c, physics_lib.ConstantSpeed):
            C.kind = _abi.MOOG_CORR_CONSTANT_SPEED
            C.n_layers = _fill_layers(C.layers, c._layer_names, layer_index)
            C.speed = c._speed
        elif isinstance(c, physics_lib.Tether):
            C.kind = (_abi.MOOG_CORR_TETHER_ZIPPED if isinstance(c, physics_lib.TetherZippedLayers)
                      else _abi.MOOG_CORR_TETHER)
            C.n_layers = _fill_layers(C.layers, c._layer_names, layer_index)
            C.update_angle_vel = int(bool(c._update_angle_vel))
            if c._anchor is not None:
                C.has_anchor = 1
                C.anchor[0], C.anchor[1] = float(c._anchor[0]), float(c._anchor[1])
            if not c._update_angle_vel:
                P.vel_alias = 1   # the tethered sprites share one velocity ndarray afterwards
            # the device restates numpy's promotion for Python-float masses only
            for name in c._layer_names:
                for sp in state[name]:
                    m = sp.factors['mass']
                    if isinstance(m, sprite_lib.SymbolicFactor) or not isinstance(m, (int, float)):
                        raise NotImplementedError(
                            'tethered sprites must have constant Python-number masses')
        elif isinstance(c, physics_lib.MazePhysics):
            C.kind = _abi.MOOG_CORR_MAZE
            C.n_layers = _fill_layers(C.layers, c._avatar_layers, layer_index)
            C.speed = float('nan') if c._constant_speed is None else float(c._constant_speed)
            C.anchor[0] = float('nan') if c._max_speed is None else float(c._max_speed)
            maze_layers.add(c._maze_layer)
            if physics.updates_per_env_step != 1:
                raise ValueError('Must have updates_per_env_step be 1 for maze.')   # maze_physics.py:207-208
            _reject_f32_velocity(state, c._avatar_layers, 'MazePhysics')
        else:
            raise NotImplementedError('corrective physics %r is not lowered' % (type(c).__name__,))
    P.n_corrective = len(physics._corrective_physics)
    if maze_layers:
        # Maze.from_state (maze.py:39-84) runs when the physics is reset; the wall layer is constant, so the
        # matrix is inferred once, here
        if len(maze_layers) != 1:
            raise NotImplementedError('one maze layer per environment')
        from . import maze_lib
        wall_layer = next(iter(maze_layers))
        if tr.maze is not None:
            # the walls are the traced maze's own squares, so Maze.from_state would give its matrix back: the
            # smallest N with every wall vertex on the 1 / N lattice is the ambient size (a wall square has a
            # corner at an odd multiple of 1 / N), and a cell centre lies in a wall iff the cell is a wall
            if [id(sp) for sp in state[wall_layer]] != [id(sp) for sp in tr.maze.get('walls', ())]:
                raise NotImplementedError('a maze layer that is not Maze(random matrix).to_sprites()')
        else:
            mz = maze_lib.Maze.from_state(state, maze_layer=wall_layer)
            if mz.maze_size > _abi.MOOG_MAX_MAZE:
                raise NotImplementedError('mazes beyond %d x %d cells' % (_abi.MOOG_MAX_MAZE, _abi.MOOG_MAX_MAZE))
            P.maze.size = int(mz.maze_size)
            for j in range(mz.maze_size):
                P.maze.rows[j] = int(sum(int(bool(mz.maze[j, i])) << i for i in range(mz.maze_size)))

    # ---- game rules ---------------------------------------------------------------
    if len(flat_rules) > _abi.MOOG_MAX_RULES:
        raise ValueError('too many game rules')

    # meta_state[key] == 'phase name' tests refer to the PhaseSequence that publishes under key
    phase_keys = {}
    for ri, (r, _parent) in enumerate(flat_rules):
        if isinstance(r, rules_lib.PhaseSequence) and r._meta_state_key is not None:
            phase_keys[r._meta_state_key] = (ri, [ph.name for ph in r._phases])

    fixation_keys = {}   # meta-state key -> index of the Fixation rule that counts under it
    for ri, (r, _parent) in enumerate(flat_rules):
        if isinstance(r, rules_lib.Fixation):
            if r._meta_state_fixation_key in fixation_keys or r._meta_state_fixation_key in phase_keys:
                raise NotImplementedError('two rules publish meta_state[%r]' % (r._meta_state_fixation_key,))
            fixation_keys[r._meta_state_fixation_key] = ri

    meta_tables = {}
    meta_sides = {}   # while a pair function is lowered: sprite index -> the layers that side ranges over

    def fixed_sprite(where):
        lname, k = where
        if lname in dynamic or k >= len(state[lname]):
            raise NotImplementedError('state[%r][%d] in a task function: the layer must hold that sprite for the whole '
                                      'episode' % (lname, k))
        return state[lname][k]

    def resolve_phase(key, name):
        if key == 'slot':   # a sprite named by position in a task function, (layer, k)
            return slot_of[id(fixed_sprite(name))] if isinstance(name, tuple) else slot_of[id(name)]
        if key == 'lmeta':
            lname, k, mkey = name
            md = fixed_sprite((lname, k)).factors.get('metadata')
            if not isinstance(md, dict) or mkey not in md:
                raise NotImplementedError('state[%r][%d].metadata[%r] is read by a task function but never set' % (lname, k, mkey))
            v = md[mkey]
            if isinstance(v, _symbolic.Sym):
                return v.node
            if not isinstance(v, (bool, int, float, np.integer, np.floating, np.bool_)):
                raise NotImplementedError('sprite.metadata[%r] = %r: only numbers and bools are lowered' % (mkey, v))
            return _symbolic.const(float(v))
        if key == 'meta':   # sprite.metadata[name]: one value per slot (spare slots of dynamic layers: the layer's last recipe)
            idx, name = name
            side = meta_sides.get(idx)
            if side is not None and len(side) == 1 and side[0] not in dynamic and len(state[side[0]]) == 1:
                return 'node', resolve_phase('lmeta', (side[0], 0, name))   # one fixed sprite: its own value
            if name not in meta_tables:
                off = int(P.n_cand)
                if off + S > _abi.MOOG_MAX_CAND:
                    raise NotImplementedError('sprite.metadata[%r]: no room for a per-slot table' % (name,))
                last = float('nan')
                for sl, sp in enumerate(slot_sprite):
                    if sp is not None:   # (a spare slot of a dynamic layer keeps the value of the recipe before it)
                        md = sp.factors.get('metadata')
                        last = float('nan')
                        if isinstance(md, dict) and name in md:
                            v = md[name]
                            if not isinstance(v, (bool, int, float, np.integer, np.floating, np.bool_)):
                                raise NotImplementedError('sprite.metadata[%r] = %r: only numbers and bools are lowered'
                                                          % (name, v))
                            last = float(v)
                    P.cand[off + sl] = last
                P.n_cand = off + S
                meta_tables[name] = off
            return 'table', meta_tables[name]
        if key is None:   # an overlap test against state[name][0]
            return layer_index(name)
        if name is None:  # the number a Fixation rule keeps under this key
            if key not in fixation_keys:
                raise NotImplementedError('meta_state[%r] is not kept by a Fixation rule' % (key,))
            return fixation_keys[key]
        if key not in phase_keys:
            raise NotImplementedError('meta_state[%r] is not published by a PhaseSequence' % (key,))
        ri, names = phase_keys[key]
        if name not in names:
            raise ValueError('no phase named %r' % (name,))
        return ri, names.index(name)

    LANE_OPS = set(getattr(_abi, 'MOOG_X_' + n) for n in (
        'CONST', 'ATTR', 'ADD', 'SUB', 'MUL', 'DIV', 'REM', 'MIN', 'MAX', 'LT', 'LE', 'GT', 'GE', 'EQ', 'NE', 'AND', 'OR', 'NEG',
        'ABS', 'SQRT', 'SIN', 'COS', 'FLOOR', 'NOT', 'SIGN', 'SELECT', 'RULE_STATE', 'RULE_STATE2', 'SLOT_CONST', 'FMA'))

    def put_filter(node, layers):
        """A sprite filter's code; (offset, filter kind): expressions that only read their own sprite may be evaluated
        for the 64 sprites of a layer at once (MOOG_FILTER_EXPR_LANES) -- chosen when the rule ranges over at least 32
        slots: the per-lane stacks cost LDS (half a KB per stack entry), which costs resident envs on small programs."""
        code = _symbolic.emit(node, [], resolve_phase)
        off = put_code(code)
        d = _symbolic.depth(code)
        slots = sum(int(P.layer_nslots[layer_index(l)]) for l in layers)
        if all(ins['op'] in LANE_OPS for ins in code) and d <= 12 and slots >= 32:
            P.xstack_depth = max(int(P.xstack_depth), d)
            return off, _abi.MOOG_FILTER_EXPR_LANES
        return off, _abi.MOOG_FILTER_EXPR

    def put_expr(node=None, stores=None):
        """Appends postfix expression code (plus X_END) to program.dcode; returns its offset."""
        code = []
        if node is not None:
            _symbolic.emit(node, code, resolve_phase)
        for attr, n in (stores or {}).items():
            _symbolic.emit(n, code, resolve_phase)
            code.append(dict(op=_abi.MOOG_X_STORE, a=_symbolic.ATTRS.index(attr)))
        return put_code(code)

    for ri, (r, parent) in enumerate(flat_rules):
        R = P.rules[ri]
        R.parent = parent
        low = rules_lib.lookup_lowering(r)
        if isinstance(r, rules_lib.VanishOnContact):
            R.kind = _abi.MOOG_RULE_VANISH_ON_CONTACT
            R.l0, R.l1 = layer_index(r._layer), layer_index(r._contacting_layer)
        elif isinstance(r, rules_lib.VanishByFilter):
            R.kind = _abi.MOOG_RULE_VANISH_BY_FILTER
            R.l0 = layer_index(r._layer)
            if getattr(r, '_traced_filter', None) is not None:   # (a config-local Vanish subclass: expand_local_rule)
                R.filter, fnode = _abi.MOOG_FILTER_EXPR, r._traced_filter
            else:
                R.filter, fnode = rules_lib._classify_filter(r._filter_fn)
            if fnode is not None:
                R.xfilter, R.filter = put_filter(fnode, [r._layer])
        elif isinstance(r, rules_lib.ChangeLayer):
            R.kind = _abi.MOOG_RULE_CHANGE_LAYER
            R.l0, R.l1 = layer_index(r._old_layer), layer_index(r._new_layer)
            R.filter, fnode = rules_lib._classify_filter(r._filter_fn)
            if fnode is not None:
                R.xfilter, R.filter = put_filter(fnode, [r._old_layer])
        elif isinstance(r, rules_lib.ModifyOnContact):
            R.kind = _abi.MOOG_RULE_MODIFY_ON_CONTACT
            R.n_layers = _fill_layers(R.layers, list(r._layers_0), layer_index)
            R.n_layers1 = _fill_layers(R.layers1, list(r._layers_1), layer_index)
            R.xmod = R.xmod1 = -1
            R.filter, fnode = rules_lib._classify_filter(r._filter_0)
            if fnode is not None:
                R.xfilter = put_expr(fnode)
            R.filter1, fnode = rules_lib._classify_filter(r._filter_1)
            if fnode is not None:
                R.xfilter1 = put_expr(fnode)
            if r._modifier_0 is not None:
                mod, vec = _symbolic.trace_modifier(r._modifier_0)
                R.xmod = put_expr(stores=mod)
                R.i0 |= 2 if vec else 0
            if r._modifier_1 is not None:
                mod, vec = _symbolic.trace_modifier(r._modifier_1)
                R.xmod1 = put_expr(stores=mod)
                R.i0 |= 4 if vec else 0
        elif isinstance(r, rules_lib.CreateSprites):
            R.kind = _abi.MOOG_RULE_CREATE_SPRITES
            R.l0 = layer_index(r._layer)
            R.op = runtime_op_of_rule[ri]
            R.n_layers = _fill_layers(R.layers, list(r._without_overlapping), layer_index)
        elif isinstance(r, rules_lib.KeepNearCenter):
            R.kind = _abi.MOOG_RULE_KEEP_NEAR_CENTER
            R.l0 = layer_index(r._agent_layer)
            R.n_layers = _fill_layers(R.layers, r._layers_to_center, layer_index)
            R.p0, R.p1 = r._grid_cell
        elif isinstance(r, rules_lib.Fixation):
            R.kind = _abi.MOOG_RULE_FIXATION
            R.l0, R.l1 = layer_index(r._agent_layer), layer_index(r._fixation_layer)
            R.p0 = float(r._fixation_threshold)
        elif isinstance(r, rules_lib.Phase):
            R.kind = _abi.MOOG_RULE_PHASE
            R.i0 = sum(1 for x in r._one_time_rules for y in rules_lib.expand_local_rule(x)
                       if not getattr(y, 'host_side', False))
            R.p0 = r._duration
            if r._random_duration is not None:   # np.random.randint(lo, hi), drawn whenever the phase is reset
                R.op, R.p0, R.p2 = 1, float(r._random_duration[0]), float(r._random_duration[1])
                P.rule_state2 = 1
            if r._end_condition is not None:
                R.cond, _p, lay, node = rules_lib.classify_condition(r._end_condition)
                if lay is not None:
                    R.l0, R.l1 = layer_index(lay[0]), layer_index(lay[1])
                if node is not None:
                    R.xfilter = put_expr(node)
                if R.cond == _abi.MOOG_RCOND_BERNOULLI:
                    R.p1 = _p
        elif isinstance(r, rules_lib.PhaseSequence):
            R.kind = _abi.MOOG_RULE_PHASE_SEQUENCE
        elif isinstance(r, rules_lib.TimedRule):
            R.kind = _abi.MOOG_RULE_TIMED
            R.p0, R.p1 = r._step_interval
            if r._random is not None:   # a callable interval: one np.random.randint draw per reset (game_rules.TimedRule)
                R.op, R.p0, R.p1, R.p2 = r._random[:4]
                if R.op == 3:   # two draws: start = randint(p0, p2), then width = randint(p1, i0)
                    R.i0 = int(r._random[4])
                P.rule_state2 = 1
        elif isinstance(r, rules_lib.ConditionalRule):
            R.kind = _abi.MOOG_RULE_CONDITIONAL
            R.cond, R.p0, lay, node = r.classify()
            if lay is not None:
                R.l0, R.l1 = layer_index(lay[0]), layer_index(lay[1])
            if node is not None:
                R.xfilter = put_expr(node)
        elif isinstance(r, rules_lib.ModifySprites):
            R.kind, R.filter, fnode, mod, vec = r.classify()
            R.n_layers = _fill_layers(R.layers, r._layers, layer_index)
            if fnode is not None:
                R.xfilter, R.filter = put_filter(fnode, r._layers)
            if mod is not None:
                R.xmod = put_expr(stores=mod)
                R.i0 = int(bool(r._sample_one)) | (2 if vec else 0)
                # every sprite gets the same constants (e.g. `s.mass = 1.`, pacman.py:124-125) and none of the
                # stores moves vertices: the device evaluates once and stores lane-parallel
                const_only = all(_symbolic.is_constant(n) for n in mod.values())
                light = set(mod) <= {'x_vel', 'y_vel', 'angle_vel', 'mass', 'c0', 'c1', 'c2', 'opacity'}
                if const_only and light and fnode is None and not r._sample_one and R.filter == _abi.MOOG_FILTER_ALWAYS:
                    R.i0 |= 16
        elif isinstance(r, rules_lib._RuleDraws):
            R.kind, R.i0 = _abi.MOOG_RULE_DRAWS, int(r.n)
            P.rule_state2 = 1
        elif isinstance(r, rules_lib._ModifyTraced):
            R.kind, R.filter = _abi.MOOG_RULE_MODIFY_SPRITES, _abi.MOOG_FILTER_ALWAYS
            R.n_layers = _fill_layers(R.layers, [r.layer], layer_index)
            where = {}
            for d in r.draws:
                di = [i for i, (x, _) in enumerate(flat_rules) if x is d][0]
                for j in range(d.n):
                    where[d.first + j] = (di, j)

            def zip_resolver(key, name, _where=where, _layer=r.layer):
                if key == 'rdraw':
                    return _where[name]
                if key is None:   # a zipped partner layer: same number of sprites, fixed slots
                    if name in dynamic or _layer in dynamic or len(state[name]) != len(state[_layer]):
                        raise NotImplementedError('a rule that zips layers of different or changing sizes')
                    return layer_index(name)
                return resolve_phase(key, name)
            code = []
            for attr, n in r.mod.items():
                _symbolic.emit(n, code, zip_resolver)
                code.append(dict(op=_abi.MOOG_X_STORE, a=_symbolic.ATTRS.index(attr)))
            R.xmod = put_code(code)
            R.i0 = 2 if r.vec else 0
        elif isinstance(r, rules_lib.Portal):
            R.kind = _abi.MOOG_RULE_PORTAL
            R.l0, R.l1 = layer_index(r._teleporting_layer), layer_index(r._portal_layer)
        elif low is not None:
            d = low(r, layer_index)
            R.kind, R.l0, R.l1 = d['kind'], d.get('l0', 0), d.get('l1', 0)
            R.p0, R.p1, R.p2 = d.get('p0', 0.), d.get('p1', 0.), d.get('p2', 0.)
        else:
            # a config-local rule class: its step() is traced once on a symbolic state
            try:
                lname, mod, vec = _symbolic.trace_rule_step(r.step)
            except (NotImplementedError, AttributeError, TypeError) as exc:
                raise NotImplementedError(
                    'game rule %r has no device lowering (see game_rules.register_lowering): %s'
                    % (type(r).__name__, exc))
            R.kind, R.filter = _abi.MOOG_RULE_MODIFY_SPRITES, _abi.MOOG_FILTER_ALWAYS
            R.n_layers = _fill_layers(R.layers, [lname], layer_index)
            R.xmod = put_expr(stores=mod)
            R.i0 = 8 | (2 if vec else 0)   # only the layer's first sprite
    P.n_rules = len(flat_rules)
    for name in pstate_names:   # (their indices were handed out above: directly behind the config's rules)
        if P.n_rules >= _abi.MOOG_MAX_RULES:
            raise ValueError('too many game rules (a number the initializer keeps across episodes takes a rule slot)')
        assert pstate_slot[name] == int(P.n_rules)
        R = P.rules[P.n_rules]
        R.kind, R.parent = _abi.MOOG_RULE_STATE_SLOT, -1
        P.n_rules += 1
    if born_rule >= 0:
        if P.n_rules >= _abi.MOOG_MAX_RULES:
            raise ValueError('too many game rules (sprites built outside the initializer take a rule slot)')
        assert born_rule == int(P.n_rules)
        R = P.rules[P.n_rules]
        R.kind, R.parent, R.op = _abi.MOOG_RULE_STATE_SLOT, -1, 2
        P.n_rules += 1
    # state that belongs to forces: one never-reset scalar per DeterministicMazeWalk (its read position)
    for F, table in det_walks:
        if P.n_rules >= _abi.MOOG_MAX_RULES:
            raise ValueError('too many game rules (a DeterministicMazeWalk keeps its read position in a rule slot)')
        off = int(P.n_cand)
        if off + len(table) > _abi.MOOG_MAX_CAND:
            raise NotImplementedError('DeterministicMazeWalk: velocity table too long (%d values, %d free)'
                                      % (len(table), _abi.MOOG_MAX_CAND - off))
        for k, v in enumerate(table):
            P.cand[off + k] = v
        P.n_cand = off + len(table)
        F.i0, F.i1, F.symmetric = off, len(table) // 2, int(P.n_rules)
        R = P.rules[P.n_rules]
        R.kind, R.parent = _abi.MOOG_RULE_STATE_SLOT, -1
        P.n_rules += 1

    # ---- task -----------------------------------------------------------------------
    if isinstance(task, tasks_lib.CompositeTask):
        subtasks, P.timeout_steps = list(task._tasks), float(task._timeout_steps)
    else:
        subtasks, P.timeout_steps = [task], float('inf')
    if len(subtasks) > _abi.MOOG_MAX_TASKS:
        raise ValueError('too many tasks')
    for ti, t in enumerate(subtasks):
        T = P.tasks[ti]
        if isinstance(t, tasks_lib.ContactReward):
            T.kind = _abi.MOOG_TASK_CONTACT_REWARD
            T.n0 = _fill_layers(T.layers0, t._layers_0, layer_index)
            T.n1 = _fill_layers(T.layers1, t._layers_1, layer_index)
            T.p1 = float(t._reset_steps_after_contact)
            T.xcond = T.xreward = -1
            meta_sides.clear()
            meta_sides.update({0: list(t._layers_0), 1: list(t._layers_1)})
            if callable(t._reward):
                T.xreward = put_expr(_symbolic.trace_value(t._reward, 2))
            else:
                T.p0 = float(t._reward)
            if t._condition is not None:
                T.xcond = put_expr(_symbolic.trace_value(t._condition, 2))
        elif isinstance(t, tasks_lib.Reset):
            T.kind = _abi.MOOG_TASK_RESET
            cond, lname, val = t.classify(layer_names)
            T.cond, T.cond_layer = cond, layer_index(lname)
            if isinstance(val, _symbolic.Node):
                T.xcond = put_expr(val)
            else:
                T.cond_value = val
            T.p0, T.p1 = float(t.reward_value()), float(t._steps_after_condition)
            T.xreward = -1
            rnode = t.reward_node()
            if rnode is not None:   # reset.py:57: reward_fn(state), evaluated when the condition first holds
                T.xreward = put_expr(rnode)
        elif isinstance(t, tasks_lib.StayAlive):
            T.kind = _abi.MOOG_TASK_STAY_ALIVE
            T.i0, T.p0 = int(t._reward_period), float(t._reward_value)
        else:
            raise NotImplementedError('task %r is not lowered' % (type(t).__name__,))
    P.n_tasks = len(subtasks)

    # ---- action space -----------------------------------------------------------------
    def lower_action(A, space):
        if isinstance(space, action_spaces.Joystick):
            A.kind = _abi.MOOG_ACTION_JOYSTICK
            A.constrained_lr = int(space._constrained_lr)
        elif isinstance(space, action_spaces.Grid):
            A.kind = _abi.MOOG_ACTION_GRID
        elif isinstance(space, action_spaces.SetPosition):
            A.kind = _abi.MOOG_ACTION_SET_POSITION
            A.n_layers = _fill_layers(A.layers, space._action_layers, layer_index)
            A.momentum = float(space._inertia)
            return
        else:
            raise NotImplementedError('action space %r is not lowered' % (type(space).__name__,))
        A.n_layers = _fill_layers(A.layers, space._action_layers, layer_index)
        A.control_velocity = int(space._control_velocity)
        A.scaling_factor = float(space._scaling_factor)
        A.momentum = float(space._momentum)

    if isinstance(action_space, action_spaces.Composite):
        subs = list(action_space.action_spaces.values())
        if not 1 <= len(subs) <= _abi.MOOG_MAX_ACTIONS:
            raise ValueError('Composite takes 1..%d action spaces' % _abi.MOOG_MAX_ACTIONS)
        P.n_actions = len(subs)
        for k, sub in enumerate(subs):
            lower_action(P.action if k == 0 else P.more_actions[k - 1], sub)
    else:
        P.n_actions = 1
        lower_action(P.action, action_space)

    # ---- observer -----------------------------------------------------------------------
    obs_items = list(observers.items()) if observers else []
    renderers = [(k, o) for k, o in obs_items if isinstance(o, observers_lib.PILRenderer)]
    others = [o for _, o in obs_items if not isinstance(o, (observers_lib.PILRenderer, observers_lib.RawState))]
    if len(renderers) != 1 or others:
        raise NotImplementedError('exactly one PILRenderer observer is supported')
    obs_key, ren = renderers[0]
    Rn = P.render
    Rn.width, Rn.height = int(ren._image_size[0]), int(ren._image_size[1])
    Rn.aa = int(ren._anti_aliasing)   # the canvas is aa x the observation (pil_renderer.py:64-66)
    Rn.cmap = _abi.MOOG_CMAP_HSV if ren._cmap == 'hsv' else _abi.MOOG_CMAP_IDENTITY
    if isinstance(ren._polygon_modifier, polygon_modifiers.TorusGeometry):
        Rn.polymod = _abi.MOOG_POLYMOD_TORUS
    elif isinstance(ren._polygon_modifier, polygon_modifiers.FirstPersonAgent):
        Rn.polymod = _abi.MOOG_POLYMOD_FIRST_PERSON
        Rn.polymod_layer = layer_index(ren._polygon_modifier._agent_layer)
    elif isinstance(ren._polygon_modifier, polygon_modifiers.DoNothing):
        Rn.polymod = _abi.MOOG_POLYMOD_NONE
    else:
        raise NotImplementedError('polygon modifier %r' % (type(ren._polygon_modifier).__name__,))
    for c in range(3):
        Rn.bg[c] = int(ren._bg_color[c])

    # ---- vertex-count limits of the device kernels ------------------------------------------
    # Polygons of up to 128 vertices are integrated, moved and rasterised (the annulus of
    # first_person_predators_prey.py:79-82 has 102); the pairwise geometry (overlap tests,
    # contact search: lanes = vertices / edges of one wavefront) handles up to 64.
    if max([P.slot_vcap[sl] for sl in range(S)] or [0]) > 128:
        raise NotImplementedError('sprites with more than 128 vertices')
    tested = set()
    for fi in range(P.n_forces):
        F = P.forces[fi]
        if F.kind == _abi.MOOG_FORCE_COLLISION:
            tested.update(F.layers_a[i] for i in range(F.n_a))
            tested.update(F.layers_b[i] for i in range(F.n_b))
    for ri in range(P.n_rules):
        R = P.rules[ri]
        if R.kind in (_abi.MOOG_RULE_VANISH_ON_CONTACT, _abi.MOOG_RULE_PORTAL, _abi.MOOG_RULE_BOOSTER):
            tested.update((R.l0, R.l1))
        elif R.kind == _abi.MOOG_RULE_MODIFY_ON_CONTACT:
            tested.update(R.layers[i] for i in range(R.n_layers))
            tested.update(R.layers1[i] for i in range(R.n_layers1))
        elif R.kind == _abi.MOOG_RULE_CREATE_SPRITES:
            tested.update(R.layers[i] for i in range(R.n_layers))
            if R.n_layers or (P.ops[R.op].disjoint & 1):
                tested.add(R.l0)
    for ti in range(P.n_tasks):
        T = P.tasks[ti]
        if T.kind == _abi.MOOG_TASK_CONTACT_REWARD:
            tested.update(T.layers0[i] for i in range(T.n0))
            tested.update(T.layers1[i] for i in range(T.n1))
    for oi in range(P.n_ops):
        G = P.ops[oi]
        if G.runtime or not (G.avoid_ops or (G.disjoint & 1)):
            continue
        tested.add(P.slot_layer[G.slot0])
        for oj in range(oi):
            if (G.avoid_ops >> oj) & 1:
                tested.add(P.slot_layer[P.ops[oj].slot0])
    for sl in range(S):
        if P.slot_vcap[sl] > 64 and P.slot_layer[sl] in tested:
            raise NotImplementedError(
                'sprites with more than 64 vertices cannot take part in overlap tests '
                '(layer %r)' % (layer_names[P.slot_layer[sl]],))

    layer_slots = {name: (P.layer_slot0[i], P.layer_nslots[i]) for i, name in enumerate(layer_names)}
    # rule_ref_index: program rule -> index of the config's rule object it stands for in a pre-order walk of the config's
    # rule forest (-1: a rule the lowering added: the extra parts of an expanded config-local rule, state slots of forces)
    # pstate_slots: (attribute of the initializer's object kept across episodes, rule slot that holds it per env)
    c = Compiled(P, layer_names, layer_slots, obs_key, _abi.layout_of(P), [],
                 rule_ref_index + [-1] * (int(P.n_rules) - len(rule_ref_index)), sorted(pstate_slot.items()),
                 # (slot, metadata key, cell that holds the look-ahead's exit, {exit: value}) of the metadata values
                 # an initializer's look-ahead decides
                 [(slot_of[id(sp)], key, cell, table) for sp, key, cell, table in getattr(tr, 'dynamic_meta', [])
                  if id(sp) in slot_of],
                 # PILRenderer(color_to_rgb=<a callable>): evaluated on the host (environment.py _refresh_colors)
                 ren.color_to_rgb if ren._cmap == 'callable' else None, layer_n_init)
    # shape id -> Sprite.shape value (sprite.py:517-523): the name, or 'custom' for raw vertices
    c.shape_names.extend(k[1] if k[0] == 'name' else 'custom' for k, _ in shapes.entries)
    return c
