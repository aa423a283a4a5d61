"""Shared test helpers: oracle binding, golden fixtures <-> state records.

The oracle (oracle/libmoog_oracle.so) is test infrastructure; only tests,
__graft_entry__.smoke() and bench.py's cpu_baseline leg load it.
"""
import ctypes
import functools
import os
import subprocess

import numpy as np

from moog import _abi, _compiler
from moog_demos import example_configs

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
GOLDEN = os.path.join(REPO, 'tests', 'golden')
ORACLE_DIR = os.path.join(REPO, 'oracle')
ORACLE_SO = os.path.join(ORACLE_DIR, 'libmoog_oracle.so')

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_bp = ctypes.POINTER(ctypes.c_uint8)


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)


@functools.lru_cache(maxsize=None)
def oracle():
    """Builds (if needed) and loads the CPU oracle."""
    src = os.path.join(ORACLE_DIR, 'moog_oracle.c')
    hdr = os.path.join(REPO, 'include', 'moog_engine.h')
    if (not os.path.exists(ORACLE_SO) or
            os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(['make', '-C', ORACLE_DIR, '-B', 'libmoog_oracle.so'],
                              stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(ORACLE_SO)
    lib.oracle_program_sizeof.restype = ctypes.c_int64
    assert lib.oracle_program_sizeof() == ctypes.sizeof(_abi.Program)
    return lib


class OracleEnv(object):
    """Host-side state records + oracle entry points for one compiled program."""

    def __init__(self, compiled, n_envs=1, seed=0, env_index0=0):
        self.c = compiled
        self.P = compiled.program
        self.L = compiled.layout
        self.n = n_envs
        self.seed, self.env_index0 = seed, env_index0
        self.f64 = np.zeros((n_envs, self.L.f64_per_env), np.float64)
        self.i32 = np.zeros((n_envs, self.L.i32_per_env), np.int32)
        self.reward = np.zeros(n_envs)
        self.discount = np.zeros(n_envs)
        self.step_type = np.zeros(n_envs, np.int32)
        self.image = np.zeros((n_envs, self.P.render.height, self.P.render.width, 3), np.uint8)
        self.lib = oracle()
        # PILRenderer(color_to_rgb=<a callable>): Python, evaluated here per live sprite and handed to the oracle's renderer
        # per (env, slot) -- the oracle's counterpart of moog_engine_set_color_override
        self.color_fn = getattr(compiled, 'color_fn', None)

    def _color_override(self):
        from moog import environment
        S = self.L.S
        rgb = np.zeros((self.n, S), np.uint32)
        alive = (self.i32[:, self.L.o_flags:self.L.o_flags + S] & _abi.MOOG_F_ALIVE) != 0
        col = self.f64[:, self.L.o_color:self.L.o_color + 3 * S].reshape(self.n, S, 3)
        for i, s_ in zip(*np.nonzero(alive)):
            rgb[i, s_] = environment.BatchedEnvironment._call_color_fn(self.color_fn, col[i, s_])
        return rgb

    def _inj(self, u):
        if u is None:
            return None, 0
        u = np.ascontiguousarray(u, np.float64).reshape(self.n, -1)
        return u, u.shape[1]

    def reset(self, uniforms=None, mask=None, render=True):
        if self.color_fn is not None and render:   # (the colours exist once the state does: draw afterwards)
            self.reset(uniforms, mask, render=False)
            self.render()
            return
        u, nu = self._inj(uniforms)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.lib.oracle_reset(ctypes.byref(self.P), _ptr(self.f64, _dp), _ptr(self.i32, _ip),
                              self.n, _ptr(m, _bp), _ptr(u, _dp), nu,
                              ctypes.c_uint64(self.seed), ctypes.c_int64(self.env_index0),
                              _ptr(self.reward, _dp), _ptr(self.discount, _dp),
                              _ptr(self.step_type, _ip), _ptr(self.image, _bp) if render else None)

    def step(self, actions, uniforms=None, render=True):
        if self.color_fn is not None and render:
            self.step(actions, uniforms, render=False)
            self.render()
            return
        u, nu = self._inj(uniforms)
        # float32 actions (the reference's Joystick spec): the scaling is a float32 product
        self.lib.oracle_set_action_f32(1 if getattr(actions, 'dtype', None) == np.float32 else 0)
        if self.P.n_actions > 1:   # Composite: [n, n_actions, 2] (a Grid move in component 0)
            a = np.ascontiguousarray(actions, np.float64).reshape(self.n, 2 * self.P.n_actions)
            af, ai = a, None
        elif self.P.action.kind == _abi.MOOG_ACTION_GRID:
            a = np.ascontiguousarray(actions, np.int32).reshape(self.n)
            af, ai = None, a
        else:
            a = np.ascontiguousarray(actions, np.float64).reshape(self.n, 2)
            af, ai = a, None
        self.lib.oracle_step(ctypes.byref(self.P), _ptr(self.f64, _dp), _ptr(self.i32, _ip),
                             self.n, _ptr(af, _dp), _ptr(ai, _ip), _ptr(u, _dp), nu,
                             ctypes.c_uint64(self.seed), ctypes.c_int64(self.env_index0),
                             _ptr(self.reward, _dp), _ptr(self.discount, _dp),
                             _ptr(self.step_type, _ip), _ptr(self.image, _bp) if render else None)

    def physics(self, uniforms=None, substep=False):
        u, nu = self._inj(uniforms)
        fn = self.lib.oracle_substep if substep else self.lib.oracle_physics
        fn(ctypes.byref(self.P), _ptr(self.f64, _dp), _ptr(self.i32, _ip), self.n,
           _ptr(u, _dp), nu, ctypes.c_uint64(self.seed), ctypes.c_int64(self.env_index0))

    def render(self):
        rgb = None
        if self.color_fn is not None:
            rgb = self._color_override()
            self.lib.oracle_set_color_override(rgb.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(0))
        try:
            return self._render()
        finally:
            if rgb is not None:
                self.lib.oracle_set_color_override(None, ctypes.c_int64(0))

    def _render(self):
        self.lib.oracle_render(ctypes.byref(self.P), _ptr(self.f64, _dp), _ptr(self.i32, _ip),
                               self.n, _ptr(self.image, _bp))
        return self.image


@functools.lru_cache(maxsize=None)
def compiled(name):
    return _compiler.compile_config(layer_capacity=example_configs.capacity(name),
                                    **example_configs.load(name))


@functools.lru_cache(maxsize=None)
def fixture(name, seed=0):
    with np.load(os.path.join(GOLDEN, '%s_s%d.npz' % (name, seed))) as z:
        return {k: z[k] for k in z.files}


def ref_rules(c, n_ref):
    """(program rule, index of the config's rule object in the fixture's pre-order walk) for the rules that stand for a
    rule object of the config (a config-local rule may be lowered to several program rules, _compiler.rule_ref_index)."""
    ref = getattr(c, 'rule_ref_index', None) or list(range(c.program.n_rules))
    return [(r, k) for r, k in enumerate(ref) if 0 <= k < n_ref]


def portal_rule_mask(P):
    m = 0
    for r in range(P.n_rules):
        if P.rules[r].kind == _abi.MOOG_RULE_PORTAL:
            m |= 1 << r
    return m


def records_from_fixture(fx, t, c, f64=None, i32=None, env=0):
    """Writes the reference state of call `t` into state records (env row)."""
    P, L = c.program, c.layout
    S = L.S
    assert list(fx['layer_caps']) == [P.layer_nslots[i] for i in range(P.n_layers)], \
        'fixture slot table does not match the compiled program'
    if f64 is None:
        f64 = np.zeros((1, L.f64_per_env))
        i32 = np.zeros((1, L.i32_per_env), np.int32)
    f, q = f64[env], i32[env]
    f[:] = 0
    q[:] = 0
    nz = lambda a: np.nan_to_num(np.asarray(a, np.float64), nan=0.0, posinf=np.inf, neginf=-np.inf)
    f[L.o_pos:L.o_pos + 2 * S] = nz(fx['pos'][t]).ravel()
    f[L.o_vel:L.o_vel + 2 * S] = nz(fx['vel'][t]).ravel()
    f[L.o_angle:L.o_angle + S] = nz(fx['angle'][t])
    f[L.o_angvel:L.o_angvel + S] = nz(fx['angvel'][t])
    f[L.o_mass:L.o_mass + S] = nz(fx['mass'][t])
    f[L.o_color:L.o_color + 3 * S] = nz(fx['color'][t]).ravel()
    f[L.o_inertia:L.o_inertia + 2 * S] = nz(fx['inertia'][t]).ravel()
    f[L.o_maxr:L.o_maxr + S] = nz(fx['maxr'][t])
    am = np.asarray(fx['action_mem'][t], np.float64).reshape(-1)
    f[L.o_action:L.o_action + len(am)] = am
    tc = np.asarray(fx['task_counters'][t], np.float64).reshape(-1)
    for k in range(P.n_tasks):
        f[L.o_task + k] = tc[k] if k < len(tc) and not np.isnan(tc[k]) else np.inf
    # the fixture lists env.game_rules (top level); the program table is the pre-order forest
    rc = np.asarray(fx['rule_counters'][t], np.float64).reshape(-1)
    tops = [r for r in range(P.n_rules) if P.rules[r].parent < 0]
    for r in range(P.n_rules):
        f[L.o_rule + r] = np.inf
    for k, r in enumerate(tops):
        if k < len(rc) and not np.isnan(rc[k]):
            f[L.o_rule + r] = rc[k]
    if 'rule_counters_flat' in fx:   # newer fixtures: every rule of the pre-order forest
        flat = np.asarray(fx['rule_counters_flat'][t], np.float64).reshape(-1)
        for r, k in ref_rules(c, len(flat)):
            if not np.isnan(flat[k]):
                f[L.o_rule + r] = flat[k]
    if P.rule_state2:   # the duration a Phase drew when it was reset
        flat2 = np.asarray(fx['rule_counters2_flat'][t], np.float64).reshape(-1)
        for r in range(P.n_rules):
            f[L.o_rule2 + r] = 0.0
        for r, k in ref_rules(c, len(flat2)):
            f[L.o_rule2 + r] = 0.0 if np.isnan(flat2[k]) else flat2[k]
    for slot, key, cell, table in (getattr(c, 'dynamic_meta', None) or []):
        # what the initializer's look-ahead stored in a sprite's metadata: the engine keeps the exit the look-ahead took
        keys = [str(x) for x in np.asarray(fx['meta_keys'][t]).reshape(-1)]
        val = float(np.asarray(fx['meta_vals'][t])[slot, keys.index(key)])
        match = [ex for ex, v in table.items() if v == val or (np.isnan(v) and np.isnan(val))]
        assert match, ('metadata value the look-ahead cannot produce', key, val, table)
        f[L.o_hdraw + cell] = float(match[0])
    if getattr(c, 'pstate_slots', None):   # numbers the initializer keeps across episodes: never cleared by resets
        names = [str(x) for x in np.asarray(fx['init_state_names'][t]).reshape(-1)]
        vals = np.asarray(fx['init_state'][t], np.float64).reshape(-1)
        for name, ri in c.pstate_slots:
            f[L.o_rule + ri] = vals[names.index(name)]
            f[L.o_rule2 + ri] = 1.0
    if 'force_state' in fx:   # forces with state of their own (DeterministicMazeWalk's read position: a MOOG_RULE_STATE_SLOT)
        fs = np.asarray(fx['force_state'][t], np.float64).reshape(-1)
        for fi in range(min(P.n_forces, len(fs))):
            if P.forces[fi].kind == _abi.MOOG_FORCE_MAZE_WALK_DET:
                f[L.o_rule + P.forces[fi].symmetric] = fs[fi]
    pm = portal_rule_mask(P)
    for s in range(S):
        nv = int(fx['nverts'][t][s])
        assert nv <= P.slot_vcap[s]
        o = L.o_verts + 2 * P.slot_voff[s]
        f[o:o + 2 * nv] = fx['verts'][t][s, :nv].ravel()
        fl = 0
        if fx['alive'][t][s]:
            fl |= _abi.MOOG_F_ALIVE
        if fx['sym_circle'][t][s]:
            fl |= _abi.MOOG_F_SYM_CIRCLE
        if fx['vel_f32'][t][s]:
            fl |= _abi.MOOG_F_VEL_F32
        if fx['angvel_f32'][t][s]:
            fl |= _abi.MOOG_F_ANGVEL_F32
        q[L.o_flags + s] = fl
        q[L.o_nverts + s] = nv if fx['alive'][t][s] else 0
        q[L.o_opacity + s] = fx['opacity'][t][s]
        q[L.o_tele + s] = pm if fx['tele'][t][s] else 0
        if P.vel_alias:
            q[L.o_valias + s] = fx['vel_group'][t][s] if 'vel_group' in fx else 0
        if P.sprite_factors:
            f[L.o_scale + s] = nz(fx['scale'][t][s])
            f[L.o_aspect + s] = nz(fx['aspect'][t][s])
            q[L.o_fmask + s] = fx['fmask'][t][s]
    q[L.o_step_count] = fx['step_count'][t]
    q[L.o_reset_next] = fx['reset_next'][t]
    if P.maze.random:   # the episode's maze: what Maze.from_state reads off the recorded wall squares
        rows = maze_rows_from_fixture(fx, t, c)
        q[L.o_maze:L.o_maze + len(rows)] = rows
    return f64, i32


def maze_rows_from_fixture(fx, t, c):
    """Row bit masks (bit column = wall) of the maze whose wall squares call t of a recording holds."""
    P = c.program
    n = P.maze.size
    rows = np.zeros(_abi.MOOG_MAX_MAZE, np.int64)
    for oi in range(P.n_ops):
        if P.ops[oi].cell_sel != _abi.MOOG_CELL_WALL_RANK:
            continue
        s = P.ops[oi].slot0
        if not fx['alive'][t][s]:
            continue
        v = fx['verts'][t][s, :4]
        j, i = int(round(v[:, 0].min() * n)), int(round(v[:, 1].min() * n))
        rows[i] |= 1 << j
    return rows.astype(np.int32)


def state_diff(fx, t, c, f64, i32, env=0):
    """Max abs differences between the records and the reference state of call t.

    Returns dict(float=max abs error over live sprites' float state incl. vertices,
    ints_ok=bool for alive mask / nverts / counters / step bookkeeping, detail=str)."""
    P, L = c.program, c.layout
    S = L.S
    f, q = f64[env], i32[env]
    alive_ref = fx['alive'][t].astype(bool)
    alive = (q[L.o_flags:L.o_flags + S] & _abi.MOOG_F_ALIVE).astype(bool)
    detail = []
    ints_ok = True
    if not np.array_equal(alive, alive_ref):
        ints_ok = False
        detail.append('alive %s vs %s' % (alive.astype(int), alive_ref.astype(int)))
    err = {}
    live = alive_ref & alive

    def cmp(name, got, ref):
        got, ref = np.asarray(got, np.float64)[live], np.asarray(ref, np.float64)[live]
        if got.size == 0:
            err[name] = 0.0
            return
        with np.errstate(invalid='ignore'):
            d = np.abs(got - ref)
        d = np.where(np.isnan(got) & np.isnan(ref), 0.0, d)
        d = np.where(np.isinf(got) & (got == ref), 0.0, d)
        err[name] = float(np.max(d)) if not np.isnan(d).any() else np.inf

    cmp('pos', f[L.o_pos:L.o_pos + 2 * S].reshape(S, 2), fx['pos'][t])
    cmp('vel', f[L.o_vel:L.o_vel + 2 * S].reshape(S, 2), fx['vel'][t])
    cmp('angle', f[L.o_angle:L.o_angle + S], fx['angle'][t])
    cmp('angvel', f[L.o_angvel:L.o_angvel + S], fx['angvel'][t])
    cmp('mass', f[L.o_mass:L.o_mass + S], fx['mass'][t])
    cmp('color', f[L.o_color:L.o_color + 3 * S].reshape(S, 3), fx['color'][t])
    verr = 0.0
    for s in range(S):
        if not live[s]:
            continue
        nv = int(fx['nverts'][t][s])
        if int(q[L.o_nverts + s]) != nv:
            ints_ok = False
            detail.append('nverts[%d] %d vs %d' % (s, q[L.o_nverts + s], nv))
            continue
        o = L.o_verts + 2 * P.slot_voff[s]
        verr = max(verr, float(np.max(np.abs(f[o:o + 2 * nv] - fx['verts'][t][s, :nv].ravel()))))
        fl = int(q[L.o_flags + s])
        for bit, key in ((_abi.MOOG_F_SYM_CIRCLE, 'sym_circle'), (_abi.MOOG_F_VEL_F32, 'vel_f32'),
                         (_abi.MOOG_F_ANGVEL_F32, 'angvel_f32')):
            if bool(fl & bit) != bool(fx[key][t][s]):
                ints_ok = False
                detail.append('flag %s slot %d' % (key, s))
        if bool(q[L.o_tele + s]) != bool(fx['tele'][t][s]):
            ints_ok = False
            detail.append('tele slot %d' % s)
        if int(q[L.o_opacity + s]) != int(fx['opacity'][t][s]):
            ints_ok = False
            detail.append('opacity slot %d' % s)
    err['verts'] = verr
    if P.sprite_factors and 'scale' in fx:
        cmp('scale', f[L.o_scale:L.o_scale + S], fx['scale'][t])
        cmp('aspect', f[L.o_aspect:L.o_aspect + S], fx['aspect'][t])
        relevant = sum(1 << b for b in (_abi.MOOG_FAC_SCALE, _abi.MOOG_FAC_ASPECT, _abi.MOOG_FAC_C0,
                                        _abi.MOOG_FAC_C1, _abi.MOOG_FAC_C2, _abi.MOOG_FAC_MASS))
        for s in range(S):
            if live[s] and (int(q[L.o_fmask + s]) ^ int(fx['fmask'][t][s])) & relevant:
                ints_ok = False
                detail.append('float32 factor mask slot %d: %x vs %x' % (s, q[L.o_fmask + s], fx['fmask'][t][s]))
    if P.vel_alias and 'vel_group' in fx:
        # sprites sharing one velocity ndarray: same partition (the group ids are arbitrary)
        def canon(g):
            g = np.where(live, np.asarray(g), 0)
            return [0 if g[s] == 0 else 1 + int(np.flatnonzero(g == g[s])[0]) for s in range(S)]
        if canon(q[L.o_valias:L.o_valias + S]) != canon(fx['vel_group'][t]):
            ints_ok = False
            detail.append('shared-velocity groups %s vs %s' % (
                canon(q[L.o_valias:L.o_valias + S]), canon(fx['vel_group'][t])))
    if int(q[L.o_step_count]) != int(fx['step_count'][t]) or \
            int(q[L.o_reset_next]) != int(fx['reset_next'][t]):
        ints_ok = False
        detail.append('step_count/reset_next %d/%d vs %d/%d' % (
            q[L.o_step_count], q[L.o_reset_next], fx['step_count'][t], fx['reset_next'][t]))
    if 'rule_counters_flat' in fx:   # every rule of the pre-order forest (NaN: the reference rule keeps no number)
        flat = np.asarray(fx['rule_counters_flat'][t], np.float64).reshape(-1)
        for r, k in ref_rules(c, len(flat)):
            if not np.isnan(flat[k]) and f[L.o_rule + r] != flat[k]:
                ints_ok = False
                detail.append('rule %d state: %r vs %r' % (r, f[L.o_rule + r], flat[k]))
    if P.rule_state2:
        flat2 = np.asarray(fx['rule_counters2_flat'][t], np.float64).reshape(-1)
        for r, k in ref_rules(c, len(flat2)):
            if P.rules[r].op == 1 and P.rules[r].kind == _abi.MOOG_RULE_PHASE and f[L.o_rule2 + r] != flat2[k]:
                ints_ok = False
                detail.append('phase %d duration: %r vs %r' % (r, f[L.o_rule2 + r], flat2[k]))
    if getattr(c, 'pstate_slots', None):
        names = [str(x) for x in np.asarray(fx['init_state_names'][t]).reshape(-1)]
        vals = np.asarray(fx['init_state'][t], np.float64).reshape(-1)
        for name, ri in c.pstate_slots:
            if f[L.o_rule + ri] != vals[names.index(name)]:
                ints_ok = False
                detail.append('initializer state %s: %r vs %r' % (name, f[L.o_rule + ri], vals[names.index(name)]))
    if 'force_state' in fx:
        fs = np.asarray(fx['force_state'][t], np.float64).reshape(-1)
        for fi in range(min(P.n_forces, len(fs))):
            if P.forces[fi].kind == _abi.MOOG_FORCE_MAZE_WALK_DET and f[L.o_rule + P.forces[fi].symmetric] != fs[fi]:
                ints_ok = False
                detail.append('force %d state: %r vs %r' % (fi, f[L.o_rule + P.forces[fi].symmetric], fs[fi]))
    if P.maze.random:
        rows = maze_rows_from_fixture(fx, t, c)
        if not np.array_equal(q[L.o_maze:L.o_maze + len(rows)], rows):
            ints_ok = False
            detail.append('maze rows %s vs %s' % (list(q[L.o_maze:L.o_maze + P.maze.size]), list(rows[:P.maze.size])))
    ref_am = np.asarray(fx['action_mem'][t], np.float64).reshape(-1)
    am = float(np.max(np.abs(f[L.o_action:L.o_action + len(ref_am)] - ref_am)))
    err['action_mem'] = am
    tc = np.asarray(fx['task_counters'][t], np.float64).reshape(-1)
    for k in range(min(P.n_tasks, len(tc))):
        if not np.isnan(tc[k]) and f[L.o_task + k] != tc[k]:
            ints_ok = False
            detail.append('task counter %d: %r vs %r' % (k, f[L.o_task + k], tc[k]))
    rc = np.asarray(fx['rule_counters'][t], np.float64).reshape(-1)
    tops = [r for r in range(P.n_rules) if P.rules[r].parent < 0]
    for k, r in enumerate(tops[:len(rc)]):
        if not np.isnan(rc[k]) and f[L.o_rule + r] != rc[k]:
            ints_ok = False
            detail.append('rule counter %d: %r vs %r' % (k, f[L.o_rule + r], rc[k]))
    if 'rule_counters_flat' in fx:
        flat = np.asarray(fx['rule_counters_flat'][t], np.float64).reshape(-1)
        for r, k in ref_rules(c, len(flat)):
            if not np.isnan(flat[k]) and f[L.o_rule + r] != flat[k]:
                ints_ok = False
                detail.append('rule state %d: %r vs %r' % (r, f[L.o_rule + r], flat[k]))
    return dict(float=max(err.values()), err=err, ints_ok=ints_ok, detail='; '.join(detail))


def action_of(fx, t):
    """The action of recorded call t, in the dtype the reference was handed (float32 for recordings made with the
    Joystick spec's own dtype, tests/golden/make_golden.py `__action_f32__`)."""
    a = np.asarray(fx['action'][t])
    if 'action_f32' in fx and int(fx['action_f32']):
        a = a.astype(np.float32)
    return a


def uniforms_of(fx, t):
    n = int(fx['n_uniforms'][t])
    return np.ascontiguousarray(fx['uniforms'][t, :max(n, 1)]).copy() if n else None


def same_or_nan(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))
