"""Generates the golden vectors in this directory from the REAL reference.

Run in the build container only (needs /root/reference, matplotlib, Pillow):

    PYTHONPATH=oracle/shim:/root/reference MPLBACKEND=Agg python tests/golden/make_golden.py

It imports the reference `moog` package unmodified (behind the dm_env stand-in of
oracle/shim), steps the BASELINE.json configs with recorded randomness, and
writes neutral .npz fixtures (no layout of this repo is baked in):

  <config>_s<seed>.npz   per-call sprite tables, bookkeeping, frames, the uniforms
                         each call consumed, sub-step states of the first steps
  predicates.npz         matplotlib Path.intersects_path / contains_points corpus
  raster.npz             Pillow ImageDraw.polygon (RGBA blend) coverage corpus
  collisions_kat.npz     outcome of the reference's known-answer scenarios
                         (tests/moog/physics/test_collisions.py:101-293) as
                         computed by the reference itself

Randomness: the reference draws from numpy's global MT19937.  To
make runs reproducible by an engine with a different generator, np.random.uniform
/ choice / randint are replaced by equivalents that consume one recorded uniform
u in [0,1) each: uniform = low + (high-low)*u (numpy's own formula),
binomial(1, p) = int(u < p), rand(shape) = one u per element, shuffle(list) = numpy's backward Fisher-Yates with j = int(u*(i+1)),
choice(n, size=k, replace=False) = the first k steps of a forward Fisher-Yates of range(n), randint(a, b, size) = one u per element,
choice(n) = int(u*n), choice(n, p) = searchsorted(cumsum(p)/sum(p), u, 'right') (numpy's
own formula), randint(a,b) = a + int(u*(b-a)).
"""
import importlib.util
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.normpath(os.path.join(HERE, '..', '..'))
CFG_DIR = os.path.join(REPO, 'moog.github.io_amd', 'moog_demos', 'example_configs')

from moog import environment  # noqa: E402  (the reference package)
from moog import sprite as ref_sprite  # noqa: E402
from moog.physics import collisions as ref_collisions  # noqa: E402
from matplotlib import path as mpl_path  # noqa: E402
from PIL import Image, ImageDraw  # noqa: E402

VMAX = 30
SNAP_VMAX = 104   # first_person_predators_prey's annulus has 102 vertices


class Tape(object):
    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.cur = []

    def u(self):
        v = float(self.rs.random_sample())
        self.cur.append(v)
        return v

    def take(self):
        out, self.cur = self.cur, []
        return out


TAPE = None
DYNAMIC_LAYERS = ()
STATE_VMAX = VMAX


def _uniform(low=0.0, high=1.0, size=None):
    assert size is None
    return low + (high - low) * TAPE.u()


def _choice(a, size=None, replace=True, p=None):
    n = a if isinstance(a, (int, np.integer)) else len(a)
    if size is not None:
        # k distinct indices (maze.py:212 sample_distinct_open_points): the first k steps of a forward
        # Fisher-Yates shuffle of range(n), one tape uniform per step
        assert not replace and p is None and isinstance(a, (int, np.integer))
        perm = list(range(n))
        for t in range(int(size)):
            j = t + int(TAPE.u() * (n - t))
            perm[t], perm[j] = perm[j], perm[t]
        return np.array(perm[:int(size)])
    if p is not None:   # numpy's own algorithm (legacy RandomState.choice with p)
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        idx = int(np.searchsorted(cdf, TAPE.u(), side='right'))
        return idx if isinstance(a, (int, np.integer)) else a[idx]
    idx = 0 if n == 1 else int(TAPE.u() * n)
    return idx if isinstance(a, (int, np.integer)) else a[idx]


def _randint(low, high=None, size=None, dtype=int):
    if high is None:
        low, high = 0, low
    if size is not None:   # one tape uniform per element, row-major (maze_generators.py:143)
        return np.array([low + int(TAPE.u() * (high - low)) for _ in range(int(np.prod(size)))]).reshape(size)
    return low + int(TAPE.u() * (high - low))


def _shuffle(x):
    """np.random.shuffle of a list (maze_generators.py:131): numpy's backward Fisher-Yates order, one tape
    uniform per swap: for i = n - 1 .. 1: j = int(u * (i + 1)); swap(x[i], x[j])."""
    for i in range(len(x) - 1, 0, -1):
        j = int(TAPE.u() * (i + 1))
        x[i], x[j] = x[j], x[i]


def _binomial(n, p, size=None):
    assert n == 1
    if size is not None:   # one tape uniform per element (multi_tracking_with_feature.py:136)
        return np.array([int(TAPE.u() < p) for _ in range(int(np.prod(size)))]).reshape(size)
    return int(TAPE.u() < p)


def _rand(*shape):
    """np.random.rand(d0, d1, ...): one tape uniform per element, row-major."""
    if not shape:
        return TAPE.u()
    return np.array([TAPE.u() for _ in range(int(np.prod(shape)))]).reshape(shape)


def track_and_fixate(env, t, rs):
    """Policy for multi_tracking_with_feature: hold the gaze on the fixation cross (with a lapse now and then, so that
    the fixation counter restarts), then on the first target once its bar has turned."""
    phase = env.meta_state.get('phase', '')
    if phase == 'change':
        goal = np.array(env.state['targets'][0].position, dtype=float)
    else:
        goal = np.array([0.5, 0.5])
    if t % 41 == 5 and phase in ('fixation', 'change'):
        return rs.uniform(0., 1., size=2)      # a lapse
    return np.clip(goal + rs.uniform(-0.03, 0.03, size=2), 0., 1.)


def _seek_cover(index):
    def policy(env, t, rs):
        """Policy for match_to_sample: once the response phase lets the agent move, steer it onto cover `index` (0 is
        the one whose disc the cue matches)."""
        if env.meta_state.get('phase', '') != 'response':
            return rs.uniform(-1., 1., size=2)
        covers = env.state['covers']
        d = np.array(covers[index % len(covers)].position, dtype=float) - np.array(env.state['agent'][0].position, dtype=float)
        return np.clip(d / max(np.linalg.norm(d), 1e-9) + rs.uniform(-0.2, 0.2, size=2), -1., 1.)
    return policy


seek_match, seek_other = _seek_cover(0), _seek_cover(1)


def _go_to(env, t, rs):
    """Policy for lookahead_zoo (Joystick): after a few idle steps steer the agent to the left pocket, the right pocket
    or the floor, a different goal every episode."""
    ep = _go_to.episode = getattr(_go_to, 'episode', 0) + (1 if env.step_count == 0 else 0)
    if env.step_count < 6:
        return rs.uniform(-0.2, 0.2, size=2)
    goal = [np.array([0.2, 0.3]), np.array([0.8, 0.3]), np.array([0.5, 0.1])][ep % 3]
    d = goal - np.array(env.state['agent'][0].position, dtype=float) - 6. * np.array(env.state['agent'][0].velocity, dtype=float)
    return np.clip(4. * d + rs.uniform(-0.1, 0.1, size=2), -1., 1.)


def _answer(side_of_episode, idle=(2, 5)):
    def policy(env, t, rs):
        """Policy for bounce_box_contact_prediction (Grid actions): watch for a while, then walk the token into one of
        the two response boxes -- left = 'they will touch', right = 'they will not' -- alternating by episode."""
        box = policy.episode = getattr(policy, 'episode', 0)
        if env.step_count == 0:
            policy.episode = box + 1
        if env.step_count < 22:
            return int(rs.randint(*idle))      # up / down / nothing: stays between the boxes
        return side_of_episode[box % len(side_of_episode)]
    return policy


def patch_numpy_random():
    np.random.uniform = _uniform
    np.random.rand = _rand
    np.random.choice = _choice
    np.random.randint = _randint
    np.random.binomial = _binomial
    np.random.shuffle = _shuffle


SHIPPED = ('pong', 'chase_avoid_torus', 'colliding_predators', 'functional_maze', 'falling_balls',
           'first_person_predators_prey', 'cleanup', 'pacman', 'parallelogram_catch')
# (multi_tracking_with_feature takes its number of targets where the others take a level)


def load_amd_config(name):
    """The five shipped configs come from the reference's own files; the scaled
    variants from this repo's recipes, run against the reference package."""
    if name in SHIPPED:
        return importlib.import_module('moog_demos.example_configs.' + name).get_config(0)
    if name in ('parallelogram_catch_l1', 'parallelogram_catch_l2'):   # moving pellets
        return importlib.import_module('moog_demos.example_configs.parallelogram_catch').get_config(int(name[-1]))
    if name in ('red_green', 'red_green_l1', 'red_green_l2', 'red_green_l3'):   # (the level is the number of obstacles)
        return importlib.import_module('moog_demos.example_configs.red_green').get_config(
            int(name[-1]) if name[-1].isdigit() else 0)
    if name in ('bounce_box_contact_prediction', 'bounce_box_contact_prediction_l1'):   # (the level is `translucent_occluder`)
        return importlib.import_module('moog_demos.example_configs.bounce_box_contact_prediction').get_config(
            name.endswith('_l1'))
    if name in ('predators_arena_l1', 'predators_arena_l2', 'predators_arena_l3'):   # (the level is the number of predators)
        return importlib.import_module('moog_demos.example_configs.predators_arena').get_config(int(name[-1]))
    if name in ('match_to_sample_l2', 'match_to_sample_l3', 'match_to_sample_l4'):   # (the level is the number of targets)
        return importlib.import_module('moog_demos.example_configs.match_to_sample').get_config(int(name[-1]))
    if name in ('multi_tracking_with_feature_l1', 'multi_tracking_with_feature_l3'):
        return importlib.import_module('moog_demos.example_configs.multi_tracking_with_feature').get_config(int(name[-1]))
    if name == 'pacman_l1':   # level 1: three ghosts, 10 x 10 maze
        return importlib.import_module('moog_demos.example_configs.pacman').get_config(1)
    if name == 'chase_avoid_torus_l1':   # level 1: 1-2 prey and 1-2 predators (randint counts)
        return importlib.import_module('moog_demos.example_configs.chase_avoid_torus').get_config(1)
    level = 0
    m = re.match(r'(.*)_l(\d+)$', name)
    if m:
        name, level = m.group(1), int(m.group(2))
    pkg = 'amd_configs'
    if pkg not in sys.modules:
        spec = importlib.util.spec_from_file_location(
            pkg, os.path.join(CFG_DIR, '__init__.py'), submodule_search_locations=[CFG_DIR])
        mod = importlib.util.module_from_spec(spec)
        sys.modules[pkg] = mod
        spec.loader.exec_module(mod)
    return importlib.import_module(pkg + '.' + name).get_config(level)


def snapshot(env, layer_names, caps, slot_of):
    S = sum(caps)
    d = dict(
        alive=np.zeros(S, np.uint8), pos=np.full((S, 2), np.nan), vel=np.full((S, 2), np.nan),
        angle=np.full(S, np.nan), angvel=np.full(S, np.nan), mass=np.full(S, np.nan),
        color=np.full((S, 3), np.nan), opacity=np.zeros(S, np.int32),
        nverts=np.zeros(S, np.int32), verts=np.full((S, STATE_VMAX, 2), np.nan),
        inertia=np.full((S, 2), np.nan), maxr=np.full(S, np.nan),
        vel_f32=np.zeros(S, np.uint8), angvel_f32=np.zeros(S, np.uint8),
        sym_circle=np.zeros(S, np.uint8), tele=np.zeros(S, np.uint8),
        scale=np.full(S, np.nan), aspect=np.full(S, np.nan),
        fmask=np.zeros(S, np.int32),       # bit i: factor i (Sprite.FACTOR_NAMES order) is a float32 value
        vel_group=np.zeros(S, np.int32))   # 1 + lowest slot among sprites sharing ONE velocity ndarray
    owners = {}
    tele_ids = set()
    for r in getattr(env, 'game_rules', ()):
        tele_ids |= set(getattr(r, '_currently_teleporting', set()))
    offs = dict(zip(layer_names, np.concatenate([[0], np.cumsum(caps)[:-1]])))
    for name in layer_names:
        for i, s in enumerate(env.state[name]):
            # layers that rules append to are Python lists: slot = list position
            k = int(offs[name]) + i if name in DYNAMIC_LAYERS else slot_of[s.id]
            assert k < int(offs[name]) + caps[layer_names.index(name)], (name, i)
            d['alive'][k] = 1
            d['pos'][k] = s.position
            d['vel'][k] = s.velocity
            d['angle'][k] = s.angle
            d['angvel'][k] = s.angle_vel
            d['mass'][k] = s.mass
            d['color'][k] = s.color
            d['opacity'][k] = s.opacity
            v = s.vertices
            d['nverts'][k] = len(v)
            d['verts'][k, :len(v)] = v
            d['inertia'][k] = s._x_y_rotational_inertia
            d['maxr'][k] = s.max_radius
            d['vel_f32'][k] = np.asarray(s.velocity).dtype == np.float32
            d['angvel_f32'][k] = getattr(s.angle_vel, 'dtype', None) == np.float32
            d['sym_circle'][k] = bool(s.is_symmetric_circle)
            d['tele'][k] = s.id in tele_ids
            d['scale'][k] = s.scale
            d['aspect'][k] = s.aspect_ratio
            vals = (None, None, None, s.angle, s.scale, s.aspect_ratio, s.color[0], s.color[1], s.color[2],
                    None, None, None, None, s.mass)
            for bit, v in enumerate(vals):
                if getattr(v, 'dtype', None) == np.float32:
                    d['fmask'][k] |= 1 << bit
            owners.setdefault(id(s.velocity), []).append(k)
    for ks in owners.values():
        if len(ks) > 1:
            d['vel_group'][ks] = 1 + min(ks)
    # numbers / bools the config keeps in sprite.metadata (e.g. what an initializer's look-ahead found out,
    # bounce_box_contact_prediction.py:116, red_green.py:199): by sorted key, NaN where a sprite has none
    metas = {}
    for name in layer_names:
        for i, s in enumerate(env.state[name]):
            md = getattr(s, 'metadata', None)
            if isinstance(md, dict):
                k = int(offs[name]) + i if name in DYNAMIC_LAYERS else slot_of[s.id]
                for key, val in md.items():
                    if isinstance(val, (bool, int, float, np.integer, np.floating, np.bool_)) or val is None:
                        metas.setdefault(str(key), {})[k] = np.nan if val is None else float(val)
    if metas:
        keys = sorted(metas)
        d['meta_keys'] = np.array(keys)
        d['meta_vals'] = np.full((S, len(keys)), np.nan)
        for j, key in enumerate(keys):
            for k, val in metas[key].items():
                d['meta_vals'][k, j] = val
    return d


def action_memory(space):
    """_action of the space; for a Composite the sub-spaces' memories in keyword order
    (zeros for spaces without memory, e.g. SetPosition)."""
    subs = list(space.action_spaces.values()) if hasattr(space, 'action_spaces') else [space]
    return np.concatenate([np.array(getattr(sp, '_action', np.zeros(2)), dtype=float).reshape(2)
                           for sp in subs])


def bookkeeping(env):
    subtasks = getattr(env.task, '_tasks', (env.task,))
    tc = [float(getattr(t, '_steps_until_reset', np.nan)) for t in subtasks]
    rc = [float(getattr(r, '_steps_until_expire', getattr(r, '_steps_until_start', np.nan)))
          for r in env.game_rules]

    def state_of(r):   # the scalar the engine keeps per rule (include/moog_engine.h moog_rule_t)
        if hasattr(r, '_meta_state_fixation_key'):   # Fixation keeps its count in the meta-state (fixation.py:44-58)
            return float(env.meta_state[r._meta_state_fixation_key])
        if hasattr(r, '_current_phase_ind'):
            return float(r._current_phase_ind)
        if hasattr(r, '_should_end'):
            return -1.0 if r._should_end else float(r._step_count)
        return float(getattr(r, '_steps_until_expire', getattr(r, '_steps_until_start', np.nan)))

    def walk(rules, out):   # pre-order over the rule forest, children in stepping order
        for r in rules:
            out.append(state_of(r))
            kids = getattr(r, '_phases', None)
            if kids is None and hasattr(r, '_one_time_rules'):
                kids = list(r._one_time_rules) + list(r._continual_rules)
            if kids is None:
                kids = getattr(r, '_rules', None)
            if kids is not None:
                walk(list(kids), out)
        return out
    rc_flat = walk(list(env.game_rules), [])

    def second_of(r):
        # the duration a Phase drew when it was reset (task_phases.py:72); the width stop - start of a TimedRule (timing.py:47:
        # both count down together, so the difference is what stays fixed between resets)
        if hasattr(r, '_current_duration'):
            return float(r._current_duration)
        if hasattr(r, '_steps_until_stop') and hasattr(r, '_steps_until_start'):
            a, b = float(r._steps_until_start), float(r._steps_until_stop)
            return np.inf if np.isinf(b) and not np.isinf(a) else b - a
        return np.nan

    def walk2(rules, out):   # second scalar per rule
        for r in rules:
            out.append(second_of(r))
            kids = getattr(r, '_phases', None)
            if kids is None and hasattr(r, '_one_time_rules'):
                kids = list(r._one_time_rules) + list(r._continual_rules)
            if kids is None:
                kids = getattr(r, '_rules', None)
            if kids is not None:
                walk2(list(kids), out)
        return out
    rc2_flat = walk2(list(env.game_rules), [])
    # forces that keep state of their own: DeterministicMazeWalk pops its list front to back (maze_walk.py:238-239);
    # what the engine keeps is the number of elements consumed so far
    fstate = []
    for entry in getattr(env.physics, '_forces', ()):
        f = entry[0]
        if hasattr(f, '_step_velocities'):
            if not hasattr(f, '_n_initial'):
                f._n_initial = len(f._step_velocities)
            fstate.append(float(f._n_initial - len(f._step_velocities)))
        else:
            fstate.append(np.nan)
    books = dict(step_count=env.step_count, reset_next=int(env.reset_next_step),
                 action_mem=action_memory(env.action_space), force_state=np.array(fstate, dtype=float),
                 task_counters=np.array(tc, dtype=float), rule_counters=np.array(rc, dtype=float),
                 rule_counters_flat=np.array(rc_flat, dtype=float),
                 rule_counters2_flat=np.array(rc2_flat, dtype=float))
    # numbers the initializer's own object keeps across episodes (predators_arena.py:56: the curriculum's mass), by
    # attribute name in sorted order
    owner = getattr(getattr(env.state_initializer, 'wrapped', env.state_initializer), '__self__', None)
    keep = sorted(k for k, v in vars(owner).items() if isinstance(v, float)) if hasattr(owner, '__dict__') else []
    if keep:
        books['init_state'] = np.array([getattr(owner, k) for k in keep], dtype=float)
        books['init_state_names'] = np.array(keep)
    return books


def make_slot_map(env, layer_names, caps):
    slot_of, off = {}, 0
    for name, cap in zip(layer_names, caps):
        assert len(env.state[name]) <= cap, (name, len(env.state[name]), cap)
        for i, s in enumerate(env.state[name]):
            slot_of[s.id] = off + i
        off += cap
    return slot_of


def record_config(name, cfg, seed, n_calls, caps_by_layer, n_sub_steps=2):
    global TAPE, DYNAMIC_LAYERS, STATE_VMAX
    caps_by_layer = dict(caps_by_layer)
    DYNAMIC_LAYERS = tuple(caps_by_layer.pop('__dynamic__', ()))
    STATE_VMAX = caps_by_layer.pop('__vmax__', VMAX)
    action_bias = caps_by_layer.pop('__bias__', None)   # [n_spaces][2] drift of the random actions
    skip = caps_by_layer.pop('__skip__', 0)   # un-recorded steps after the reset: the recording starts from a later state
    sub_calls = tuple(caps_by_layer.pop('__sub_calls__', ()))   # further calls whose sub-step states are recorded
    script = caps_by_layer.pop('__script__', None)   # (env, call, RandomState) -> action: a policy instead of random actions
    action_f32 = bool(caps_by_layer.pop('__action_f32__', False))   # hand the reference float32 actions (joystick.py:42-43)
    TAPE = Tape(seed)
    act_rs = np.random.RandomState(1000 + seed)
    env = environment.Environment(**cfg)
    if not hasattr(env, 'game_rules') or env.game_rules is None:
        env.game_rules = ()
    is_grid = type(env.action_space).__name__ == 'Grid'
    space_kind = type(env.action_space).__name__

    def draw_action(space):
        """(the action handed to the reference, its packed [2] record)"""
        kind = type(space).__name__
        if kind == 'Grid':
            a = int(act_rs.randint(5))
            return a, np.array([float(a), 0.])
        lo = 0. if kind == 'SetPosition' else -1.
        a = act_rs.uniform(lo, 1., size=2)
        return a, a.copy()
    K = env.physics.updates_per_env_step

    sub_log, sub_log_calls = [], []
    real_apply = env.physics.apply_physics
    state_box = {}

    def apply_and_log(state, k):
        real_apply(state, k)
        if state_box.get('log') is not None:
            s = snapshot(env, state_box['layers'], state_box['caps'], state_box['slot_of'])
            state_box['log'].append({k_: s[k_] for k_ in ('pos', 'vel', 'angle', 'angvel', 'verts')})
    env.physics.apply_physics = apply_and_log

    # Slots are assigned when the state is created, i.e. before the rule.step of
    # Environment.reset (environment.py:92-94) can pop freshly generated sprites.
    real_init = env.state_initializer
    init_box = {}

    def init_and_map():
        state = real_init()
        names = list(state.keys())
        caps_ = [caps_by_layer.get(l, len(state[l])) for l in names]
        if 'caps' in init_box:
            caps_ = init_box['caps']
        off, m = 0, {}
        for l, cap in zip(names, caps_):
            assert len(state[l]) <= cap, (l, len(state[l]), cap)
            for i, s in enumerate(state[l]):
                m[s.id] = off + i
            off += cap
        init_box.update(names=names, caps=caps_, slot_of=m)
        return state
    init_and_map.wrapped = real_init   # (bookkeeping reads the numbers its object keeps across episodes)
    env.state_initializer = init_and_map

    ts = env.reset()
    layer_names, caps, slot_of = init_box['names'], init_box['caps'], init_box['slot_of']
    state_box.update(layers=layer_names, caps=caps, slot_of=slot_of, log=None)

    rows = []

    def push(ts, action, uniforms):
        row = snapshot(env, layer_names, caps, slot_of)
        row.update(bookkeeping(env))
        row['step_type'] = int(ts.step_type)
        row['reward'] = np.nan if ts.reward is None else float(ts.reward)
        row['discount'] = np.nan if ts.discount is None else float(ts.discount)
        row['image'] = np.asarray(ts.observation['image'])
        row['action'] = action
        row['uniforms'] = uniforms
        rows.append(row)

    zero_action = 4 if is_grid else np.zeros(2)
    if space_kind == 'Composite':
        zero_action = np.zeros((len(env.action_space.action_spaces), 2))
    for _ in range(skip):   # (plain Grid / Joystick spaces only; no reset may fall into the skipped part)
        assert not env.reset_next_step
        ts = env.step(int(act_rs.randint(5)) if is_grid else act_rs.uniform(-1., 1., size=2))
        TAPE.take()
    push(ts, zero_action, TAPE.take())
    for t in range(1, n_calls + 1):
        if space_kind == 'Composite':      # dict action, recorded as [n_spaces, 2]
            drawn = [(k, draw_action(sp)) for k, sp in env.action_space.action_spaces.items()]
            if action_bias is not None:   # steer the agents so that the rules have something to do
                drawn = [(k, (np.clip(d[0] + b, -1., 1.), np.clip(d[1] + b, -1., 1.)))
                         for (k, d), b in zip(drawn, np.asarray(action_bias, dtype=float))]
            ref_action = {k: d[0] for k, d in drawn}
            action = np.stack([d[1] for _, d in drawn])
        elif space_kind == 'SetPosition':
            ref_action, action = draw_action(env.action_space)
            if script is not None:
                ref_action = np.asarray(script(env, t, act_rs), dtype=float)
                action = ref_action.copy()
        else:
            action = int(act_rs.randint(5)) if is_grid else act_rs.uniform(-1., 1., size=2)
            if script is not None:   # a policy instead of random actions
                action = script(env, t, act_rs)
                action = int(action) if is_grid else np.asarray(action, dtype=float)
            ref_action = action if is_grid else np.array(action)
            if action_f32 and not is_grid:   # the recorded action is the float32 value (as float64)
                ref_action = np.array(action, dtype=np.float32)
                action = ref_action.astype(np.float64)
        will_reset = env.reset_next_step
        if (t <= n_sub_steps or t in sub_calls) and not will_reset:
            state_box['log'] = []
        ts = env.step(ref_action)
        if will_reset:
            slot_of = init_box['slot_of']
            state_box['slot_of'] = slot_of
        if state_box['log'] is not None:
            sub_log.append(state_box['log'])
            sub_log_calls.append(t)
            state_box['log'] = None
        push(ts, action, TAPE.take())

    out = {'layer_names': np.array(layer_names), 'layer_caps': np.array(caps, np.int32),
           'K': np.int32(K), 'is_grid': np.int32(is_grid), 'action_f32': np.int32(action_f32)}
    umax = max(1, max(len(r['uniforms']) for r in rows))
    U = np.full((len(rows), umax), np.nan)
    for i, r in enumerate(rows):
        U[i, :len(r['uniforms'])] = r['uniforms']
    out['uniforms'] = U
    out['n_uniforms'] = np.array([len(r['uniforms']) for r in rows], np.int32)
    for key in rows[0]:
        if key == 'uniforms':
            continue
        out[key] = np.stack([np.asarray(r[key]) for r in rows])
    out['sub_calls'] = np.array(sub_log_calls, np.int32)   # the calls whose sub-step states follow
    for key in ('pos', 'vel', 'angle', 'angvel', 'verts'):
        if sub_log:
            out['sub_' + key] = np.stack([np.stack([s[key] for s in step]) for step in sub_log])
    path = os.path.join(HERE, '%s_s%d.npz' % (name, seed))
    np.savez_compressed(path, **out)
    n_first = int((out['step_type'] == 0).sum())
    print('%-26s seed %d calls %d  resets %d  S %d  uniforms/call max %d  %.0f KB' % (
        name, seed, n_calls, n_first, sum(caps), umax, os.path.getsize(path) / 1024.))


def make_resize(seed=13):
    """Image.resize(size, resample=Image.LANCZOS) as PILRenderer uses it (pil_renderer.py:112) on canvases that
    look like rendered frames (flat background, opaque and translucent polygons) and on noise."""
    rs = np.random.RandomState(seed)
    cases = [((128, 128), 2), ((64, 64), 2), ((96, 48), 2), ((64, 64), 3), ((48, 80), 4), ((256, 256), 2), ((32, 32), 5)]
    out = {'n_cases': np.int32(len(cases))}
    for ci, ((ow, oh), aa) in enumerate(cases):
        cw, ch = aa * ow, aa * oh
        ins, outs = [], []
        for rep in range(4):
            if rep == 3:
                arr = rs.randint(0, 256, size=(ch, cw, 3)).astype(np.uint8)
                img = Image.fromarray(arr, 'RGB')
            else:
                img = Image.new('RGB', (cw, ch), tuple(int(v) for v in rs.randint(0, 256, size=3)))
                draw = ImageDraw.Draw(img, 'RGBA')
                for _ in range(12):
                    n = rs.randint(3, 9)
                    c = rs.uniform(0, 1, size=2) * [cw, ch]
                    r = rs.uniform(0.02, 0.3) * cw
                    ang = np.sort(rs.uniform(0, 2 * np.pi, size=n))
                    pts = [(float(c[0] + r * np.cos(a)), float(c[1] + r * np.sin(a))) for a in ang]
                    draw.polygon(pts, fill=tuple(int(v) for v in rs.randint(0, 256, size=3)) + (int(rs.choice([255, 255, 128, 60])),))
            ins.append(np.asarray(img).copy())
            outs.append(np.asarray(img.resize((ow, oh), resample=Image.LANCZOS)).copy())
        out['in_%d' % ci] = np.stack(ins)
        out['out_%d' % ci] = np.stack(outs)
    path = os.path.join(HERE, 'resize.npz')
    np.savez_compressed(path, **out)
    print('resize corpus: %d cases x 4 images  %.0f KB' % (len(cases), os.path.getsize(path) / 1024.))


# ---- predicate corpora --------------------------------------------------------------
def random_sprite(rs, shapes_pool):
    shape = shapes_pool[rs.randint(len(shapes_pool))]
    return ref_sprite.Sprite(
        x=rs.uniform(0.3, 0.7), y=rs.uniform(0.3, 0.7), shape=shape,
        angle=rs.uniform(0, 2 * np.pi), scale=rs.uniform(0.05, 0.25),
        aspect_ratio=rs.uniform(0.6, 1.4))


def make_predicates(n_pairs=3000, n_pts=12, seed=7):
    from moog import shapes as ref_shapes
    rs = np.random.RandomState(seed)
    pool = list(ref_shapes.SHAPES.keys()) + [
        1.8 * np.array([[-0.3, -0.3], [0.1, -0.7], [0.4, 0.6], [-0.1, 0.25]]),
        1.5 * np.array([[-0.5, -0.3], [-0.1, -0.7], [0.7, 0.1], [0., -0.1], [-0.3, 0.25]])]
    va = np.full((n_pairs, VMAX, 2), np.nan)
    vb = np.full((n_pairs, VMAX, 2), np.nan)
    na = np.zeros(n_pairs, np.int32)
    nb = np.zeros(n_pairs, np.int32)
    hit = np.zeros(n_pairs, np.uint8)
    pts = np.zeros((n_pairs, n_pts, 2))
    inside = np.zeros((n_pairs, n_pts), np.uint8)
    for i in range(n_pairs):
        a, b = random_sprite(rs, pool), random_sprite(rs, pool)
        if i % 3 == 0:  # near-touching pairs exercise the tolerance branches
            d = b.position - a.position
            d = d / (np.linalg.norm(d) + 1e-12)
            b.position = a.position + d * (a.max_radius + b.max_radius) * rs.uniform(0.3, 0.9)
        na[i], nb[i] = len(a.vertices), len(b.vertices)
        va[i, :na[i]], vb[i, :nb[i]] = a.vertices, b.vertices
        hit[i] = mpl_path.Path.intersects_path(a.path, b.path, filled=True)
        p = a.position + (rs.uniform(-1, 1, size=(n_pts, 2)) * a.max_radius)
        p[:min(n_pts, nb[i])] = b.vertices[:n_pts]  # vertices of the other sprite as probes
        pts[i] = p
        inside[i] = a._path.contains_points(p)
    path = os.path.join(HERE, 'predicates.npz')
    np.savez_compressed(path, va=va, na=na, vb=vb, nb=nb, hit=hit, pts=pts, inside=inside)
    print('predicates: %d pairs, %d hits, %d inside  %.0f KB' % (
        n_pairs, hit.sum(), inside.sum(), os.path.getsize(path) / 1024.))


def make_raster(n_poly=3000, seed=11):
    from moog import shapes as ref_shapes
    rs = np.random.RandomState(seed)
    pool = list(ref_shapes.SHAPES.keys()) + [
        1.8 * np.array([[-0.3, -0.3], [0.1, -0.7], [0.4, 0.6], [-0.1, 0.25]]),
        1.5 * np.array([[-0.5, -0.3], [-0.1, -0.7], [0.7, 0.1], [0., -0.1], [-0.3, 0.25]]),
        np.array([[0., 0.05], [1., 0.05], [1., -0.45], [0., -0.45]])]
    xy = np.zeros((n_poly, VMAX, 2), np.int32)
    nv = np.zeros(n_poly, np.int32)
    size = np.zeros(n_poly, np.int32)
    cover = []
    gray = []
    for i in range(n_poly):
        W = 64 if i % 4 else 128
        shape = pool[rs.randint(len(pool))]
        s = ref_sprite.Sprite(
            x=rs.uniform(-0.3, 1.3), y=rs.uniform(-0.3, 1.3), shape=shape,
            angle=rs.uniform(0, 2 * np.pi) if i % 5 else 0.,
            scale=rs.uniform(0.02, 0.6), aspect_ratio=rs.uniform(0.4, 1.6))
        v = s.vertices * W
        kind = i % 10
        if kind == 7:      # comb: > 8 crossings per scanline (generic scanline path)
            teeth = rs.randint(4, 8)
            x0, y0 = rs.uniform(-0.1, 0.5) * W, rs.uniform(-0.1, 0.5) * W
            tw, th = rs.uniform(2, 7), rs.uniform(5, 40)
            pts_ = [(x0, y0)]
            for t in range(teeth):
                xa = x0 + (2 * t) * tw
                pts_ += [(xa + rs.uniform(0, 1), y0 + th + rs.uniform(-2, 2)), (xa + tw, y0 + th * rs.uniform(0.2, 0.6)),
                         (xa + 2 * tw, y0 + th + rs.uniform(-2, 2))][:3]
            pts_.append((x0 + 2 * teeth * tw, y0))
            v = np.array(pts_[:VMAX])
            if i % 20 == 7:
                ang = rs.uniform(0, 2 * np.pi)
                R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
                v = (v - v.mean(0)) @ R.T + v.mean(0)
        elif kind == 8:    # fan of same-lean spikes sharing tips (repeated corner fix-ups)
            cx, cy = rs.uniform(0.2, 0.8) * W, rs.uniform(0.2, 0.8) * W
            n_sp = rs.randint(2, 5)
            pts_ = []
            for t in range(n_sp):
                pts_ += [(cx + rs.randint(-3, 4), cy + rs.randint(-2, 3)),
                         (cx + rs.uniform(5, 30), cy + rs.uniform(5, 30) * (1 if t % 2 else -1)),
                         (cx + rs.uniform(8, 40), cy + rs.uniform(2, 20) * (1 if t % 2 else -1))]
            v = np.array(pts_)
        elif kind == 9:    # tiny sprites: coincident integer vertices, merged horizontal runs
            s2 = ref_sprite.Sprite(x=rs.uniform(0.05, 0.95), y=rs.uniform(0.05, 0.95), shape=shape,
                                   angle=rs.uniform(0, 2 * np.pi), scale=rs.uniform(0.01, 0.06),
                                   aspect_ratio=rs.uniform(0.5, 1.5))
            v = s2.vertices * W
        pts = [tuple(p) for p in v]
        canvas = Image.new('RGB', (W, W), (10, 20, 30))
        draw = ImageDraw.Draw(canvas, 'RGBA')
        draw.polygon(pts, fill=(200, 100, 50, 128))
        img = np.array(canvas)
        iv = np.array([[int(px), int(py)] for px, py in pts], np.int32)  # C truncation
        nv[i] = len(iv)
        xy[i, :nv[i]] = iv
        size[i] = W
        full = np.zeros((128, 128), np.uint8)
        full[:W, :W] = img[:, :, 0]
        gray.append(full)
    gray = np.stack(gray)
    path = os.path.join(HERE, 'raster.npz')
    np.savez_compressed(path, xy=xy, nv=nv, size=size, red=gray,
                        bg=np.array([10, 20, 30]), ink=np.array([200, 100, 50, 128]))
    print('raster: %d polygons  %.0f KB' % (n_poly, os.path.getsize(path) / 1024.))


def make_collision_kat():
    """Runs the scenarios of tests/moog/physics/test_collisions.py:101-293 through
    the reference and stores the exact float64 outcomes (the test file itself
    only pins them to 1e-3)."""
    def pairwise(sprites, force, symmetric):
        n = len(sprites)
        inds = ([(i, j) for j in range(n) for i in range(n)] if symmetric
                else [(i, j) for i in range(n) for j in range(i)])
        for i, j in inds:
            force.step(sprites[i], sprites[j], updates_per_env_step=1)

    circ_same = [
        ([0.5, 0.35], [0., 0.], 1., False), ([0.5, 0.35], [0., 0.], 1., True),
        ([0.5, 0.35], [0., 0.], 0.5, True), ([0.5, 0.35], [0., 0.], 0., True),
        ([0.5, 0.35], [0., 0.01], 1., True), ([0.44, 0.37], [0., 0.], 1., False),
        ([0.44, 0.37], [0., 0.], 1., True), ([0.44, 0.37], [0., 0.], 0.5, True),
        ([0.44, 0.37], [0., 0.01], 1., True), ([0.43, 0.36], [0.015, 0.01], 1., True)]
    circ_diff = [([0.5, 0.35], [0., 0.]), ([0.5, 0.35], [0., 0.]), ([0.44, 0.37], [0., 0.01]),
                 ([0.43, 0.36], [0.015, 0.01])]
    tri = [(0., 1., False), (0., 1., True), (0., 0.5, True), (0.1, 1., True), (-0.02, 1., True)]
    rows = []
    for pos0, vel0, el, sym in circ_same:
        f = ref_collisions.Collision(elasticity=el, symmetric=sym, update_angle_vel=False)
        s0 = ref_sprite.Sprite(x=pos0[0], y=pos0[1], scale=0.1, shape='circle',
                               x_vel=vel0[0], y_vel=vel0[1], c1=255)
        s1 = ref_sprite.Sprite(x=0.5, y=0.5, scale=0.1, shape='circle', y_vel=-0.01, c0=255)
        for _ in range(6):
            pairwise([s0, s1], f, sym)
            s0.update_pos_from_vel(delta_t=1.)
            s1.update_pos_from_vel(delta_t=1.)
        rows.append(np.concatenate([s0.position, s0.velocity, [s0.angle_vel],
                                    s1.position, s1.velocity, [s1.angle_vel]]))
    for pos0, vel0 in circ_diff:
        f = ref_collisions.Collision(elasticity=1., symmetric=True, update_angle_vel=False)
        s0 = ref_sprite.Sprite(x=pos0[0], y=pos0[1], scale=0.1, shape='circle',
                               x_vel=vel0[0], y_vel=vel0[1], c1=255)
        s1 = ref_sprite.Sprite(x=0.5, y=0.5, scale=0.1, shape='circle', y_vel=-0.01, c0=255,
                               mass=2.)
        for _ in range(6):
            pairwise([s0, s1], f, True)
            s0.update_pos_from_vel(delta_t=1.)
            s1.update_pos_from_vel(delta_t=1.)
        rows.append(np.concatenate([s0.position, s0.velocity, [s0.angle_vel],
                                    s1.position, s1.velocity, [s1.angle_vel]]))
    for av0, el, upd in tri:
        f = ref_collisions.Collision(elasticity=el, symmetric=True, update_angle_vel=upd)
        s0 = ref_sprite.Sprite(x=0.5, y=0, scale=0.05, shape=np.array([[1, 1], [1, 3], [-2, -2]]),
                               x_vel=0.005, y_vel=0., c0=255, angle=1., angle_vel=av0)
        s1 = ref_sprite.Sprite(x=0.31, y=0.88, scale=0.05,
                               shape=np.array([[2, 1], [0, 1], [-1, -3]]), x_vel=-0.005, y_vel=0.,
                               c1=255)
        for _ in range(10):
            pairwise([s0, s1], f, True)
            s0.update_pos_from_vel(delta_t=1.)
            s1.update_pos_from_vel(delta_t=1.)
        rows.append(np.concatenate([s0.position, s0.velocity, [s0.angle_vel],
                                    s1.position, s1.velocity, [s1.angle_vel]]))
    path = os.path.join(HERE, 'collisions_kat.npz')
    np.savez_compressed(path, final=np.stack(rows))
    print('collisions_kat: %d scenarios' % len(rows))


def make_logger_fixture():
    """A run of the reference's LoggingEnvironment (env_wrappers/logger.py) on a config without
    randomness (tether_zoo level 0): the episode files it writes, as one JSON fixture."""
    import json
    import shutil
    import tempfile
    from moog.env_wrappers import logger as ref_logger
    cfg = load_amd_config('tether_zoo_l0')
    tmp = tempfile.mkdtemp()
    env = ref_logger.LoggingEnvironment(environment.Environment(**cfg), log_dir=tmp)
    rs = np.random.RandomState(3)
    actions = rs.uniform(-1., 1., size=(45, 2))
    env.reset()
    for a in actions:
        env.step(a)
    run_dir = env._log_dir
    episodes = []
    for fn in sorted(os.listdir(run_dir)):
        if fn.isdigit():
            with open(os.path.join(run_dir, fn)) as f:
                episodes.append(json.load(f))
    with open(os.path.join(run_dir, 'attributes.txt')) as f:
        attributes = json.load(f)
    with open(os.path.join(run_dir, 'description.txt')) as f:
        description = f.read()
    shutil.rmtree(tmp)
    path = os.path.join(HERE, 'logger_tether_zoo_l0.json')
    with open(path, 'w') as f:
        json.dump({'attributes': attributes, 'description': description, 'actions': actions.tolist(),
                   'episodes': episodes}, f)
    print('logger fixture: %d episodes, %d steps  %.0f KB' % (
        len(episodes), sum(len(e) for e in episodes), os.path.getsize(path) / 1024.))


def main():
    if sys.argv[1:] == ['logger']:
        make_logger_fixture()
        return
    only = sys.argv[1:]   # config names, or name:seed; the corpora are made by a run without arguments
    if only == ['resize']:
        make_resize()
        return
    if not only:
        make_resize()
        make_collision_kat()   # before np.random is patched (uses no randomness anyway)
        make_predicates()
        make_raster()
    patch_numpy_random()
    plan = [
        ('pong', 96, {}, (0, 1)),
        ('chase_avoid_torus', 64, {}, (0, 1)),
        ('colliding_predators', 64, {}, (0, 1)),
        ('colliding_predators', 48, {'__action_f32__': True}, (2,)),   # float32 actions: scaling_factor * action in float32
        ('chase_avoid_torus', 48, {'__action_f32__': True}, (2,)),
        ('functional_maze', 96, {'prey': 4}, (0, 1)),
        ('falling_balls', 48, {}, (0,)),
        ('colliding_predators_32', 40, {}, (0,)),
        ('falling_balls_64', 12, {}, (0,)),
        ('falling_balls_64', 64, {'__skip__': 42, '__sub_calls__': (22, 23)}, (1,)),   # from the piled-up state of step 42, across the timeout at step 100
        ('forces_zoo', 96, {}, (0, 1)),
        ('chase_avoid_torus_l1', 48, {'prey': 2, 'predators': 2}, (0,)),
        ('tether_zoo_l0', 45, {}, (0,)),
        ('tether_zoo_l1', 45, {}, (0,)),
        ('tether_zoo_l2', 45, {}, (0,)),
        ('tether_zoo_l3', 45, {}, (0,)),
        ('tether_zoo_l4', 45, {}, (0,)),
        ('distrib_zoo', 60, {}, (0, 1)),
        ('lambda_zoo', 90, {'bin': 8, '__dynamic__': ('bin',)}, (0, 1)),
        ('cond_zoo', 120, {'extras': 8, '__dynamic__': ('extras',)}, (0, 1)),
        ('phase_zoo', 120, {}, (0, 1)),
        ('phase_zoo_l1', 120, {}, (0, 1)),
        ('actions_zoo', 60, {}, (0, 1)),
        ('actions_zoo_l1', 40, {}, (0,)),
        ('cleanup', 150, {'__bias__': [[0., -0.7], [0., 0.7], [0.3, -0.5]]}, (0, 1)),
        ('rules_zoo_l0', 40, {'bin': 8, '__dynamic__': ('bin',)}, (0,)),
        ('rules_zoo_l1', 80, {'prey': 8, 'predators': 8, '__dynamic__': ('prey', 'predators')}, (0, 1)),
        ('first_person_predators_prey', 70, {'prey': 16, 'predators': 40, '__vmax__': SNAP_VMAX,
                                             '__dynamic__': ('prey', 'predators')}, (0,)),
        ('rules_zoo_l2', 90, {'prey': 8, '__dynamic__': ('prey',)}, (0,)),
        ('maze_zoo', 120, {}, (0, 1)),
        ('maze_zoo_l1', 120, {}, (0,)),
        ('maze_zoo_l2', 160, {}, (0, 1)),
        ('pacman', 150, {'walls': 136, 'prey': 48}, (0, 1)),   # the per-episode random maze: walls + prey = 144 cells
        ('pacman_l1', 100, {'walls': 136, 'prey': 75}, (0,)),
        ('sampler_zoo', 60, {'blocks': 8}, (0, 1)),
        ('sampler_zoo_l1', 70, {'blocks': 8, '__dynamic__': ('blocks',)}, (0,)),
        ('sampler_zoo_l2', 50, {'blocks': 8}, (0, 1)),
        ('sampler_zoo_l3', 60, {'blocks': 8}, (0, 1)),
        ('parallelogram_catch', 60, {'__vmax__': SNAP_VMAX}, (0, 1)),
        ('parallelogram_catch_l1', 60, {'__vmax__': SNAP_VMAX}, (0, 1)),
        ('parallelogram_catch_l2', 60, {'__vmax__': SNAP_VMAX}, (0,)),
        ('multi_tracking_with_feature_l3', 230, {'__vmax__': SNAP_VMAX, '__script__': track_and_fixate}, (0, 1)),
        ('multi_tracking_with_feature_l1', 230, {'__vmax__': SNAP_VMAX, '__script__': track_and_fixate}, (0,)),
        ('match_to_sample_l3', 175, {'__vmax__': SNAP_VMAX, '__script__': seek_match}, (0,)),
        ('match_to_sample_l3', 175, {'__vmax__': SNAP_VMAX, '__script__': seek_other}, (1,)),
        ('match_to_sample_l4', 175, {'__vmax__': SNAP_VMAX, '__script__': seek_match}, (0,)),
        ('match_to_sample_l2', 175, {'__vmax__': SNAP_VMAX, '__script__': seek_other}, (0,)),
        ('tracing_zoo', 130, {'__script__': lambda env, t, rs: np.array([rs.uniform(-0.4, 0.4), rs.uniform(0.2, 1.)])}, (0, 1)),
        ('tracing_zoo_l1', 150, {}, (0, 1)),
        ('combo_zoo', 130, {'__action_f32__': True}, (0, 1)),
        ('lookahead_zoo', 140, {'__script__': _go_to}, (0, 1)),
        ('lookahead_zoo_l1', 140, {'__script__': _go_to}, (0, 1)),
        ('red_green_l1', 110, {'__script__': _answer([1, 0, 0, 1], (4, 5))}, (0,)),   # (right = red, left = green)
        ('red_green', 80, {'__script__': _answer([0, 1, 1, 0], (4, 5))}, (0,)),
        ('red_green_l3', 80, {'__script__': _answer([1, 1, 0, 0], (4, 5))}, (0,)),
        ('bounce_box_contact_prediction', 110, {'__script__': _answer([0, 1, 1, 0])}, (0,)),
        ('bounce_box_contact_prediction_l1', 110, {'__script__': _answer([1, 0, 0, 1])}, (0,)),
        ('predators_arena_l2', 260, {}, (0,)),   # ten resets: the curriculum's mass after every one of them
        ('predators_arena_l2', 120, {}, (1,)),
        ('predators_arena_l1', 120, {}, (0,)),
        ('predators_arena_l3', 100, {}, (0,)),
        ('dependent_zoo', 50, {}, (0, 1)),
        ('aa_zoo', 30, {}, (0,)),
        ('aa_zoo_l1', 30, {}, (0,)),
        ('aa_zoo_l2', 12, {}, (0,)),
        ('aa_zoo_l3', 30, {}, (0,)),
        ('aa_zoo_l4', 30, {}, (0,)),
        ('aa_zoo_l5', 30, {}, (0,)),
        ('callables_zoo', 70, {}, (0, 1)),      # round 4: traced force_fn, callable rule intervals, ContactReward with meta_state
        ('callables_zoo_l1', 50, {}, (0,)),
        ('callables_zoo_l2', 40, {}, (0,)),     # PILRenderer(color_to_rgb=<a Python function>)
        ('callables_zoo_l3', 60, {}, (0, 1)),   # round 6: DelayedRule with a callable start AND a callable duration (two draws per reset)
    ]
    for name, n_calls, caps, seeds in plan:
        for seed in seeds:
            if only and name not in only and '%s:%d' % (name, seed) not in only:
                continue
            record_config(name, load_amd_config(name), seed, n_calls, caps)


if __name__ == '__main__':
    main()
