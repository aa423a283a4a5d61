"""HIP engine vs the CPU oracle / the reference golden vectors (needs an MI355X).
Everything goes through the C ABI (moog.environment.BatchedEnvironment -> ctypes)."""
import os

import numpy as np
import pytest

import helpers
from helpers import compiled, fixture, records_from_fixture, state_diff

pytestmark = pytest.mark.gpu
TOL = 1e-5
RUNS = [('pong', 0), ('pong', 1), ('chase_avoid_torus', 0), ('chase_avoid_torus', 1),
        ('colliding_predators', 0), ('colliding_predators', 1), ('colliding_predators', 2), ('chase_avoid_torus', 2), ('functional_maze', 0),
        ('functional_maze', 1), ('falling_balls', 0), ('colliding_predators_32', 0),
        ('falling_balls_64', 0), ('falling_balls_64', 1), ('forces_zoo', 0), ('forces_zoo', 1), ('chase_avoid_torus_l1', 0),
        ('tether_zoo_l0', 0), ('tether_zoo_l1', 0), ('tether_zoo_l2', 0), ('tether_zoo_l3', 0),
        ('tether_zoo_l4', 0), ('distrib_zoo', 0), ('distrib_zoo', 1),
        ('rules_zoo_l0', 0), ('rules_zoo_l1', 0), ('rules_zoo_l1', 1),
        ('lambda_zoo', 0), ('lambda_zoo', 1), ('rules_zoo_l2', 0),
        ('first_person_predators_prey', 0), ('cond_zoo', 0), ('cond_zoo', 1), ('phase_zoo', 0), ('phase_zoo', 1), ('phase_zoo_l1', 0), ('phase_zoo_l1', 1), ('match_to_sample_l3', 0), ('match_to_sample_l3', 1), ('match_to_sample_l4', 0), ('match_to_sample_l2', 0), ('predators_arena_l2', 0), ('predators_arena_l2', 1), ('predators_arena_l1', 0), ('predators_arena_l3', 0), ('bounce_box_contact_prediction', 0), ('bounce_box_contact_prediction_l1', 0), ('red_green_l1', 0), ('red_green', 0), ('red_green_l3', 0), ('lookahead_zoo', 0), ('lookahead_zoo', 1), ('lookahead_zoo_l1', 0), ('lookahead_zoo_l1', 1), ('tracing_zoo', 0), ('tracing_zoo', 1), ('tracing_zoo_l1', 0), ('tracing_zoo_l1', 1), ('combo_zoo', 0), ('combo_zoo', 1),
        ('actions_zoo', 0), ('actions_zoo', 1), ('actions_zoo_l1', 0), ('cleanup', 0), ('cleanup', 1),
        ('aa_zoo', 0), ('aa_zoo_l1', 0), ('aa_zoo_l2', 0), ('aa_zoo_l3', 0), ('aa_zoo_l4', 0), ('aa_zoo_l5', 0), ('callables_zoo', 0), ('callables_zoo', 1), ('callables_zoo_l1', 0), ('callables_zoo_l2', 0), ('callables_zoo_l3', 0), ('callables_zoo_l3', 1), ('maze_zoo', 0), ('maze_zoo', 1), ('maze_zoo_l1', 0), ('maze_zoo_l2', 0), ('maze_zoo_l2', 1),
        ('pacman', 0), ('pacman', 1), ('pacman_l1', 0),
        ('sampler_zoo', 0), ('sampler_zoo', 1), ('sampler_zoo_l1', 0),
        ('parallelogram_catch', 0), ('parallelogram_catch', 1), ('parallelogram_catch_l1', 0), ('parallelogram_catch_l1', 1),
        ('parallelogram_catch_l2', 0), ('multi_tracking_with_feature_l3', 0), ('multi_tracking_with_feature_l3', 1),
        ('multi_tracking_with_feature_l1', 0), ('dependent_zoo', 0), ('dependent_zoo', 1),
        ('sampler_zoo_l2', 0), ('sampler_zoo_l2', 1), ('sampler_zoo_l3', 0), ('sampler_zoo_l3', 1)]


def make_env(name, n, seed=0, **kw):
    import torch  # noqa: F401
    from moog import environment
    from moog_demos import example_configs
    kw.setdefault('layer_capacity', example_configs.capacity(name))
    return environment.BatchedEnvironment(num_envs=n, seed=seed, **example_configs.load(name), **kw)


def upload(env, f64, i32):
    import torch
    env.state_f64.copy_(torch.from_numpy(f64))
    env.state_i32.copy_(torch.from_numpy(i32))


def download(env):
    import torch
    torch.cuda.synchronize()
    return env.state_f64.cpu().numpy(), env.state_i32.cpu().numpy()


def padded_uniforms(fx, ts):
    width = max(1, int(fx['uniforms'].shape[1]))
    u = np.zeros((len(ts), width))
    for i, t in enumerate(ts):
        n = int(fx['n_uniforms'][t])
        u[i, :n] = fx['uniforms'][t, :n]
    return u


@pytest.mark.parametrize('name,seed', RUNS)
def test_teacher_forced_vs_reference(name, seed):
    """All recorded calls at once: env i starts from the reference state of call
    i and must land on the reference state of call i+1 (floats <= 1e-5, ints,
    rewards, step types and frames exact)."""
    c, fx = compiled(name), fixture(name, seed)
    T = len(fx['step_type'])
    ts = list(range(1, T))
    env = make_env(name, len(ts))
    L = c.layout
    f64 = np.zeros((len(ts), L.f64_per_env))
    i32 = np.zeros((len(ts), L.i32_per_env), np.int32)
    for i, t in enumerate(ts):
        records_from_fixture(fx, t - 1, c, f64, i32, env=i)
    upload(env, f64, i32)
    actions = np.stack([helpers.action_of(fx, t) for t in ts])
    env.check_faults = False
    out = env.step(actions, injected_uniforms=padded_uniforms(fx, ts))
    f, q = download(env)
    img = out.observation['image'].cpu().numpy()
    worst = 0.0
    for i, t in enumerate(ts):   # (every call, the two knife-edge calls of the pile-up recording included)
        d = state_diff(fx, t, c, f, q, env=i)
        assert d['ints_ok'], (t, d)
        assert d['float'] <= TOL, (t, d)
        worst = max(worst, d['float'])
        assert int(out.step_type[i]) == int(fx['step_type'][t]), t
        assert helpers.same_or_nan(float(out.reward[i]), fx['reward'][t]), t
        assert helpers.same_or_nan(float(out.discount[i]), fx['discount'][t]), t
        assert np.array_equal(img[i], fx['image'][t]), 'frame %d differs' % t
        assert int(q[i, L.o_fault]) == 0
    print(name, seed, 'worst teacher-forced error vs reference', worst)


@pytest.mark.parametrize('name,seed', RUNS)
def test_free_running_vs_reference(name, seed):
    """One env free-running from the first reference state for <= 64 calls."""
    c, fx = compiled(name), fixture(name, seed)
    env = make_env(name, 1)
    import test_oracle_golden as tog
    T = min([len(fx['step_type']), 65, tog.FREE_WINDOW.get((name, seed), 65)])
    f64, i32 = records_from_fixture(fx, 0, c)
    upload(env, f64, i32)
    env.check_faults = False
    for t in range(1, T):
        a = helpers.action_of(fx, t)
        a = a.reshape((1,) + a.shape) if a.ndim == 2 else a.reshape((1, 2) if not fx['is_grid'] else (1,))
        out = env.step(a, injected_uniforms=padded_uniforms(fx, [t]))
        f, q = download(env)
        d = state_diff(fx, t, c, f, q)
        assert d['ints_ok'], (t, d)
        assert d['float'] <= TOL, (t, d)
        assert int(out.step_type[0]) == int(fx['step_type'][t])
        # (a reward that is a function of the float state -- Reset(reward_fn=lambda state: 10 * x + 1), callables_zoo --
        #  drifts with it: inside the same budget as the state; every other reward is exact)
        r, want = float(out.reward[0]), float(fx['reward'][t])
        assert helpers.same_or_nan(r, want) or abs(r - want) <= TOL * max(1.0, abs(want)), (t, r, want)
    assert np.array_equal(out.observation['image'][0].cpu().numpy(), fx['image'][T - 1])


@pytest.mark.parametrize('name,seed', [r for r in RUNS if r != ('falling_balls_64', 1)])   # (that recording starts at step 42)
def test_reset_sampler_vs_reference(name, seed):
    """Device-side state initialisation replaying the reference's recorded draws."""
    c, fx = compiled(name), fixture(name, seed)
    env = make_env(name, 1)
    out = env.reset(injected_uniforms=padded_uniforms(fx, [0]))
    f, q = download(env)
    d = state_diff(fx, 0, c, f, q)
    assert d['ints_ok'] and d['float'] <= TOL, d
    assert np.array_equal(out.observation['image'][0].cpu().numpy(), fx['image'][0])
    assert int(out.step_type[0]) == 0 and np.isnan(float(out.reward[0]))


@pytest.mark.parametrize('name', ['pong', 'chase_avoid_torus', 'colliding_predators',
                                  'functional_maze', 'falling_balls', 'colliding_predators_32',
                                  'forces_zoo', 'tether_zoo_l1', 'tether_zoo_l3', 'tether_zoo_l4',
                                  'distrib_zoo', 'rules_zoo_l0', 'rules_zoo_l1', 'rules_zoo_l2',
                                  'lambda_zoo', 'first_person_predators_prey', 'maze_zoo', 'maze_zoo_l1', 'maze_zoo_l2',
                                  'pacman', 'pacman_l1', 'match_to_sample_l3', 'match_to_sample_l4', 'predators_arena_l2',
                                  'bounce_box_contact_prediction', 'red_green_l1', 'lookahead_zoo', 'lookahead_zoo_l1', 'tracing_zoo', 'tracing_zoo_l1', 'combo_zoo'])
def test_engine_vs_oracle_own_rng(name):
    """Same Philox streams on both sides, 64 envs, resets included: integer
    records bit-exact, floats <= 1e-9, frames bit-exact from the engine state."""
    import torch
    n, steps = 64, 24
    # (the recipes' own LAYER_CAPACITY values are the ones the reference fixtures were recorded with: one env; 64 envs of
    #  random play ask for more room in the layers rules append to)
    kw = {'layer_capacity': {'prey': 24, 'predators': 24}} if name == 'rules_zoo_l1' else {}
    env = make_env(name, n, seed=11, env_index0=1000, **kw)
    o = helpers.OracleEnv(env.compiled, n_envs=n, seed=11, env_index0=1000)
    env.reset()
    o.reset(render=False)
    rs = np.random.RandomState(5)
    grid = env._is_grid
    for k in range(steps):
        a = rs.randint(0, 5, size=n) if grid else rs.uniform(-1, 1, size=(n, 2))
        out = env.step(a)
        o.step(a, render=False)
        f, q = download(env)
        assert np.array_equal(q, o.i32), 'int state differs at step %d' % k
        with np.errstate(invalid='ignore'):
            err = np.abs(f - o.f64)
        err = np.where(np.isnan(f) & np.isnan(o.f64), 0, err)
        err = np.where(f == o.f64, 0, err)
        assert float(np.max(err)) <= 1e-9, (k, float(np.max(err)))
        assert np.array_equal(out.step_type.cpu().numpy(), o.step_type)
        assert helpers.same_or_nan(out.reward.cpu().numpy(), o.reward)
        assert helpers.same_or_nan(out.discount.cpu().numpy(), o.discount)
        # keep the two in lock step (removes 1-ulp libm/ocml cos/sin drift)
        o.f64[:], o.i32[:] = f, q
    assert np.array_equal(out.observation['image'].cpu().numpy(), o.render())


@pytest.mark.gpu
@pytest.mark.parametrize('name,rows', [('colliding_predators_32', None), ('colliding_predators_32', 64),
                                       ('chase_avoid_torus', None), ('chase_avoid_torus', 64),
                                       ('functional_maze', 64), ('falling_balls_64', None),
                                       ('first_person_predators_prey', 128)])
def test_frames_many_states(name, rows, monkeypatch):
    """Frames of 256 envs over 12 steps against the oracle renderer, every step; with `rows` the
    rasteriser's row records are capped (MOOG_RASTER_ROWS) so that a frame takes several passes."""
    if rows is not None:
        monkeypatch.setenv('MOOG_RASTER_ROWS', str(rows))
    n = 256
    env = make_env(name, n, seed=21, env_index0=300)
    o = helpers.OracleEnv(env.compiled, n_envs=n, seed=21, env_index0=300)
    env.reset()
    rs = np.random.RandomState(8)
    for k in range(12):
        a = rs.randint(0, 5, size=n) if env._is_grid else rs.uniform(-1, 1, size=(n, 2))
        out = env.step(a)
        o.f64[:], o.i32[:] = download(env)
        img = out.observation['image'].cpu().numpy()
        ref = o.render()
        bad = np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0]
        assert bad.size == 0, ('frames differ at step %d' % k, bad[:8].tolist(), int(bad.size))


@pytest.mark.parametrize('name', ['colliding_predators_32', 'pong', 'functional_maze'])
def test_static_prefix_cache_and_fallback(name, monkeypatch):
    """The rasteriser composes on top of a cached picture of the leading constant sprites (the walls)
    when a frame's prefix equals the reference record bit for bit.  Frames whose prefix was changed
    (a wall moved by one ulp / by pixels, recoloured, made translucent or removed) must take the
    ordinary path, next to untouched frames that use the cache: all compared with the oracle
    renderer, and the whole batch with the cache disabled gives the same pictures."""
    n = 96
    env = make_env(name, n, seed=5, env_index0=40)
    o = helpers.OracleEnv(env.compiled, n_envs=n, seed=5, env_index0=40)
    env.reset()
    rs = np.random.RandomState(3)
    for _ in range(3):
        env.step(rs.randint(0, 5, size=n) if env._is_grid else rs.uniform(-1, 1, size=(n, 2)))
    f, q = download(env)
    L, P = env.layout, env.compiled.program
    for i in range(0, n, 2):   # every other env keeps its prefix
        kind = (i // 2) % 6
        s = (i // 12) % 4      # which of the leading sprites
        v0 = L.o_verts + 2 * P.slot_voff[s]
        if kind == 0:
            f[i, v0] = np.nextafter(f[i, v0], 2.0)
        elif kind == 1:
            f[i, v0:v0 + 2 * int(q[i, L.o_nverts + s]):2] += 0.11
        elif kind == 2:
            f[i, L.o_color + 3 * s + 2] = 0.9 if P.render.cmap else 200
        elif kind == 3:
            q[i, L.o_opacity + s] = 100
        elif kind == 4:
            q[i, L.o_flags + s] &= ~1
        else:
            q[i, L.o_nverts + s] -= 1
    upload(env, f, q)
    o.f64[:], o.i32[:] = f, q
    img = env.observation()['image'].cpu().numpy()
    ref = o.render()
    bad = np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0]
    assert bad.size == 0, ('frames differ', bad[:8].tolist(), int(bad.size))
    monkeypatch.setenv('MOOG_RASTER_NO_STATIC', '1')
    env2 = make_env(name, n, seed=5, env_index0=40)
    upload(env2, f, q)
    assert np.array_equal(env2.observation()['image'].cpu().numpy(), img)


@pytest.mark.parametrize('size', [(64, 64), (50, 37), (150, 41)])
def test_static_prefix_picture(size):
    """moog_engine_static_prefix hands out the cached picture of the leading constant sprites in frame layout
    (uint8[H, W, 3], also when the rasteriser drew it 16-aligned): equal to the oracle's frame of a state in which
    every other sprite is gone."""
    from moog import environment, observers
    from moog_demos import example_configs
    cfg = example_configs.load('colliding_predators_32')
    old = cfg['observers']['image']
    cfg['observers'] = {'image': observers.PILRenderer(image_size=size, bg_color=old._bg_color, color_to_rgb=old.color_to_rgb)}
    env = environment.BatchedEnvironment(num_envs=4, seed=5, **cfg)
    o = helpers.OracleEnv(env.compiled, n_envs=4, seed=5)
    env.reset()
    ns, pic = env.static_prefix()
    assert ns == 4 and tuple(pic.shape) == (size[1], size[0], 3)
    f, q = download(env)
    L = env.layout
    q[:, L.o_flags + ns:L.o_flags + L.S] &= ~1
    o.f64[:], o.i32[:] = f, q
    assert np.array_equal(pic.cpu().numpy(), o.render()[0])
    env.close()


def test_philox_bit_exact():
    """The device RNG stream equals the oracle's: a reset driven by it gives the
    same integer records (shape ids, counts, rng counters)."""
    env = make_env('functional_maze', 32, seed=99, env_index0=7)
    o = helpers.OracleEnv(env.compiled, n_envs=32, seed=99, env_index0=7)
    env.reset()
    o.reset(render=False)
    f, q = download(env)
    assert np.array_equal(q, o.i32)
    with np.errstate(invalid='ignore'):
        err = np.where(f == o.f64, 0.0, np.abs(f - o.f64))
    assert float(np.max(err)) <= 1e-12


def test_raster_corpus_vs_pillow():
    """The 3000-polygon Pillow corpus through the HIP rasteriser (one polygon per env)."""
    import collections
    import torch
    from moog import environment, action_spaces, observers, physics as physics_lib, sprite, tasks
    z = dict(np.load(helpers.GOLDEN + '/raster.npz'))
    for W in (64, 128):
        idx = np.nonzero(z['size'] == W)[0]
        cfg = dict(
            state_initializer=lambda: collections.OrderedDict(
                [('a', [sprite.Sprite(shape='circle', c0=200, c1=100, c2=50, opacity=128)]),
                 ('agent', [])]),
            physics=physics_lib.Physics(updates_per_env_step=1),
            task=tasks.CompositeTask(),
            action_space=action_spaces.Grid(action_layers='agent'),
            observers={'image': observers.PILRenderer(image_size=(W, W), bg_color=tuple(z['bg']))})
        env = environment.BatchedEnvironment(num_envs=len(idx), **cfg)
        env.reset()
        f, q = download(env)
        L, P = env.layout, env.compiled.program
        for i, k in enumerate(idx):
            nv = int(z['nv'][k])
            xy = z['xy'][k, :nv].astype(np.float64)
            v = (xy + np.where(xy >= 0, 0.5, -0.5)) / W   # (int)(W * v) == xy exactly
            q[i, L.o_nverts] = nv
            f[i, L.o_verts:L.o_verts + 2 * nv] = v.ravel()
        upload(env, f, q)
        img = env.observation()['image'].cpu().numpy()
        bad = [int(k) for i, k in enumerate(idx)
               if not np.array_equal(img[i, ::-1, :, 0], z['red'][k, :W, :W])]
        assert not bad, ('polygons that differ from Pillow at %d^2' % W, bad[:10], len(bad))


def test_full_size_properties():
    """BASELINE size (4096 envs x 32 sprites): determinism (same seed twice ->
    bit-identical records and frames), independence of batch position, and
    physical sanity (finite state, sprites stay in the arena)."""
    import torch
    outs = []
    for rep in range(2):
        env = make_env('colliding_predators_32', 4096, seed=1)
        env.reset()
        g = torch.Generator(device='cpu').manual_seed(0)
        for _ in range(5):
            a = (torch.rand((4096, 2), generator=g, dtype=torch.float64) * 2 - 1)
            out = env.step(a)
        f, q = download(env)
        outs.append((f, q, out.observation['image'].cpu().numpy()))
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][0], outs[1][0], equal_nan=True)
    assert np.array_equal(outs[0][2], outs[1][2])
    f, q, img = outs[0]
    L = env.layout
    pos = f[:, L.o_pos:L.o_pos + 2 * L.S].reshape(-1, L.S, 2)[:, 4:]
    assert np.isfinite(pos).all() and pos.min() > -0.1 and pos.max() < 1.1
    # a shard of the same global env indices reproduces the same envs
    env2 = make_env('colliding_predators_32', 64, seed=1, env_index0=128)
    env2.reset()
    g = torch.Generator(device='cpu').manual_seed(0)
    for _ in range(5):
        a = (torch.rand((4096, 2), generator=g, dtype=torch.float64) * 2 - 1)
        env2.step(a[128:192])
    f2, q2 = download(env2)
    assert np.array_equal(q2, q[128:192]) and np.array_equal(f2, f[128:192], equal_nan=True)


@pytest.mark.parametrize('name,n,steps,size,kernel', [('chase_avoid_torus', 4096, 8, None, 'specialised'),
                                                      ('colliding_predators_32', 4096, 8, None, 'specialised'),
                                                      ('functional_maze', 8192, 6, 128, 'specialised'),
                                                      ('falling_balls_64', 8192, 6, None, 'specialised'),
                                                      ('colliding_predators_32', 4096, 4, None, 'generic'),
                                                      ('falling_balls_64', 8192, 3, None, 'generic')])
def test_full_size_vs_oracle(name, n, steps, size, kernel, monkeypatch):
    """BASELINE.json's configs at their full per-GPU sizes against the oracle itself (OpenMP over envs
    makes it affordable): reset + a few steps in lock step -- integer records bit-exact, floats <= 1e-9,
    rewards / step types exact, and every one of the final frames bit-exact.  `kernel`: the binary that steps -- the
    program-specialised kernels bench.py times (lib/spec/step_<hash>.so, built by __graft_entry__.build(); the test FAILS
    when the engine did not pick one up, so the parity claim is about that binary) and, for two of the workloads, the generic
    kernels (MOOG_STEP_SPEC=0)."""
    import torch  # noqa: F401
    from moog import _spec, environment
    from moog_demos import example_configs
    if kernel == 'generic':
        monkeypatch.setenv('MOOG_STEP_SPEC', '0')
    else:
        monkeypatch.delenv('MOOG_STEP_SPEC', raising=False)
        monkeypatch.delenv('MOOG_SPEC_DIR', raising=False)
    if size is None:
        env = make_env(name, n, seed=17, env_index0=0)
    else:
        cfg = __import__('moog_demos.example_configs.' + name, fromlist=['x']).get_config(0, image_size=(size, size))
        env = environment.BatchedEnvironment(num_envs=n, seed=17, env_index0=0,
                                             layer_capacity=example_configs.capacity(name), **cfg)
    assert env.step_kernel() == kernel, (env.step_kernel(), _spec.path_of(env.compiled.program))
    print('%s steps with the %s kernel%s' % (name, kernel, ': ' + os.path.basename(_spec.path_of(env.compiled.program))
                                              if kernel == 'specialised' else ''))
    o = helpers.OracleEnv(env.compiled, n_envs=n, seed=17, env_index0=0)
    env.reset()
    o.reset(render=False)
    rs = np.random.RandomState(9)
    for k in range(steps):
        a = rs.randint(0, 5, size=n) if env._is_grid else rs.uniform(-1, 1, size=(n, 2))
        out = env.step(a)
        o.step(a, render=False)
        f, q = download(env)
        assert np.array_equal(q, o.i32), 'int state differs at step %d' % k
        with np.errstate(invalid='ignore'):
            err = np.abs(f - o.f64)
        err = np.where(np.isnan(f) & np.isnan(o.f64), 0, err)
        err = np.where(f == o.f64, 0, err)
        assert float(np.max(err)) <= 1e-9, (k, float(np.max(err)))
        assert np.array_equal(out.step_type.cpu().numpy(), o.step_type)
        assert helpers.same_or_nan(out.reward.cpu().numpy(), o.reward)
        o.f64[:], o.i32[:] = f, q   # lock step (removes 1-ulp libm / ocml drift)
    img = out.observation['image'].cpu().numpy()
    ref = o.render()
    bad = np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0]
    assert bad.size == 0, ('frames differ', bad[:8].tolist(), int(bad.size))


def test_gym_wrapper_contract():
    """The Gym surface pinned by the reference's tests/moog/env_wrappers/test_gym_wrapper.py:49-131:
    spaces, uint8 image observations, `done` exactly at the timeout step and False on
    the auto-reset step that follows."""
    import collections
    from moog import action_spaces, environment, observers, physics as physics_lib, sprite, tasks
    from moog.env_wrappers import gym_wrapper
    cfg = dict(
        state_initializer=lambda: collections.OrderedDict(
            [('agent', [sprite.Sprite(x=0.5, y=0.5, shape='square', scale=0.1, c0=255)])]),
        physics=physics_lib.Physics((physics_lib.Drag(coeff_friction=0.25), 'agent'),
                                    updates_per_env_step=5),
        task=tasks.CompositeTask(timeout_steps=5),
        action_space=action_spaces.Joystick(scaling_factor=0.01, action_layers='agent'),
        observers={'image': observers.PILRenderer(image_size=(64, 64))})
    env = gym_wrapper.GymWrapper(environment.Environment(**cfg))
    assert env.observation_space['image'].shape == (64, 64, 3)
    assert env.observation_space['image'].dtype == np.uint8
    assert env.action_space.shape == (2,)
    obs = env.reset()
    assert obs['image'].dtype == np.uint8 and obs['image'].shape == (64, 64, 3)
    for t in range(1, 6):
        obs, reward, done, info = env.step(np.array([0.5, -0.5]))
        assert done == (t == 5) and reward == 0
        assert info['discount'] == (0.0 if t == 5 else 1.0)
    obs, reward, done, info = env.step(np.array([0.5, -0.5]))   # auto-reset: FIRST timestep
    assert not done and reward == 0 and info['discount'] is None
    assert np.array_equal(env.render(), obs['image'])


def _shard_devices():
    """Device lists for the sharding test: two handles on one GPU always; distinct GPUs when the box has them."""
    import torch
    lists = [['cuda:0', 'cuda:0']]
    k = torch.cuda.device_count()
    if k >= 2:
        lists.append(['cuda:%d' % i for i in range(min(k, 8))])
    return lists


@pytest.mark.parametrize('which', ['one_gpu_two_handles', 'distinct_gpus'])
def test_multi_device_sharding_matches_single_engine(which):
    """Two (or, on a multi-GPU box, one per GPU) engine handles over a split env axis reproduce the
    single-engine batch: global-index RNG keys, no exchange between shards."""
    import torch
    from moog import sharding
    from moog_demos import example_configs
    devices = _shard_devices()
    if which == 'distinct_gpus' and len(devices) < 2:
        pytest.skip('one GPU visible: distinct devices cannot be exercised here (bench.py --gpus N / tools/bench_ranks.sh are the N > 1 path)')
    devices = devices[-1] if which == 'distinct_gpus' else devices[0]
    n = 96
    cfg = example_configs.load('chase_avoid_torus')
    multi = sharding.MultiDeviceEnvironment(n, devices, seed=4, **cfg)
    single = make_env('chase_avoid_torus', n, seed=4)
    a = multi.gather(multi.reset())
    b = single.reset()
    assert np.array_equal(a.observation['image'].numpy(), b.observation['image'].cpu().numpy())
    rs = np.random.RandomState(1)
    for _ in range(12):
        act = rs.uniform(-1, 1, size=(n, 2))
        a = multi.gather(multi.step(act))
        b = single.step(act)
    assert np.array_equal(a.step_type.numpy(), b.step_type.cpu().numpy())
    assert helpers.same_or_nan(a.reward.numpy(), b.reward.cpu().numpy())
    assert np.array_equal(a.observation['image'].numpy(), b.observation['image'].cpu().numpy())
    f = torch.cat([s.state_f64.cpu() for s in multi.shards]).numpy()
    assert np.array_equal(f, single.state_f64.cpu().numpy(), equal_nan=True)


def test_edge_cases_empty_and_dead_layers():
    """Empty layers, every prey vanished, a single env, and an env count that is not a
    multiple of anything: engine == oracle."""
    import torch
    for name, n in (('falling_balls', 1), ('pong', 3), ('functional_maze', 5)):
        env = make_env(name, n, seed=2)
        o = helpers.OracleEnv(env.compiled, n_envs=n, seed=2)
        env.reset()
        o.reset(render=False)
        L = env.layout
        if name == 'functional_maze':   # kill all prey: Reset(condition=layer empty) must arm
            s0, cnt = env.compiled.layer_slots['prey']
            env.state_i32[:, L.o_flags + s0:L.o_flags + s0 + cnt] = 0
            o.i32[:, L.o_flags + s0:L.o_flags + s0 + cnt] = 0
        rs = np.random.RandomState(3)
        for k in range(10):
            a = rs.randint(0, 5, size=n) if env._is_grid else rs.uniform(-1, 1, size=(n, 2))
            out = env.step(a)
            o.step(a, render=False)
            f, q = download(env)
            assert np.array_equal(q, o.i32), (name, k)
            assert np.array_equal(out.step_type.cpu().numpy(), o.step_type), (name, k)
            assert helpers.same_or_nan(out.reward.cpu().numpy(), o.reward), (name, k)
            o.f64[:], o.i32[:] = f, q
        assert np.array_equal(out.observation['image'].cpu().numpy(), o.render())


def test_cost_schedule_is_result_neutral():
    """Cost-ordered launch of the step kernel (a scheduling hint) changes nothing."""
    import torch
    outs = []
    for sched in (False, True):
        env = make_env('colliding_predators_32', 512, seed=8)
        if sched:
            env.enable_cost_schedule()
        env.reset()
        g = torch.Generator(device='cpu').manual_seed(1)
        for _ in range(12):
            out = env.step(torch.rand((512, 2), generator=g, dtype=torch.float64) * 2 - 1)
        f, q = download(env)
        outs.append((f, q, out.observation['image'].cpu().numpy(), out.reward.cpu().numpy()))
        if sched:
            assert float(env._cost.min()) > 0 and sorted(env._perm.cpu().tolist()) == list(range(512))
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][0], outs[1][0], equal_nan=True)
    assert np.array_equal(outs[0][2], outs[1][2])
    assert np.array_equal(outs[0][3], outs[1][3], equal_nan=True)


@pytest.mark.parametrize('name,n,G,steps', [('colliding_predators_32', 768, 4, 30), ('chase_avoid_torus', 512, 2, 25),
                                             ('functional_maze', 256, 8, 25), ('falling_balls_64', 256, 2, 12)])
def test_sub_batches_are_result_neutral(name, n, G, steps):
    """SubBatchedEnvironment: G asynchronous sub-batches, one stream each, queued several calls ahead without a join.  Every
    env's states, time steps and EVERY frame of every call equal those of one BatchedEnvironment of n envs (the random
    streams are keyed by the global env index; the parts only differ in how their launches interleave)."""
    import torch
    from moog import environment
    from moog_demos import example_configs
    g = torch.Generator(device='cpu').manual_seed(5)
    grid = None
    acts = []
    ref = make_env(name, n, seed=21)
    grid = ref._is_grid
    for k in range(steps):
        acts.append((torch.randint(0, 5, (n,), generator=g, dtype=torch.int32) if grid
                     else torch.rand((n, 2), generator=g, dtype=torch.float64) * 2 - 1).cuda())
    w = None

    def digest(ts):
        img = ts.observation['image']
        ww = (torch.arange(img[0].numel(), device=img.device, dtype=torch.int64) % 8191) + 1
        return ((img.reshape(img.shape[0], -1).to(torch.int64) * ww).sum(1).cpu().numpy(), ts.step_type.cpu().numpy(),
                np.nan_to_num(ts.reward.cpu().numpy(), nan=-7.0), np.nan_to_num(ts.discount.cpu().numpy(), nan=-7.0))

    ref.enable_cost_schedule()
    ref.reset()
    want = [digest(ref.step(a)) for a in acts]
    wf, wq = download(ref)
    ref.close()

    env = environment.SubBatchedEnvironment(num_envs=n, sub_batches=G, seed=21,
                                            layer_capacity=example_configs.capacity(name), **example_configs.load(name))
    env.enable_cost_schedule()
    env.reset()
    m = n // G
    got = [[None] * G for _ in range(steps)]
    # sub-batch g runs `lead[g]` calls ahead of the slowest one: the streams really interleave different calls
    lead = [(3 * g) % 5 for g in range(G)]
    done = [0] * G
    for k in range(steps + max(lead)):
        for gi in range(G):
            kk = k + lead[gi] - max(lead)
            if 0 <= kk < steps:
                env.step_async(gi, acts[kk][gi * m:(gi + 1) * m])
                got[kk][gi] = digest(env.recv(gi))   # (the digest reads on the caller's stream, behind recv's wait)
                done[gi] += 1
    assert done == [steps] * G
    torch.cuda.synchronize()
    f, q = env.state_f64.cpu().numpy(), env.state_i32.cpu().numpy()
    env.raise_faults()
    env.close()
    assert np.array_equal(q, wq)
    assert np.array_equal(f, wf, equal_nan=True)
    for k in range(steps):
        for part in range(4):
            cat = np.concatenate([got[k][gi][part] for gi in range(G)])
            assert np.array_equal(cat, want[k][part]), 'call %d, output %d differs in envs %s' % (
                k, part, np.nonzero(cat != want[k][part])[0][:8])
    # the synchronous whole-batch form
    env = environment.SubBatchedEnvironment(num_envs=n, sub_batches=G, seed=21,
                                            layer_capacity=example_configs.capacity(name), **example_configs.load(name))
    env.reset()
    for k in range(3):
        d = digest(env.step(acts[k]))
        for part in range(4):
            assert np.array_equal(d[part], want[k][part])
    env.close()


def observable_state(env, f, q):
    """What a record says about the episode: every scalar field, and the per-slot fields of the sprites that exist (a dead
    slot's leftovers -- the values of whichever sprite lived there last -- are nothing a component ever reads)."""
    from moog import _abi
    L, P = env.layout, env.compiled.program
    S = L.S
    alive = (q[:, L.o_flags:L.o_flags + S] & _abi.MOOG_F_ALIVE) != 0
    out = {}
    for name, width in (('o_pos', 2), ('o_vel', 2), ('o_angle', 1), ('o_angvel', 1), ('o_mass', 1), ('o_color', 3),
                        ('o_inertia', 2), ('o_maxr', 1), ('o_scale', 1), ('o_aspect', 1)):
        o = getattr(L, name)
        if o >= 0:
            v = f[:, o:o + width * S].reshape(len(f), S, width).copy()
            v[~alive] = 0
            out[name] = v
    for name in ('o_flags', 'o_nverts', 'o_opacity', 'o_shape', 'o_tele', 'o_valias', 'o_fmask'):
        o = getattr(L, name)
        if o >= 0:
            v = q[:, o:o + S].copy()
            v[~alive] = 0
            out[name] = v
    nv = np.where(alive, q[:, L.o_nverts:L.o_nverts + S], 0)
    verts = []
    for s_ in range(S):
        vo, cap = int(P.slot_voff[s_]), int(P.slot_vcap[s_])
        v = f[:, L.o_verts + 2 * vo:L.o_verts + 2 * (vo + cap)].reshape(len(f), cap, 2).copy()
        v[np.arange(cap)[None, :] >= nv[:, s_:s_ + 1]] = 0
        verts.append(v.reshape(len(f), -1))
    out['verts'] = np.concatenate(verts, axis=1)
    n_act = 2 * max(1, int(P.n_actions))
    for name, count in (('o_action', n_act), ('o_task', int(P.n_tasks)), ('o_rule', int(P.n_rules)), ('o_hdraw', int(P.n_hdraws)),
                        ('o_rule2', int(P.n_rules))):
        o = getattr(L, name)
        if o >= 0 and count > 0:
            out[name] = f[:, o:o + count]
    for name, count in (('o_step_count', 1), ('o_reset_next', 1), ('o_fault', 1), ('o_rng', 2),
                        ('o_maze', _abi.MOOG_MAX_MAZE + _abi.MOOG_MAX_MAZE_POINTS)):
        o = getattr(L, name)
        if o >= 0:
            out[name] = q[:, o:o + count]
    return out


@pytest.mark.parametrize('name,n,steps,min_episodes', [('bounce_box_contact_prediction', 96, 260, 192), ('red_green_l1', 64, 200, 64),
                                                       ('maze_zoo', 128, 150, 1), ('pacman', 32, 60, 1)])
def test_reset_pool_is_result_neutral(name, n, steps, min_episodes):
    """moog_engine_set_reset_pool: the next episode of every env is built on a side stream while the current one runs,
    and taken over by the step kernel when the episode ends.  Time steps, EVERY frame of every call and the final
    records equal those of an engine that resets in place, bit for bit -- over several episodes per env, with a host-side
    reset of some envs in the middle, and with records edited behind the engine's back: the episode counter of some envs
    moved on, a sprite built outside the initializer recoloured (pool records built before that are stale and must be
    rejected, not used)."""
    import torch
    g = torch.Generator(device='cpu').manual_seed(11)
    acts = None

    def digest(ts):
        img = ts.observation['image']
        ww = (torch.arange(img[0].numel(), device=img.device, dtype=torch.int64) % 8191) + 1
        return ((img.reshape(img.shape[0], -1).to(torch.int64) * ww).sum(1).cpu().numpy(), ts.step_type.cpu().numpy(),
                np.nan_to_num(ts.reward.cpu().numpy(), nan=-7.0), np.nan_to_num(ts.discount.cpu().numpy(), nan=-7.0))

    def run(pool):
        nonlocal acts
        env = make_env(name, n, seed=33, reset_pool=pool)
        if acts is None:
            grid = env._is_grid
            acts = [(torch.randint(0, 5, (n,), generator=g, dtype=torch.int32) if grid
                     else torch.rand((n, 2), generator=g, dtype=torch.float64) * 2 - 1).cuda() for _ in range(steps)]
        env.check_faults = False
        env.reset()
        out = []
        mask = torch.zeros(n, dtype=torch.bool, device='cuda')
        mask[::3] = True
        L, P = env.layout, env.compiled.program
        persist = [s_ for s_ in range(L.S) if P.slot_persist[s_]] if P.born_rule > 0 else []
        for k in range(steps):
            if k == steps // 3:   # the draws of these envs' next episodes come from another segment than the pool assumed
                env.state_i32[1::4, L.o_rng + 1] += 5
            if k == steps // 2:
                env.reset(mask)   # (a third of the envs start a new episode; the engine drops its pool)
            if k == (2 * steps) // 3 and persist:   # an input of the reset changes after the fills have read it
                env.state_f64[2::5, L.o_color + 3 * persist[0]] += 1.0
            out.append(digest(env.step(acts[k])))
        f, q = download(env)
        stats = env.reset_pool
        obs = observable_state(env, f, q)
        env.close()
        return out, obs, stats

    want, wobs, wstats = run(False)
    got, gobs, gstats = run(True)
    assert not wstats['on'] and gstats['on']
    episodes = sum(int((d[1] == 0).sum()) for d in want)
    assert episodes >= min_episodes, 'the run is too short to exercise the pool: %d episode starts' % episodes
    assert gstats['adopted'] > 0 and gstats['adopted'] + gstats['in_place'] == episodes, (gstats, episodes)
    if min_episodes > 1:
        assert gstats['rejected'] > 0   # the edited records
    for k in range(steps):
        for part in range(4):
            assert np.array_equal(got[k][part], want[k][part]), 'call %d, output %d differs in envs %s (pool %s)' % (
                k, part, np.nonzero(got[k][part] != want[k][part])[0][:8], gstats)
    for key in wobs:
        assert np.array_equal(gobs[key], wobs[key], equal_nan=True), (key, gstats)


@pytest.mark.parametrize('name,n,steps', [('predators_arena_l2', 96, 130), ('parallelogram_catch', 96, 80), ('match_to_sample_l3', 64, 80),
                                          ('callables_zoo', 64, 80)])
def test_late_reset_is_result_neutral(name, n, steps, monkeypatch):
    """Late reset (moog_engine_kernel_variant): a program that needs the rare components only to BUILD an episode is stepped
    by the kernel without them; the full reset kernel behind every step launch opens the episodes that kernel could not.
    Time steps, every frame and the final records equal those of the same program stepped by the kernel that carries
    everything (MOOG_NO_LATE_RESET=1), episode boundaries included."""
    import torch

    def digest(ts):
        img = ts.observation['image']
        ww = (torch.arange(img[0].numel(), device=img.device, dtype=torch.int64) % 8191) + 1
        return ((img.reshape(img.shape[0], -1).to(torch.int64) * ww).sum(1).cpu().numpy(), ts.step_type.cpu().numpy(),
                np.nan_to_num(ts.reward.cpu().numpy(), nan=-7.0), np.nan_to_num(ts.discount.cpu().numpy(), nan=-7.0))

    def run(late):
        if late:
            monkeypatch.delenv('MOOG_NO_LATE_RESET', raising=False)
        else:
            monkeypatch.setenv('MOOG_NO_LATE_RESET', '1')
        env = make_env(name, n, seed=17, reset_pool=False)
        variant = env.kernel_variant
        env.check_faults = False
        g = torch.Generator(device='cpu').manual_seed(4)
        grid = env._is_grid
        out = [digest(env.reset())]
        for k in range(steps):
            a = (torch.randint(0, 5, (n,), generator=g, dtype=torch.int32) if grid
                 else torch.rand((n, 2), generator=g, dtype=torch.float64) * 2 - 1)
            if k % 10 == 5:   # (episodes of some of these configs outlast the run: end a rotating quarter of them by hand)
                env.state_i32[(k // 10) % 4::4, env.layout.o_reset_next] = 1
            out.append(digest(env.step(a)))
        f, q = download(env)
        obs = observable_state(env, f, q)
        env.close()
        return out, obs, variant

    want, wobs, wv = run(False)
    got, gobs, gv = run(True)
    assert wv == (2, False) and gv == (1, True), (wv, gv)
    assert sum(int((d[1] == 0).sum()) for d in want[1:]) >= n // 4, 'too few episode boundaries in the run'
    for k in range(len(want)):
        for part in range(4):
            assert np.array_equal(got[k][part], want[k][part]), 'call %d, output %d differs in envs %s' % (
                k, part, np.nonzero(got[k][part] != want[k][part])[0][:8])
    for key in wobs:
        assert np.array_equal(gobs[key], wobs[key], equal_nan=True), key


def test_reset_pool_refusals():
    """The reset pool is refused where it cannot keep its promise (moog_engine_set_reset_pool): programs of the plain kernels,
    and programs whose initializer keeps a number across episodes (predators_arena's curriculum: the reset depends on the
    episode that has just ended).  'auto' only asks for it where an initializer plays physics forward."""
    from moog import _engine
    for name in ('colliding_predators', 'predators_arena_l2'):
        with pytest.raises(_engine.EngineError):
            make_env(name, 8, reset_pool=True)
        env = make_env(name, 8)   # 'auto'
        assert not env.reset_pool['on'] and env.reset_pool_refusal is None
        env.close()
    env = make_env('bounce_box_contact_prediction', 8)
    assert env.reset_pool['on']
    env.close()
    env = make_env('bounce_box_contact_prediction', 8, reset_pool=False)
    assert not env.reset_pool['on']
    env.close()


@pytest.mark.parametrize('name,n,steps,walls,edit', [('pacman', 48, 70, 136, 100), ('maze_zoo', 96, 60, None, 20),
                                                     ('pacman_l1', 32, 40, None, 3)])
def test_env_prefix_frames_equal(name, n, steps, walls, edit, monkeypatch):
    """Per-env prefix of the rasteriser (moog_engine_env_prefix): every env's leading sprites that stay put within an episode
    (a random maze's walls) live in a cached picture of the env's own.  EVERY frame of every call -- across resets, while
    food is being eaten (which shortens the prefix to the walls), after a record was edited behind the engine's back --
    equals the frame of an engine that draws every sprite every time."""
    import torch
    g = torch.Generator(device='cpu').manual_seed(3)

    def run(on):
        monkeypatch.setenv('MOOG_RASTER_ENV_BG', '1' if on else '0')
        env = make_env(name, n, seed=5)
        env.check_faults = False
        grid = env._is_grid
        gg = torch.Generator(device='cpu').manual_seed(3)
        frames = [env.reset().observation['image'].cpu().numpy().copy()]
        slots = [env.env_prefix_slots]
        L = env.layout
        for k in range(steps):
            a = (torch.randint(0, 5, (n,), generator=gg, dtype=torch.int32) if grid
                 else torch.rand((n, 2), generator=gg, dtype=torch.float64) * 2 - 1)
            if k == steps // 2:   # a wall of some envs recoloured by hand: their pictures are stale
                env.state_f64[::7, L.o_color + 3 * edit + 1] += 0.125
            if k == steps // 2 + 5:
                env.reset(torch.arange(n) % 5 == 0)
            frames.append(env.step(a).observation['image'].cpu().numpy().copy())
            slots.append(env.env_prefix_slots)
        env.close()
        return frames, slots

    want, s0 = run(False)
    got, s1 = run(True)
    assert max(s0) == 0
    if walls is not None:   # the prefix settles on the walls (food gets eaten, ghosts move), then ends at the sprite that was edited
        assert s1[steps // 2 - 1] == walls and s1[-1] == edit, s1
    for k in range(len(want)):
        assert np.array_equal(got[k], want[k]), 'frame of call %d differs in envs %s (prefix %s)' % (
            k, np.nonzero((got[k] != want[k]).reshape(n, -1).any(1))[0][:8], s1[k])


def test_layer_capacity_auto_grows_transparently():
    """layer_capacity='auto' (the reference's layers are unbounded Python lists, create_sprites.py:34, change_layer.py:43):
    first_person_predators_prey started from the fixtures' capacities -- which a 256-env random-policy batch overflows
    within ~60 calls (LAYER_FULL with fixed capacities) -- grows its layers when their high-water mark comes close, and every
    call's time steps and frames equal those of an engine that had roomy layers from the start."""
    import torch
    from moog import environment
    from moog_demos import example_configs
    name, n, steps = 'first_person_predators_prey', 256, 90
    cfg = example_configs.load(name)
    small = dict(example_configs.capacity(name))
    big = environment.BatchedEnvironment(num_envs=n, seed=5, layer_capacity={'prey': 32, 'predators': 96}, **cfg)
    auto = environment.BatchedEnvironment(num_envs=n, seed=5, layer_capacity=dict(small, auto=True), **cfg)
    big.reset()
    auto.reset()
    g = torch.Generator(device='cpu').manual_seed(3)
    for k in range(steps):
        a = torch.rand((n, 2), generator=g, dtype=torch.float64) * 2 - 1
        x, y = big.step(a), auto.step(a)
        assert torch.equal(x.step_type, y.step_type), k
        assert torch.equal(torch.nan_to_num(x.reward, nan=-7.), torch.nan_to_num(y.reward, nan=-7.)), k
        assert torch.equal(x.observation['image'], y.observation['image']), 'frames differ at call %d' % k
    grown = getattr(auto, 'capacity_growths', [])
    assert grown, 'the batch never came close to the fixtures\' capacities: the test does not exercise the growth'
    use = auto.layer_usage()
    assert all(u['dropped'] == 0 for u in use.values()), use
    # the same sprites, layer by layer (slots differ: the layers are wider)
    for env_i in (0, n // 2, n - 1):
        sa, sb = big.sprites(env_i), auto.sprites(env_i)
        for layer in sa:
            assert len(sa[layer]) == len(sb[layer]), layer
            for p, q in zip(sa[layer], sb[layer]):
                assert p['x'] == q['x'] and p['y'] == q['y'] and np.array_equal(p['vertices'], q['vertices'])
    big.close()
    auto.close()


def _simulation_env():
    """The environment of the reference's tests/moog/env_wrappers/test_simulation.py:32-58."""
    import collections
    from moog import action_spaces, environment, game_rules, observers, physics as physics_lib
    from moog import sprite, tasks
    from moog.env_wrappers import simulation

    def _state_initializer():
        agent = sprite.Sprite(x=0.5, y=0.5, scale=0.1, c0=128)
        target = sprite.Sprite(x=0.75, y=0.5, scale=0.1, c1=128)
        return collections.OrderedDict([('agent', [agent]), ('target', [target])])

    def _modify_meta_state(meta_state):
        meta_state['key'] = meta_state['key'] + 1

    env = environment.Environment(
        state_initializer=_state_initializer,
        physics=physics_lib.Physics(),
        task=tasks.ContactReward(1., 'agent', 'target', reset_steps_after_contact=2),
        action_space=action_spaces.Grid(0.1, action_layers='agent', control_velocity=True),
        observers={'image': observers.PILRenderer(image_size=(64, 64))},
        meta_state_initializer=lambda: {'key': 0},
        game_rules=(game_rules.ModifyMetaState(_modify_meta_state),))
    return simulation.SimulationEnvironment(env)


def test_simulation_wrapper_reference_known_answers():
    """tests/moog/env_wrappers/test_simulation.py:64-119 (testStep, testSimStepSimPop):
    the same action scripts reach LAST on the same steps through real steps, simulated
    steps and pops, and the meta-state counter ends at 7."""
    env = _simulation_env()
    episode_actions = [1, 4, 3, 1, 2, 0]
    env.reset()
    for a in episode_actions[:-1]:
        assert not env.step(a).last()
    assert env.step(episode_actions[-1]).last()

    env = _simulation_env()
    env.reset()
    for a in [1, 4, 3]:
        assert not env.sim_step(a).last()
    env.sim_pop(-1)
    for a in [3, 1, 2]:
        assert not env.sim_step(a).last()
    assert env.sim_step(0).last()
    for i in [-1, -2]:
        env.sim_pop(i)
    for a in [1, 2]:
        assert not env.sim_step(a).last()
    assert env.sim_step(0).last()
    assert env.sim_step(0) is None          # no simulation across the episode boundary
    for a in episode_actions[:-1]:
        assert not env.step(a).last()
    assert env.step(episode_actions[-1]).last()
    assert env.meta_state['key'] == 7


def test_simulation_wrapper_batched_snapshot_restore():
    """Batched sim_step / sim_pop: popping to level 0 restores the state records exactly
    (RNG counters aside), and re-simulating the same actions reproduces the same states."""
    from moog.env_wrappers import simulation
    import torch
    g = torch.Generator(device='cpu').manual_seed(3)
    acts = [(torch.rand((256, 2), generator=g, dtype=torch.float64) * 2 - 1) for _ in range(6)]
    for seed in range(5, 25):                # a batch in which no episode ends within the script
        probe = make_env('colliding_predators_32', 256, seed=seed)
        probe.reset()
        if not any(bool((probe.step(a).step_type == 2).any().item()) for a in acts):
            break
    env = make_env('colliding_predators_32', 256, seed=seed)
    sim = simulation.SimulationEnvironment(env)
    sim.reset()
    sim.step(acts[0])
    f0, q0 = download(env)
    o = env.layout.o_rng
    for a in acts[1:4]:
        assert sim.sim_step(a) is not None
    f3, q3 = download(env)
    sim.sim_pop(0)
    f, q = download(env)
    q[:, o:o + 4] = q0[:, o:o + 4]
    assert np.array_equal(f, f0, equal_nan=True) and np.array_equal(q, q0)
    for a in acts[1:4]:
        sim.sim_step(a)
    f, q = download(env)
    q[:, o:o + 4] = q3[:, o:o + 4]
    assert np.array_equal(f, f3, equal_nan=True) and np.array_equal(q, q3)
    sim.sim_pop(1)                           # back to the state after the first sim_step
    assert len(sim.stack) == 1
    ts = sim.step(acts[1])                   # a real step rewinds to level 0 first
    assert sim.stack == []
    env2 = make_env('colliding_predators_32', 256, seed=seed)
    env2.reset()
    env2.step(acts[0]); env2.step(acts[1])
    f2, q2 = download(env2)
    f, q = download(env)
    q[:, o:o + 4] = q2[:, o:o + 4]
    assert np.array_equal(f, f2, equal_nan=True) and np.array_equal(q, q2)
    del ts


@pytest.mark.parametrize('level', [10, 11, 12])
def test_tether_known_answers(level):
    """The reference's own tether scenarios (tests/moog/physics/test_tether_physics.py:109-215)
    through `physics.step` on the engine, 1e-3 as there."""
    from test_oracle_golden import check_tether_kat
    env = make_env('tether_zoo_l%d' % level, 1)
    env.reset()
    s0 = env.compiled.layer_slots['sprites'][0]

    def sprite_state():
        pos = env.field('position')[0].cpu().numpy()
        vel = env.field('velocity')[0].cpu().numpy()
        w = env.field('angle_vel')[0].cpu().numpy()
        return [(pos[s], vel[s], w[s]) for s in range(s0, s0 + 3)]
    check_tether_kat(level, sprite_state, env.physics_step)


def test_tether_zipped_layer_mismatch_raises():
    """TetherZippedLayers over layers of different lengths raises ValueError
    (tether_physics.py:192-198)."""
    import collections
    from moog import action_spaces, environment, observers, physics as physics_lib, sprite, tasks
    cfg = dict(
        state_initializer=lambda: collections.OrderedDict([
            ('a', [sprite.Sprite(x=0.3, y=0.5, scale=0.1), sprite.Sprite(x=0.7, y=0.5, scale=0.1)]),
            ('b', [sprite.Sprite(x=0.5, y=0.2, scale=0.1)])]),
        physics=physics_lib.Physics(
            corrective_physics=[physics_lib.TetherZippedLayers(('a', 'b'))], updates_per_env_step=2),
        task=tasks.CompositeTask(timeout_steps=5),
        action_space=action_spaces.Joystick(scaling_factor=0.01, action_layers='a'),
        observers={'image': observers.PILRenderer(image_size=(64, 64))})
    env = environment.BatchedEnvironment(num_envs=2, **cfg)
    env.reset()
    env.step(np.zeros((2, 2)))
    with pytest.raises(ValueError):
        env.raise_faults()


def test_logging_wrapper_writes_the_reference_format(tmp_path):
    """LoggingEnvironment (env_wrappers/logger.py:33-224): same files, same step structure
    and the same logged values as the reference's logger on the same action script (fixture:
    tests/golden/logger_tether_zoo_l0.json, a config without randomness)."""
    import json
    import os
    from moog import environment
    from moog.env_wrappers import logger
    from moog_demos import example_configs
    with open(os.path.join(helpers.GOLDEN, 'logger_tether_zoo_l0.json')) as f:
        ref = json.load(f)
    env = logger.LoggingEnvironment(
        environment.Environment(keep_sprite_factors=True, **example_configs.load('tether_zoo_l0')),
        log_dir=str(tmp_path))
    env.reset()
    for a in ref['actions']:
        env.step(np.array(a))
    files = sorted(fn for fn in os.listdir(env.log_dir) if fn.isdigit())
    assert files == ['%05d' % i for i in range(len(ref['episodes']))]
    with open(os.path.join(env.log_dir, 'attributes.txt')) as f:
        assert json.load(f) == ref['attributes']
    with open(os.path.join(env.log_dir, 'description.txt')) as f:
        assert f.read() == ref['description']
    id_col = ref['attributes'].index('id')
    for fn, ref_ep in zip(files, ref['episodes']):
        with open(os.path.join(env.log_dir, fn)) as f:
            ep = json.load(f)
        assert len(ep) == len(ref_ep)
        ids, ref_ids = {}, {}
        for step, ref_step in zip(ep, ref_ep):
            assert [k for k, _ in step[:5]] == [k for k, _ in ref_step[:5]] == [
                'time', 'reward', 'step_type', 'action', 'meta_state']
            assert step[1][1] == ref_step[1][1] and step[2][1] == ref_step[2][1]
            assert np.allclose(step[3][1], ref_step[3][1]) and step[4][1] == ref_step[4][1]
            assert [l[0] for l in step[5]] == [l[0] for l in ref_step[5]]
            for (_, sprites), (_, ref_sprites) in zip(step[5], ref_step[5]):
                assert len(sprites) == len(ref_sprites)
                for s, r in zip(sprites, ref_sprites):
                    assert len(s) == len(r), 'vertices are logged on the same steps'
                    for j, (a, b) in enumerate(zip(s, r)):
                        if j == id_col:      # ids differ, identity over time must not
                            continue
                        if isinstance(b, (int, float)) and not isinstance(b, bool):
                            assert abs(a - b) <= 1e-9, (ref['attributes'][j], a, b)
                        elif isinstance(b, list):
                            assert np.allclose(a, b, atol=1e-9)
                        else:
                            assert a == b, (ref['attributes'][j], a, b)
            # the same sprite keeps the same id from step to step in both logs
            cur = [s[id_col] for _, sprites in step[5] for s in sprites]
            ref_cur = [s[id_col] for _, sprites in ref_step[5] for s in sprites]
            same = [c in ids for c in cur]
            ref_same = [c in ref_ids for c in ref_cur]
            assert same == ref_same
            ids, ref_ids = set(cur), set(ref_cur)


def test_dynamic_layer_overflow_is_reported():
    """The reference's layers are unbounded Python lists; the engine's have a capacity
    (`layer_capacity`).  Appending to a full layer sets a fault that surfaces as RuntimeError."""
    from moog import environment
    from moog_demos import example_configs
    env = environment.BatchedEnvironment(num_envs=64, seed=2, layer_capacity={'predators': 1, 'prey': 1},
                                         **example_configs.load('rules_zoo_l1'))
    env.reset()
    with pytest.raises(RuntimeError) as err:
        for _ in range(12):
            env.step(np.zeros((64, 2)))
    assert bool((env.field('alive')[:, env.compiled.layer_slots['predators'][0]]).any())
    # the error names the overflowing layers and the demand; layer_usage() is the sizing hint
    assert 'Overflowing layers' in str(err.value) and 'high_water' in str(err.value)
    use = env.layer_usage()
    assert set(use) == {'predators', 'prey'}
    assert any(u['dropped'] > 0 and u['high_water'] == u['capacity'] + 1 for u in use.values())
    # with room to spare nothing is dropped and the high-water mark is the capacity to ask for
    env2 = environment.BatchedEnvironment(num_envs=64, seed=2, layer_capacity={'predators': 24, 'prey': 24},
                                          **example_configs.load('rules_zoo_l1'))
    env2.reset()
    for _ in range(12):
        env2.step(np.zeros((64, 2)))
    use2 = env2.layer_usage()
    assert all(u['dropped'] == 0 and 1 <= u['high_water'] <= u['capacity'] for u in use2.values()), use2


def test_raw_state_observer():
    """observers.RawState (raw_state.py:6-22) next to the renderer: the facade returns the
    state as an OrderedDict of sprite dicts, the batched engine a lazy view of the records."""
    import collections
    from moog import action_spaces, environment, observers, physics as physics_lib, sprite, tasks
    cfg = dict(
        state_initializer=lambda: collections.OrderedDict([
            ('walls', []), ('agent', [sprite.Sprite(x=0.25, y=0.75, shape='triangle', scale=0.1, c0=255)])]),
        physics=physics_lib.Physics((physics_lib.Drag(coeff_friction=0.25), 'agent'), updates_per_env_step=2),
        task=tasks.CompositeTask(timeout_steps=5),
        action_space=action_spaces.Joystick(scaling_factor=0.01, action_layers='agent'),
        observers={'image': observers.PILRenderer(image_size=(64, 64)), 'state': observers.RawState()})
    env = environment.Environment(**cfg)
    ts = env.reset()
    assert list(ts.observation) == ['image', 'state'] and ts.observation['image'].shape == (64, 64, 3)
    state = ts.observation['state']
    assert list(state) == ['walls', 'agent'] and state['walls'] == []
    agent = state['agent'][0]
    assert abs(agent['x'] - 0.25) < 1e-12 and abs(agent['y'] - 0.75) < 1e-12   # centroid shift, sprite.py:406
    assert (agent['shape'], agent['c0']) == ('triangle', 255.0)
    assert agent['vertices'].shape == (3, 2)
    ts = env.step(np.array([1.0, 0.0]))
    assert ts.observation['state']['agent'][0]['x'] > 0.25
    assert 'state' not in env.observation_spec()


def test_composite_action_dict_api():
    """Composite (composite.py:51-62): dict actions keyed like the sub-spaces, through the
    batched engine (dict of tensors) and the single-env facade (dict of arrays); both equal
    the packed-tensor form."""
    import torch
    from moog import environment
    from moog_demos import example_configs
    cfg = example_configs.load('actions_zoo')
    env = environment.BatchedEnvironment(num_envs=32, seed=3, **cfg)
    env2 = environment.BatchedEnvironment(num_envs=32, seed=3, **example_configs.load('actions_zoo'))
    env.reset(); env2.reset()
    assert set(env.action_spec()) == {'agent_0', 'agent_1', 'eye'}
    g = torch.Generator().manual_seed(0)
    for _ in range(6):
        joy = torch.rand((32, 2), generator=g, dtype=torch.float64) * 2 - 1
        move = torch.randint(0, 5, (32,), generator=g)
        eye = torch.rand((32, 2), generator=g, dtype=torch.float64)
        env.step({'agent_0': joy, 'agent_1': move, 'eye': eye})
        packed = torch.zeros((32, 3, 2), dtype=torch.float64)
        packed[:, 0], packed[:, 1, 0], packed[:, 2] = joy, move.double(), eye
        env2.step(packed)
    f, q = download(env)
    f2, q2 = download(env2)
    assert np.array_equal(f, f2, equal_nan=True) and np.array_equal(q, q2)
    with pytest.raises(KeyError):
        env.step({'agent_0': joy})
    single = environment.Environment(**example_configs.load('actions_zoo'))
    single.reset()
    ts = single.step({'agent_0': np.array([0.5, -0.5]), 'agent_1': 3, 'eye': np.array([0.2, 0.9])})
    assert ts.observation['image'].shape == (64, 64, 3)
    ra = env.random_action()
    assert set(ra) == {'agent_0', 'agent_1', 'eye'} and ra['agent_1'].shape == (32,)


def _poly_env(W, n_envs, n_sprites, opacities):
    import collections
    from moog import environment, action_spaces, observers, physics as physics_lib, sprite, tasks
    cfg = dict(
        state_initializer=lambda: collections.OrderedDict(
            [('a', [sprite.Sprite(shape='circle', c0=40 + 60 * i, c1=200 - 50 * i, c2=30 + 70 * i,
                                  opacity=opacities[i % len(opacities)]) for i in range(n_sprites)]),
             ('agent', [])]),
        physics=physics_lib.Physics(updates_per_env_step=1),
        task=tasks.CompositeTask(),
        action_space=action_spaces.Grid(action_layers='agent'),
        observers={'image': observers.PILRenderer(image_size=(W, W), bg_color=(10, 20, 30))})
    return environment.BatchedEnvironment(num_envs=n_envs, **cfg)


@pytest.mark.parametrize('W', [64, 128])
def test_raster_degenerate_polygon_fuzz(W):
    """Differential fuzz of the rasteriser's rare paths against the oracle renderer (itself
    validated against Pillow): polygons on a coarse integer lattice (coinciding vertices,
    zero-width spikes, runs of horizontal edges, several fix-ups per row), combs with more
    than 12 crossings per row, vertices far off the canvas, all-horizontal polygons, four
    overlapping sprites per frame with mixed opacity."""
    n, ns = 2048, 4
    env = _poly_env(W, n, ns, (255, 128, 255, 77))
    env.reset()
    f, q = download(env)
    L, P = env.layout, env.compiled.program
    import os
    rs = np.random.RandomState(1234 + W + 1000 * int(os.environ.get('MOOG_FUZZ_SEED', '0')))
    for e in range(n):
        kind = e % 8
        for s in range(ns):
            nv = int(rs.randint(3, 31))
            if kind <= 2:      # coarse lattice: heavy degeneracy
                step = (2, 3, 5)[kind]
                ox, oy = rs.randint(-4, W - 8, size=2)
                xy = np.stack([ox + step * rs.randint(0, 5, size=nv), oy + step * rs.randint(0, 5, size=nv)], 1)
            elif kind == 3:    # comb: many crossings per row
                t = np.arange(nv)
                xy = np.stack([rs.randint(0, W // 2) + 2 * t, np.where(t % 2 == 0, rs.randint(0, W // 2), rs.randint(W // 2, W)) + rs.randint(-2, 3, size=nv)], 1)
            elif kind == 4:    # far off-canvas vertices mixed with near ones
                xy = rs.randint(-10, W + 10, size=(nv, 2))
                far = rs.rand(nv) < 0.3
                xy[far] = rs.choice([-30000, -17000, 17000, 30000], size=(int(far.sum()), 2))   # (the engine clamps at +-32000)
            elif kind == 5:    # all vertices on one or two rows
                xy = np.stack([rs.randint(-5, W + 5, size=nv), rs.randint(0, W) + rs.randint(0, 2, size=nv)], 1)
            elif kind == 6:    # small random polygon straddling a canvas border
                c = rs.choice([-2, 0, W - 3, W])
                xy = np.stack([c + rs.randint(-6, 7, size=nv), rs.choice([-2, 0, W - 3, W]) + rs.randint(-6, 7, size=nv)], 1)
            else:              # plain random
                xy = rs.randint(-8, W + 8, size=(nv, 2))
            xy = xy.astype(np.float64)
            v = (xy + np.where(xy >= 0, 0.5, -0.5)) / W   # (int)(W * v) == xy exactly
            v0 = P.slot_voff[s]
            q[e, L.o_nverts + s] = nv
            f[e, L.o_verts + 2 * v0:L.o_verts + 2 * (v0 + nv)] = v.ravel()
    upload(env, f, q)
    img = env.observation()['image'].cpu().numpy()
    o = helpers.OracleEnv(env.compiled, n_envs=n)
    o.f64[:], o.i32[:] = f, q
    ref = o.render()
    bad = np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0]
    assert bad.size == 0, ('frames differ', [(int(b), int(b) % 8) for b in bad[:10]], int(bad.size))


@pytest.mark.parametrize('name', ['colliding_predators_32', 'chase_avoid_torus', 'lambda_zoo'])
def test_step_register_variants_agree(name, monkeypatch):
    """The step kernel exists in two register allocations (3 / 4 waves per SIMD, chosen per program);
    both give bit-identical records, rewards and frames."""
    import torch
    n = 128
    envs = []
    for wps in ('3', '4'):
        monkeypatch.setenv('MOOG_STEP_WPS', wps)
        envs.append(make_env(name, n, seed=5, env_index0=40))
    rs = np.random.RandomState(2)
    for env in envs:
        env.reset()
    for k in range(20):
        a = rs.randint(0, 5, size=n) if envs[0]._is_grid else rs.uniform(-1, 1, size=(n, 2))
        outs = [env.step(a) for env in envs]
        (f0, q0), (f1, q1) = download(envs[0]), download(envs[1])
        assert np.array_equal(q0, q1), k
        assert np.array_equal(f0.view(np.int64), f1.view(np.int64)), k
        assert np.array_equal(outs[0].observation['image'].cpu().numpy(), outs[1].observation['image'].cpu().numpy()), k
        assert helpers.same_or_nan(outs[0].reward.cpu().numpy(), outs[1].reward.cpu().numpy())


# ---- round 2: the checks that used to reach only the oracle, now through the HIP engine -------------------

def test_collision_known_answers_hip():
    """The reference's own 19 collision scenarios (tests/moog/physics/test_collisions.py:101-293) through the
    HIP engine: the tabulated answers (atol 1e-3, as in the reference) and the exact outcomes the reference
    computes for them (collisions_kat.npz, 1e-9)."""
    import test_oracle_golden as tog
    from moog import environment
    table = np.load(helpers.GOLDEN + '/collisions_kat.npz')['final']
    kat = tog.helpers_kat_cases()
    assert len(kat) == table.shape[0] == 19
    for i, case in enumerate(kat):
        env = environment.BatchedEnvironment(num_envs=1, **tog.kat_config(case))
        env.reset()
        for _ in range(case['steps']):
            env.physics_step()
        f, _ = download(env)
        L = env.layout
        f = f[0]
        got = np.concatenate([f[L.o_pos:L.o_pos + 2], f[L.o_vel:L.o_vel + 2], f[L.o_angvel:L.o_angvel + 1],
                              f[L.o_pos + 2:L.o_pos + 4], f[L.o_vel + 2:L.o_vel + 4],
                              f[L.o_angvel + 1:L.o_angvel + 2]])
        assert np.allclose(got, table[i], atol=1e-9, rtol=0), (i, got, table[i])
        for k, v in case['expected'].items():
            assert np.allclose(got[tog.KAT_SLICES[k]], v, atol=1e-3), (i, k, got[tog.KAT_SLICES[k]], v)
        env.close()


def _corpus_env(n, nverts_a, nverts_b, portal=False):
    """n envs with two polygon sprites (slots 0 and 1) whose vertices the test overwrites, and a small
    sprite in slot 2.  portal=False: layers a / b / c with ContactReward(a, b); portal=True: the two
    polygons are the portals of a Portal rule that teleports the sprite of layer c."""
    import collections
    from moog import action_spaces, environment, game_rules, observers, physics as physics_lib, sprite, tasks
    poly = lambda k: np.array([[np.cos(2 * np.pi * i / k), np.sin(2 * np.pi * i / k)] for i in range(k)])
    A = lambda: sprite.Sprite(x=0.3, y=0.3, shape=poly(nverts_a), scale=0.1, c0=255)
    B = lambda: sprite.Sprite(x=0.7, y=0.7, shape=poly(nverts_b), scale=0.1, c1=255)
    C = lambda: sprite.Sprite(x=5., y=5., shape='square', scale=0.01, c2=255)
    if portal:
        layers = lambda: collections.OrderedDict([('p', [A(), B()]), ('c', [C()]), ('agent', [])])
        rules = (game_rules.Portal(teleporting_layer='c', portal_layer='p'),)
        task = tasks.CompositeTask()
    else:
        layers = lambda: collections.OrderedDict([('a', [A()]), ('b', [B()]), ('c', [C()]), ('agent', [])])
        rules = ()
        task = tasks.CompositeTask(tasks.ContactReward(1., layers_0='a', layers_1='b'))
    env = environment.BatchedEnvironment(
        num_envs=n, state_initializer=layers, physics=physics_lib.Physics(updates_per_env_step=1), task=task,
        action_space=action_spaces.Grid(action_layers='agent'),
        observers={'image': observers.PILRenderer(image_size=(64, 64))}, game_rules=rules)
    env.reset()
    return env


def _write_polygon(env, f, q, slot, verts, nv, row):
    L, P = env.layout, env.compiled.program
    o = L.o_verts + 2 * P.slot_voff[slot]
    f[row, o:o + 2 * nv] = verts[:nv].ravel()
    q[row, L.o_nverts + slot] = nv
    f[row, L.o_maxr + slot] = 1e3           # the bounding-circle shortcut never decides
    q[row, L.o_flags + slot] &= ~2          # not a symmetric circle: the polygon tests decide


@pytest.mark.parametrize('corpus', ['predicates.npz', 'predicates_nan.npz'])
def test_matplotlib_predicates_hip(corpus):
    """The 3000-pair matplotlib corpus (Path.intersects_path(filled=True), 12 contains_point probes per
    polygon) through the HIP predicates: overlaps_sprite via ContactReward, contains_point via Portal.
    predicates_nan.npz: polygons without a finite vertex (empty paths for matplotlib: they overlap everything)."""
    z = dict(np.load(helpers.GOLDEN + '/' + corpus))
    n = len(z['hit'])
    cap_a, cap_b = int(z['na'].max()), int(z['nb'].max())
    # -- overlaps_sprite: reward 1 exactly when the two paths intersect
    env = _corpus_env(n, cap_a, cap_b)
    f, q = download(env)
    for i in range(n):
        _write_polygon(env, f, q, 0, z['va'][i], int(z['na'][i]), i)
        _write_polygon(env, f, q, 1, z['vb'][i], int(z['nb'][i]), i)
    upload(env, f, q)
    out = env.step(np.full(n, 4, np.int32))
    got = out.reward.cpu().numpy() == 1.0
    bad = np.nonzero(got != z['hit'].astype(bool))[0]
    assert bad.size == 0, ('intersects_path differs from matplotlib', bad[:10].tolist(), int(bad.size))
    env.close()
    if 'pts' not in z:
        return
    # -- contains_point: sprite 'c' is teleported (to portal 'b', parked far away) exactly when its centre is
    #    inside portal 'a'
    npts = z['pts'].shape[1]
    env = _corpus_env(n, cap_a, 4, portal=True)
    f0, q0 = download(env)
    L = env.layout
    far = np.array([[9., 9.], [9.1, 9.], [9.1, 9.1], [9., 9.1]])
    for i in range(n):
        _write_polygon(env, f0, q0, 0, z['va'][i], int(z['na'][i]), i)
        _write_polygon(env, f0, q0, 1, far, 4, i)
    f0[:, L.o_pos + 2:L.o_pos + 4] = 9.05
    bad_in = 0
    for k in range(npts):
        f = f0.copy()
        f[:, L.o_pos + 4:L.o_pos + 6] = z['pts'][:, k, :]
        upload(env, f, q0)
        env.step(np.full(n, 4, np.int32))
        f1, _ = download(env)
        moved = np.abs(f1[:, L.o_pos + 4] - 9.05) < 1e-12
        bad_in += int((moved != z['inside'][:, k].astype(bool)).sum())
    assert bad_in == 0, bad_in


def test_free_running_window_full_size():
    """One un-resynchronised window at BASELINE size: 4096 envs of colliding_predators_32 stepped by the
    engine and by the oracle from the same reset (same Philox streams, same actions), no copying of state
    in between.  Integer records stay identical and floats within the 1e-5 budget of BASELINE.json for the
    whole window.  The window is as long as the chaotic dynamics allow for two implementations that differ
    by library-level roundings (1 ulp in sin / cos at sprite creation): measured with tools/dbg/drift.py the
    largest difference over the batch is 1.6e-12 after 20 steps and passes 1e-5 in the first env at step 23
    (every collision multiplies a difference), so the window is 16 steps."""
    n, steps = 4096, 16
    env = make_env('colliding_predators_32', n, seed=23, env_index0=0)
    o = helpers.OracleEnv(env.compiled, n_envs=n, seed=23, env_index0=0)
    env.reset()
    o.reset(render=False)
    rs = np.random.RandomState(4)
    worst = 0.0
    for k in range(steps):
        a = rs.uniform(-1, 1, size=(n, 2))
        out = env.step(a)
        o.step(a, render=False)
        f, q = download(env)
        assert np.array_equal(q, o.i32), 'int state differs at step %d' % k
        with np.errstate(invalid='ignore'):
            err = np.where(f == o.f64, 0, np.abs(f - o.f64))
        err = np.where(np.isnan(f) & np.isnan(o.f64), 0, err)
        worst = max(worst, float(np.max(err)))
        assert worst <= TOL, (k, worst)
        assert np.array_equal(out.step_type.cpu().numpy(), o.step_type)
        assert helpers.same_or_nan(out.reward.cpu().numpy(), o.reward)
    print('free-running full-size window: worst float difference', worst)


def test_falling_balls_64_pile_up():
    """BASELINE config 5 in the regime the short runs never reach: after ~45 steps a fifth to a tenth of the sixty balls
    lie in a pile on the floor (recursion depth 2, dozens of simultaneous contacts per substep; the
    reference's own run of this crowded variant also shoots balls out of the open top).  The engine runs alone to step 45, then
    engine and oracle run in lock step: 1024 envs for 40 steps, and the full 8192 envs for 8 steps; integer
    records exact, floats <= 1e-9, rewards / step types exact, final frames bit-exact."""
    for n, pre, steps in ((1024, 45, 40), (8192, 45, 8)):
        env = make_env('falling_balls_64', n, seed=31, env_index0=0)
        o = helpers.OracleEnv(env.compiled, n_envs=n, seed=31, env_index0=0)
        env.reset()
        rs = np.random.RandomState(12)
        for _ in range(pre):
            env.step(rs.randint(0, 5, size=n))
        f, q = download(env)
        o.f64[:], o.i32[:] = f, q
        L = env.layout
        y = f[:, L.o_pos + 1:L.o_pos + 2 * L.S:2][:, 4:]
        assert (y < 0.3).mean() > 0.1, 'no pile on the floor yet'   # (the crowded reference run also ejects balls upwards)
        for k in range(steps):
            a = rs.randint(0, 5, size=n)
            out = env.step(a)
            o.step(a, render=False)
            f, q = download(env)
            assert np.array_equal(q, o.i32), 'int state differs at step %d' % k
            with np.errstate(invalid='ignore'):
                err = np.where(f == o.f64, 0, np.abs(f - o.f64))
            err = np.where(np.isnan(f) & np.isnan(o.f64), 0, err)
            assert float(np.max(err)) <= 1e-9, (n, k, float(np.max(err)))
            assert np.array_equal(out.step_type.cpu().numpy(), o.step_type)
            assert helpers.same_or_nan(out.reward.cpu().numpy(), o.reward)
            o.f64[:], o.i32[:] = f, q
        img = out.observation['image'].cpu().numpy()
        assert np.array_equal(img, o.render())
        env.close()


def test_falling_balls_64_pile_at_rest():
    """BASELINE config 5 where it is most expensive (11 ms per step in round 3): the pile AT REST, calls 100-130 of an
    episode -- hundreds of path tests and contact searches per env and call, recursion depth 2, balls stacked on balls.
    2048 envs run alone to call 100; then 24 calls in lock step with the oracle (integer records exact, floats <= 1e-9,
    rewards / step types exact, final frames bit-exact), and from the same state a free-running window of 6 calls, no
    state copied in between (integers exact, floats inside BASELINE.json's 1e-5)."""
    n, pre, steps, free = 2048, 100, 24, 6
    env = make_env('falling_balls_64', n, seed=37, env_index0=0)
    o = helpers.OracleEnv(env.compiled, n_envs=n, seed=37, env_index0=0)
    env.reset()
    rs = np.random.RandomState(13)
    for _ in range(pre):
        env.step(rs.randint(0, 5, size=n))
    f0, q0 = download(env)
    L = env.layout
    y = f0[:, L.o_pos + 1:L.o_pos + 2 * L.S:2][:, 4:]
    assert (y < 0.3).mean() > 0.1, 'no pile on the floor'
    acts = [rs.randint(0, 5, size=n) for _ in range(steps)]
    # free-running window first (then the lock-step run restarts from the saved state)
    o.f64[:], o.i32[:] = f0, q0
    worst = 0.0
    for k in range(free):
        out = env.step(acts[k])
        o.step(acts[k], render=False)
        f, q = download(env)
        assert np.array_equal(q, o.i32), 'free-running: int state differs at call %d' % k
        with np.errstate(invalid='ignore'):
            err = np.where((f == o.f64) | (np.isnan(f) & np.isnan(o.f64)), 0, np.abs(f - o.f64))
        worst = max(worst, float(np.max(err)))
        assert worst <= TOL, (k, worst)
    upload(env, f0, q0)
    o.f64[:], o.i32[:] = f0, q0
    for k in range(steps):
        out = env.step(acts[k])
        o.step(acts[k], render=False)
        f, q = download(env)
        assert np.array_equal(q, o.i32), 'int state differs at call %d' % k
        with np.errstate(invalid='ignore'):
            err = np.where((f == o.f64) | (np.isnan(f) & np.isnan(o.f64)), 0, np.abs(f - o.f64))
        assert float(np.max(err)) <= 1e-9, (k, float(np.max(err)))
        assert np.array_equal(out.step_type.cpu().numpy(), o.step_type)
        assert helpers.same_or_nan(out.reward.cpu().numpy(), o.reward)
        o.f64[:], o.i32[:] = f, q
    assert np.array_equal(out.observation['image'].cpu().numpy(), o.render())
    env.close()


def test_make_disjoint_path_is_taken():
    """collisions.py:586-748 (_make_disjoint / _position_correction, the corner-corner fallback with the
    reference's quirks) is rare: count how often the step kernel takes it on the workloads whose full-size
    parity the tests above establish (the counter rides on the kernel's profiling outputs)."""
    for name, n, steps in (('colliding_predators_32', 4096, 30), ('falling_balls_64', 2048, 60)):
        env = make_env(name, n, seed=17, env_index0=0)
        env.reset()
        env.set_debug(128, 0)    # reward := work counters (state unaffected)
        rs = np.random.RandomState(9)
        total = 0
        for _ in range(steps):
            out = env.step(rs.randint(0, 5, size=n) if env._is_grid else rs.uniform(-1, 1, size=(n, 2)))
            r = out.reward.cpu().numpy()
            total += int(np.nansum(np.floor(np.nan_to_num(r) / 1e10)))
        print(name, 'make_disjoint calls in %d env-steps:' % (n * steps), total)
        assert total > 0, name
        env.close()


def test_deferred_fault_surfaces_on_the_next_call():
    """Default fault handling: no per-step synchronisation, but a device-side fault raised in one call is
    re-raised (with the reference's exception type) by the next call."""
    import collections
    from moog import action_spaces, environment, game_rules, observers, physics as physics_lib, sprite, tasks
    cfg = dict(   # three portals: portal.py:51-54 raises ValueError("even number of portals")
        state_initializer=lambda: collections.OrderedDict(
            [('portal', [sprite.Sprite(x=0.2 * k + 0.2, y=0.5, shape='square', scale=0.05) for k in range(3)]),
             ('agent', [sprite.Sprite(x=0.5, y=0.1, shape='circle', scale=0.04)])]),
        physics=physics_lib.Physics(updates_per_env_step=1),
        task=tasks.CompositeTask(),
        action_space=action_spaces.Grid(action_layers='agent'),
        observers={'image': observers.PILRenderer(image_size=(64, 64))},
        game_rules=(game_rules.Portal(teleporting_layer='agent', portal_layer='portal'),))
    env = environment.BatchedEnvironment(num_envs=8, **cfg)
    env.check_faults = False
    env.reset()                 # (the rule also runs inside reset; keep that fault for the step path)
    env.clear_faults()
    env.check_faults = True
    env.step(np.zeros(8, np.int32))          # raises the device fault, not yet visible to the host
    import torch
    torch.cuda.synchronize()
    with pytest.raises(ValueError):
        env.step(np.zeros(8, np.int32))
    env.clear_faults()
    env.check_faults = 'sync'
    with pytest.raises(ValueError):
        env.step(np.zeros(8, np.int32))


@pytest.mark.parametrize('name,size,n', [('colliding_predators_32', (256, 256), 64), ('functional_maze', (512, 256), 48),
                                         ('chase_avoid_torus', (192, 320), 48), ('first_person_predators_prey', (1024, 1024), 6),
                                         ('falling_balls_64', (320, 144), 32)])
def test_frames_larger_than_one_tile(name, size, n):
    """Canvases wider / taller than the 128 x 128 tile a workgroup renders (the reference's benchmark renders
    256, 512 and 1024 pixels, tests/runtime_benchmark.py:31-38; pacman renders 256): every tile of every frame
    against the oracle renderer over a few steps.  (image_size = (width, height) as in pil_renderer.py:64-66.)"""
    from moog import environment, observers
    from moog_demos import example_configs
    cfg = example_configs.load(name)
    old = cfg['observers']['image']
    cfg['observers'] = {'image': observers.PILRenderer(image_size=size, bg_color=old._bg_color, color_to_rgb=old.color_to_rgb,
                                                       polygon_modifier=old.polygon_modifier)}
    env = environment.BatchedEnvironment(num_envs=n, seed=3, env_index0=11, layer_capacity=example_configs.capacity(name), **cfg)
    assert tuple(env.image.shape[1:3]) == (size[1], size[0])
    o = helpers.OracleEnv(env.compiled, n_envs=n, seed=3, env_index0=11)
    env.reset()
    rs = np.random.RandomState(8)
    for k in range(4):
        a = rs.randint(0, 5, size=n) if env._is_grid else rs.uniform(-1, 1, size=(n, 2))
        out = env.step(a)
        o.f64[:], o.i32[:] = download(env)
        img = out.observation['image'].cpu().numpy()
        ref = o.render()
        bad = np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0]
        assert bad.size == 0, ('frames differ at step %d' % k, bad[:8].tolist(), int(bad.size))


@pytest.mark.gpu
def test_batched_meta_state():
    """ModifyMetaState (modify_meta_state.py:8-50) over a batch: every env owns a meta-state object, the rule runs
    once per env and call, and an env's object is re-initialised when that env auto-resets (environment.py:100-104)."""
    import collections
    from moog import action_spaces, environment, game_rules, observers, physics as physics_lib, sprite, tasks
    n = 6
    bump = game_rules.ModifyMetaState(lambda m: m.__setitem__('calls', m['calls'] + 1))
    cfg = dict(
        state_initializer=lambda: collections.OrderedDict(
            [('target', [sprite.Sprite(x=0.8, y=0.5, shape='square', scale=0.1, c0=64)]),
             ('agent', [sprite.Sprite(x=0.2, y=0.5, shape='circle', scale=0.1, c0=255)])]),
        physics=physics_lib.Physics(updates_per_env_step=1),
        task=tasks.CompositeTask(tasks.ContactReward(1., 'agent', 'target', reset_steps_after_contact=0),
                                 timeout_steps=50),
        action_space=action_spaces.SetPosition(action_layers='agent'),
        observers={'image': observers.PILRenderer(image_size=(64, 64))},
        game_rules=(bump,), meta_state_initializer=lambda: {'calls': 0})
    env = environment.BatchedEnvironment(num_envs=n, **cfg)
    env.reset()
    assert [m['calls'] for m in env.meta_state] == [1] * n      # rules step once inside reset (environment.py:92-94)
    far, hit = [0.2, 0.5], [0.8, 0.5]
    env.step(np.array([far] * n))
    assert [m['calls'] for m in env.meta_state] == [2] * n
    ts = env.step(np.array([hit if i % 2 else far for i in range(n)]))   # odd envs touch the target: episode over
    assert list(ts.step_type.cpu().numpy()) == [2 if i % 2 else 1 for i in range(n)]
    assert [m['calls'] for m in env.meta_state] == [3] * n
    env.step(np.array([far] * n))                                 # odd envs auto-reset: fresh object, one rule step
    assert [m['calls'] for m in env.meta_state] == [1 if i % 2 else 4 for i in range(n)]
    env.close()


@pytest.mark.gpu
def test_multi_agent_wrapper():
    """MultiAgentEnvironment (env_wrappers/multi_agent.py:12-52): the caller drives one key of the Composite action
    space, agent objects fill in the rest from the observation; same trajectory as stepping with the full dict."""
    import torch
    from moog import env_wrappers, environment
    from moog_demos import example_configs

    class Drift(object):
        def __init__(self, value):
            self.value, self.seen = value, 0

        def step(self, observation):
            assert 'image' in observation
            self.seen += 1
            return self.value
    n = 8
    joy = torch.full((n, 2), 0.25, dtype=torch.float64)
    move = torch.full((n,), 3, dtype=torch.int64)
    eye = torch.full((n, 2), 0.6, dtype=torch.float64)
    base = environment.BatchedEnvironment(num_envs=n, seed=4, **example_configs.load('actions_zoo'))
    other = environment.BatchedEnvironment(num_envs=n, seed=4, **example_configs.load('actions_zoo'))
    a1, a2 = Drift(move), Drift(eye)
    wrapped = env_wrappers.MultiAgentEnvironment(base, 'agent_0', agent_1=a1, eye=a2)
    wrapped.reset(); other.reset()
    for _ in range(5):
        ts = wrapped.step(joy)
        other.step({'agent_0': joy, 'agent_1': move, 'eye': eye})
    assert (a1.seen, a2.seen) == (5, 5) and ts.observation['image'].shape == (n, 64, 64, 3)
    f, q = download(base)
    f2, q2 = download(other)
    assert np.array_equal(f, f2, equal_nan=True) and np.array_equal(q, q2)
    assert wrapped.action_spec().keys() == base.action_spec().keys() and wrapped.step_count is not None


@pytest.mark.gpu
@pytest.mark.parametrize('name,levels', [
    ('chase_avoid_torus', [0, 1]), ('colliding_predators', [0]), ('falling_balls', [0]),
    ('first_person_predators_prey', [0]), ('functional_maze', [0]), ('multi_tracking_with_feature', [2, 3, 4]),
    ('match_to_sample', [2, 3, 4]), ('predators_arena', [1, 2, 3]), ('bounce_box_contact_prediction', [True, False]),
    ('red_green', [0, 1, 2, 3]), ('pacman', [0, 1]), ('parallelogram_catch', [0, 1, 2]), ('pong', [0]), ('cleanup', [0])])
def test_example_configs_run(name, levels):
    """The reference's own smoke test of its example configs (tests/moog_demos/example_configs/test_examples.py:
    52-64): every config, every level, two episodes of 200 random actions through the single-environment facade."""
    import importlib
    from moog import environment
    from moog_demos import example_configs
    module = importlib.import_module('moog_demos.example_configs.' + name)
    for level in levels:
        # (layers that rules append to have a fixed capacity on the engine; the reference's lists are unbounded)
        capacity, steps = example_configs.capacity(name), 200
        if name == 'first_person_predators_prey':
            # a predator appears every other step and a prey every fifth (:178-190) and they leave slowly: the
            # reference's lists just grow, the engine's layers are sized up front (and the rasteriser keeps a
            # frame's edges in LDS), so this config runs shorter episodes with room for every arrival
            capacity, steps = {'prey': 30, 'predators': 70}, 120
        env = environment.Environment(layer_capacity=capacity, **module.get_config(level))
        for _ in range(2):
            ts = env.reset()
            assert ts.first()
            for _ in range(steps):
                ts = env.step(action=env.action_space.random_action())
                assert ts.observation['image'].dtype == np.uint8
        env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('size,aa', [((64, 64), 2), ((64, 48), 3), ((32, 32), 4), ((48, 64), 5), ((16, 16), 8),
                                     ((32, 16), 16), ((512, 256), 2), ((272, 272), 3), ((1024, 64), 2),
                                     ((50, 37), 1), ((17, 9), 1), ((1000, 30), 1), ((131, 131), 1), ((30, 22), 3), ((17, 9), 2),
                                     ((33, 35), 7), ((201, 77), 2), ((1, 1), 1), ((5, 3), 16)])
def test_anti_aliasing_sweep(size, aa):
    """PILRenderer(anti_aliasing=aa) (pil_renderer.py:64-66,111-112) over scale factors 2..16 and frame shapes whose
    rows span one and several blocks of the resize kernels: canvas + both LANCZOS passes against the oracle's
    restatement of Pillow's resample (itself pinned by tests/golden/resize.npz).  Any size is accepted, as by the
    reference: a canvas whose width is no multiple of 16 is drawn 16-aligned and cropped."""
    from moog import environment, observers
    from moog_demos import example_configs
    cfg = example_configs.load('colliding_predators_32')
    old = cfg['observers']['image']
    cfg['observers'] = {'image': observers.PILRenderer(image_size=size, anti_aliasing=aa, bg_color=old._bg_color,
                                                       color_to_rgb=old.color_to_rgb)}
    n = 6
    env = environment.BatchedEnvironment(num_envs=n, seed=5, **cfg)
    o = helpers.OracleEnv(env.compiled, n_envs=n, seed=5)
    env.reset()
    rs = np.random.RandomState(2)
    for k in range(2):
        out = env.step(rs.uniform(-1, 1, size=(n, 2)))
        o.f64[:], o.i32[:] = download(env)
        img = out.observation['image'].cpu().numpy()
        assert img.shape == (n, size[1], size[0], 3)
        assert np.array_equal(img, o.render()), 'frames differ at step %d' % k
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name,rows,compact', [
    ('colliding_predators_32', None, None), ('colliding_predators_32', 64, None), ('pong', None, None), ('colliding_predators', None, None),
    ('falling_balls', 40, None), ('falling_balls_64', None, None), ('rules_zoo', 64, None), ('lambda_zoo', None, None), ('functional_maze', None, None),
    ('functional_maze', 128, None), ('cleanup', None, None), ('match_to_sample_l3', None, None), ('predators_arena_l2', None, None),
    ('parallelogram_catch', None, None), ('multi_tracking_with_feature_l1', None, None), ('chase_avoid_torus', None, None),
    ('chase_avoid_torus', 64, None), ('aa_zoo', None, None),
    # the edge records in their 4-byte form (RmEdgesCompact: what the engine picks by itself when the 16-byte records keep frames
    # off a CU -- falling_balls_64, first_person_predators_prey) and in their 16-byte form where it would pick the other
    ('colliding_predators_32', None, 1), ('colliding_predators_32', 64, 1), ('falling_balls_64', None, 0), ('match_to_sample_l3', None, 1),
    ('chase_avoid_torus', 64, 1), ('functional_maze', 128, 1), ('cleanup', None, 1), ('first_person_predators_prey', None, 0)])
def test_mask_rasteriser_matches_the_span_rasteriser(name, rows, compact, monkeypatch):
    """One-tile frames of polygons with <= 128 vertices are drawn by the mask rasteriser (csrc/moog_raster_mask_core.h: no
    crossing lists, census by bit mask); MOOG_RASTER_MASK=0 selects the push / sort / span kernel for every frame.  Both
    must give the same frames, bit for bit -- frames that come with a step, frames of uploaded state, frames after resets
    -- also when the row records are capped so that frames take several passes."""
    n = 192
    monkeypatch.setenv('MOOG_RASTER_MASK', '0')
    ref = make_env(name, n, seed=31, env_index0=17)
    assert ref.raster_path() == 'spans'
    monkeypatch.setenv('MOOG_RASTER_MASK', '1')
    if rows is not None:
        monkeypatch.setenv('MOOG_RASTER_ROWS', str(rows))
    if compact is not None:
        monkeypatch.setenv('MOOG_RASTER_COMPACT', str(compact))
    env = make_env(name, n, seed=31, env_index0=17)
    if env.raster_path() != 'mask':
        pytest.skip('the program keeps the span kernel (multi-tile frames)')
    a0 = ref.reset().observation['image'].cpu().numpy()
    a1 = env.reset().observation['image'].cpu().numpy()
    assert np.array_equal(a0, a1), 'frames of the reset differ'
    rs = np.random.RandomState(4)
    for k in range(14):
        a = env.random_action()
        i0 = ref.step(a).observation['image'].cpu().numpy()
        i1 = env.step(a).observation['image'].cpu().numpy()
        bad = np.nonzero((i0 != i1).reshape(n, -1).any(axis=1))[0]
        assert bad.size == 0, ('frames differ at step %d' % k, bad[:8].tolist(), int(bad.size))
        f0, q0 = download(ref)
        f1, q1 = download(env)
        assert np.array_equal(q0, q1) and np.array_equal(f0, f1, equal_nan=True)
        o1 = env.observation()['image'].cpu().numpy()   # the same frame from the records
        assert np.array_equal(o1, i1), 'frame from the records differs from the frame of the step'


def _draw_record_parts(rec, S):
    """The bytes of a draw record that count (csrc/moog_draw_record.h): header, the S items, the n_pts points and owner bytes."""
    hdr = rec[:16].view(np.int32)
    n_pts = int(hdr[0])
    items = rec[16:16 + 16 * S]
    o_pts = 16 + 16 * S
    # (o_owner = o_pts + 4 x point capacity: the capacity follows from the stride)
    return hdr.copy(), items.copy(), n_pts, o_pts


@pytest.mark.parametrize('name', ['colliding_predators_32', 'chase_avoid_torus', 'functional_maze', 'falling_balls_64', 'parallelogram_catch',
                                  'first_person_predators_prey'])
def test_draw_records_of_the_step_kernel_equal_the_derived_ones(name):
    """The frame's draw record (csrc/moog_draw_record.h) is written twice over: by the step kernel, from the record it holds in
    LDS (moog_engine_step; for late-reset programs also by the reset kernel behind it), and by the derive kernel from the stored
    state (moog_engine_render).  Same function, two homes: header, items, points and owners must be equal byte for byte, call
    after call, across auto-resets -- and the host model (tests/test_raster_mask_model.py) runs that function on the CPU."""
    n = 256
    env = make_env(name, n, seed=12)
    if env.raster_path() != 'mask':
        pytest.skip('no draw records: the span kernel draws this program')
    P = env.compiled.program
    ncopy = 9 if P.render.polymod == 1 else 1
    S = int(P.n_slots) * ncopy
    cap = int(env.layout.TOTV) * ncopy
    env.reset()
    compared = 0
    for k in range(24):
        env.step(env.random_action())
        a, in_step = env.draw_records()
        if not in_step:
            pytest.skip('the step kernel does not write this program\'s draw records')
        env.observation()                      # the derive kernel, over the state the step launch stored
        b, _ = env.draw_records()
        for i in range(n):
            ha, ia, na, o_pts = _draw_record_parts(a[i], S)
            hb, ib, nb, _ = _draw_record_parts(b[i], S)
            assert np.array_equal(ha, hb), (k, i, ha, hb)
            assert np.array_equal(ia, ib), ('items differ', k, i)
            assert np.array_equal(a[i, o_pts:o_pts + 4 * na], b[i, o_pts:o_pts + 4 * nb]), ('points differ', k, i)
            o_own = o_pts + 4 * cap
            assert np.array_equal(a[i, o_own:o_own + na], b[i, o_own:o_own + nb]), ('owner bytes differ', k, i)
            compared += na
    assert compared > 0


def test_raster_path_by_program():
    """Which rasteriser a program's frames take (moog_engine_raster_path): the mask rasteriser for one-tile frames of
    polygons with <= 128 vertices (the 102-vertex annuli take its cooperative row routine), the nine copies per sprite of a
    torus included; multi-tile frames (pacman 256 x 256) and frames whose tables outgrow 64 KB of LDS keep the span kernel."""
    for name, want in (('colliding_predators_32', 'mask'), ('functional_maze', 'mask'), ('falling_balls_64', 'mask'),
                       ('pacman', 'spans'), ('chase_avoid_torus', 'mask'), ('match_to_sample_l3', 'mask'),
                       ('first_person_predators_prey', 'mask')):
        env = make_env(name, 4, seed=1)
        assert env.raster_path() == want, name
        # the edge records' form: 4 bytes where 16 would keep frames off a CU (falling_balls_64: 1816 vertex slots), 16 elsewhere
        # (the headline workload's rows phase is a tenth cheaper with them)
        if name in ('falling_balls_64', 'chase_avoid_torus'):
            assert env.raster_compact_edges(), name
        if name in ('colliding_predators_32', 'functional_maze'):
            assert not env.raster_compact_edges(), name
        env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name,n,pre,steps', [('falling_balls_64', 1024, 30, 12), ('colliding_predators_32', 512, 5, 10)])
def test_sprites_without_a_finite_vertex(name, n, pre, steps):
    """A sprite whose position went NaN has no finite vertex: for the reference it overlaps every sprite (matplotlib's empty
    path) while a collision with it changes nothing.  The engine leaves such pairs out of the collision candidates
    (moog_device.h broad_pair); the oracle runs the reference's full path.  Non-finite sprites are planted in every third env
    (a ball / predator with NaN position and vertices; in some envs two of them), then engine and oracle run in lock step:
    integer records exact, floats <= 1e-9 (NaN where the oracle has NaN), rewards / step types exact, frames bit-exact."""
    env = make_env(name, n, seed=77, env_index0=5)
    o = helpers.OracleEnv(env.compiled, n_envs=n, seed=77, env_index0=5)
    env.reset()
    rs = np.random.RandomState(3)
    act = (lambda: rs.randint(0, 5, size=n)) if env._is_grid else (lambda: rs.uniform(-1, 1, size=(n, 2)))
    for _ in range(pre):
        env.step(act())
    f, q = download(env)
    L, P = env.layout, env.compiled.program
    planted = 0
    for i in range(0, n, 3):
        alive = [s for s in range(4, L.S) if q[i, L.o_flags + s] & 1]
        for s in rs.choice(alive, size=min(len(alive), 1 + (i // 3) % 2), replace=False):
            f[i, L.o_pos + 2 * s:L.o_pos + 2 * s + 2] = np.nan
            v0 = L.o_verts + 2 * P.slot_voff[s]
            f[i, v0:v0 + 2 * int(q[i, L.o_nverts + s])] = np.nan
            planted += 1
    assert planted > n // 4
    upload(env, f, q)
    o.f64[:], o.i32[:] = f, q
    for k in range(steps):
        a = act()
        out = env.step(a)
        o.step(a, render=False)
        f, q = download(env)
        assert np.array_equal(q, o.i32), 'int state differs at step %d' % k
        assert np.array_equal(np.isnan(f), np.isnan(o.f64)), 'NaN pattern differs at step %d' % k
        with np.errstate(invalid='ignore'):
            err = np.where((f == o.f64) | (np.isnan(f) & np.isnan(o.f64)), 0, np.abs(f - o.f64))
        assert float(np.max(err)) <= 1e-9, (k, float(np.max(err)))
        assert np.array_equal(out.step_type.cpu().numpy(), o.step_type)
        assert helpers.same_or_nan(out.reward.cpu().numpy(), o.reward)
        o.f64[:], o.i32[:] = f, q
        assert np.array_equal(out.observation['image'].cpu().numpy(), o.render()), 'frames differ at step %d' % k
    env.close()


@pytest.mark.gpu
def test_filters_over_big_layers_lane_parallel():
    """Filters that only read their own sprite are evaluated for 64 sprites of a layer at once, one per lane, when the
    rule ranges over >= 32 slots (MOOG_FILTER_EXPR_LANES, eval_expr_t<true>): VanishByFilter, ChangeLayer and ModifySprites
    on layers of 48 and 70 slots (more than one pass of 64 lanes), dtype-sensitive filters (float32 velocities), list
    compaction after the vanishes -- engine against the oracle (which walks the sprites one by one) in lock step."""
    import collections
    from moog import action_spaces, environment, game_rules as gr, observers, physics as physics_lib, sprite, tasks, _abi
    from moog.state_initialization import distributions as distribs, sprite_generators

    def get_config():
        dots = distribs.Product(
            [distribs.Continuous('x', 0.05, 0.95), distribs.Continuous('y', 0.05, 0.95),
             distribs.Continuous('x_vel', -0.03, 0.03), distribs.Continuous('y_vel', -0.03, 0.03),
             distribs.Continuous('c0', 0., 1.), distribs.Discrete('opacity', [255, 128])],
            shape='triangle', scale=0.03, c1=1., c2=1.)
        make_dots = sprite_generators.generate_sprites(dots, num_sprites=60)
        make_more = sprite_generators.generate_sprites(dots, num_sprites=40)

        def state_initializer():
            return collections.OrderedDict([('dots', make_dots()), ('more', make_more()), ('bin', []),
                                            ('agent', [sprite.Sprite(x=0.5, y=0.5, shape='square', scale=0.05, c0=0.3, c1=1., c2=1.)])])

        def leaving(s):
            low = (s.position < 0.1) * (s.velocity < 0.)
            high = (s.position > 0.9) * (s.velocity > 0.)
            return any(low) or any(high)

        def faint(s):
            return s.opacity < 200 and s.c0 > 0.5 and np.abs(s.x_vel) + np.abs(s.y_vel) > 0.02

        def brake(s):
            s.velocity = s.velocity * 0.5
            s.c2 = 0.5

        rules = (gr.VanishByFilter('dots', leaving), gr.ChangeLayer('more', 'bin', filter_fn=leaving),
                 gr.ModifySprites(('dots', 'more'), brake, filter_fn=faint))
        return {
            'state_initializer': state_initializer,
            'physics': physics_lib.Physics((physics_lib.Drag(coeff_friction=0.01), ['dots', 'more']), updates_per_env_step=2),
            'task': tasks.CompositeTask(tasks.ContactReward(1, layers_0='agent', layers_1=('dots', 'more')), timeout_steps=25),
            'action_space': action_spaces.Joystick(scaling_factor=0.02, action_layers='agent'),
            'observers': {'image': observers.PILRenderer(image_size=(64, 64), color_to_rgb='hsv_to_rgb')},
            'game_rules': rules,
        }
    n = 48
    env = environment.BatchedEnvironment(num_envs=n, seed=21, env_index0=300, layer_capacity={'dots': 70, 'more': 48, 'bin': 48},
                                         **get_config())
    P = env.compiled.program
    assert int(P.xstack_depth) > 0 and [P.rules[r].filter for r in range(3)] == [_abi.MOOG_FILTER_EXPR_LANES] * 3
    o = helpers.OracleEnv(env.compiled, n_envs=n, seed=21, env_index0=300)
    env.reset()
    o.reset(render=False)
    rs = np.random.RandomState(4)
    for k in range(60):
        a = rs.uniform(-1, 1, size=(n, 2))
        out = env.step(a)
        o.step(a, render=False)
        f, q = download(env)
        assert np.array_equal(q, o.i32), 'int state differs at step %d' % k
        with np.errstate(invalid='ignore'):
            err = np.where(f == o.f64, 0, np.abs(f - o.f64))
        assert float(np.nanmax(err)) <= 1e-9, (k, float(np.nanmax(err)))
        assert np.array_equal(out.step_type.cpu().numpy(), o.step_type)
        o.f64[:], o.i32[:] = f, q
    assert np.array_equal(out.observation['image'].cpu().numpy(), o.render())
    env.raise_faults()
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name,n,steps', [('colliding_predators_32', 512, 30), ('falling_balls_64', 128, 20), ('functional_maze', 256, 30),
                                          ('cleanup', 128, 40), ('pacman', 32, 30)])
def test_specialised_step_kernel_is_result_neutral(name, n, steps, monkeypatch, tmp_path):
    """moog/_spec.py compiles the step kernel with the program as a compile-time constant (csrc/moog_step_spec.hip): same
    source, same arithmetic -- records, time steps and frames of every call equal the generic kernel's bit for bit, across
    auto-resets, for a plain program, a program of the variant with the expression evaluator and one of the variant with
    every component.  The engine says which kernel it uses; a kernel built for another program is not picked up."""
    from moog import _spec
    monkeypatch.setenv('MOOG_SPEC_DIR', str(tmp_path))
    ref = make_env(name, n, seed=6, env_index0=3)
    assert ref.step_kernel() == 'generic'
    path = _spec.build(ref.compiled.program)
    assert os.path.dirname(path) == str(tmp_path)
    env = make_env(name, n, seed=6, env_index0=3)
    assert env.step_kernel() == 'specialised'
    other = make_env('pong', 8, seed=1)
    assert other.step_kernel() == 'generic'
    other.close()
    t0, t1 = ref.reset(), env.reset()
    assert np.array_equal(t0.observation['image'].cpu().numpy(), t1.observation['image'].cpu().numpy())
    for k in range(steps):
        a = ref.random_action()
        t0, t1 = ref.step(a), env.step(a)
        f0, q0 = download(ref)
        f1, q1 = download(env)
        assert np.array_equal(q0, q1), 'integer records differ at call %d' % k
        assert np.array_equal(f0, f1, equal_nan=True), 'float records differ at call %d' % k
        assert np.array_equal(t0.step_type.cpu().numpy(), t1.step_type.cpu().numpy())
        assert np.array_equal(t0.reward.cpu().numpy(), t1.reward.cpu().numpy(), equal_nan=True)
        assert np.array_equal(t0.observation['image'].cpu().numpy(), t1.observation['image'].cpu().numpy())
    env.raise_faults()


@pytest.mark.gpu
def test_runtime_benchmark_reports_every_phase(capsys):
    """the batched counterpart of the reference's tests/runtime_benchmark.py:64-157 runs -- pong with one env and
    with a batch -- and reports its five phases (step + render, step without render, reset, physics only, render only) and the
    six renderer settings of runtime_benchmark.py:31-38 (64 .. 1024 pixels, anti_aliasing 1 / 2), every figure a positive time."""
    from moog_demos import runtime_benchmark
    phases = ('step + render', 'step, no render', 'reset only', 'physics only', 'render only')
    r1 = runtime_benchmark.main(['--config', 'pong', '--num_envs', '1', '--reps', '5'])
    assert tuple(r1['phases']) == phases and all(v > 0 for v in r1['phases'].values()) and r1['render'] == {}
    r2 = runtime_benchmark.main(['--config', 'pong', '--num_envs', '4096', '--reps', '5', '--render_sizes', '--render_envs', '16'])
    assert tuple(r2['phases']) == phases and all(v > 0 for v in r2['phases'].values())
    assert sorted(r2['render']) == [(64, 1), (128, 1), (256, 1), (512, 1), (512, 2), (1024, 1)]
    assert all(v > 0 for v in r2['render'].values())
    out = capsys.readouterr().out
    assert out.count('render only,') == 6 and 'pong: 4096 envs' in out and 'pong: 1 envs' in out
    # a batch amortises the launches: 4096 envs per call cost far less than 4096 calls of one env
    assert r2['phases']['step + render'] < 200 * r1['phases']['step + render']


@pytest.mark.gpu
def test_specialised_kernel_of_another_build_is_refused(monkeypatch, tmp_path, capfd):
    """A specialised step kernel is tied to the kernel sources and flags it was built from (moog/_digest.py): an object that
    carries another digest -- same ABI number, same program, same file name -- is reported and left alone, the engine steps
    with the generic kernels; moog._spec.build() replaces it and remove_stale() deletes it."""
    from moog import _digest, _engine, _spec
    monkeypatch.setenv('MOOG_SPEC_DIR', str(tmp_path))
    assert '%016x' % _engine.load_library().moog_source_digest() == _digest.source_digest() == _digest.digest_of(_engine.LIB_PATH)
    probe = make_env('pong', 8, seed=1)
    path = _spec.build(probe.compiled.program, digest='00000000deadbeef')
    assert _digest.digest_of(path) == '00000000deadbeef' and not _spec.is_current(path)
    env = make_env('pong', 8, seed=1)
    assert env.step_kernel() == 'generic'
    assert 'built from other kernel sources' in capfd.readouterr().err
    env.close()
    assert _spec.build(probe.compiled.program) == path and _spec.is_current(path)
    env = make_env('pong', 8, seed=1)
    assert env.step_kernel() == 'specialised'
    env.close()
    _spec.build(probe.compiled.program, force=True, digest='00000000deadbeef')
    assert _spec.remove_stale() == [os.path.basename(path)] and not os.path.exists(path)


@pytest.mark.gpu
@pytest.mark.parametrize('rows', [None, 64])
def test_long_polygons_planted_vs_oracle(rows, monkeypatch):
    """Polygons of 33 .. 102 vertices (annuli, combs, lattices, stars, scatter, heads only: tests/test_raster_mask_model.py
    long_polygon) planted in the long slots of match_to_sample's records on the device, drawn by the mask rasteriser's long-polygon
    rows (rm_p4_big: 128-bit edge words, indexed edge lists, LDS atomics) and compared with the oracle renderer; also with the row
    records capped (several passes, the LDS-free sort)."""
    import torch
    from test_raster_mask_model import long_polygon
    if rows is not None:
        monkeypatch.setenv('MOOG_RASTER_ROWS', str(rows))
    name, n = 'match_to_sample_l3', 480
    env = make_env(name, n, seed=9)
    assert env.raster_path() == 'mask'
    env.reset()
    f, q = download(env)
    P, L = env.compiled.program, env.compiled.layout
    W = P.render.width
    big = [s for s in range(P.n_slots) if P.slot_vcap[s] > 32]
    assert big
    rs = np.random.RandomState(177)
    for e in range(n):
        for s in big:
            nv = int(rs.randint(33, P.slot_vcap[s] + 1))
            pts = long_polygon(rs, W, nv, (e + s) % 6)
            wv = (pts + np.where(pts >= 0, 0.5, -0.5)) / float(W)   # world coordinates whose scaled (int) is the wanted point
            v0 = L.o_verts + 2 * int(P.slot_voff[s])
            f[e, v0:v0 + 2 * nv] = wv.reshape(-1)
            q[e, L.o_nverts + s] = nv
            q[e, L.o_flags + s] |= 1
            q[e, L.o_opacity + s] = (255, 128)[e % 2]
    upload(env, f, q)
    o = helpers.OracleEnv(env.compiled, n_envs=n, seed=9)
    o.f64[:], o.i32[:] = f, q
    img = env.observation()['image'].cpu().numpy()
    ref = o.render()
    bad = np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0]
    assert bad.size == 0, ('frames differ', bad[:8].tolist(), int(bad.size))


@pytest.mark.gpu
def test_torus_copies_planted_vs_oracle():
    """Torus frames with the sprites moved to where the nine copies matter (across edges and corners, far outside, blown up beyond
    the canvas so that copies overlap; every vertex of some sprites NaN / infinite / beyond int): the mask rasteriser's torus phases
    (64-bit LDS atomics on order-preserving keys, visible copies only) against the oracle renderer."""
    name, n = 'chase_avoid_torus', 512
    env = make_env(name, n, seed=3)
    assert env.raster_path() == 'mask'
    env.reset()
    env.step(env.random_action())
    f, q = download(env)
    P, L = env.compiled.program, env.compiled.layout
    rs = np.random.RandomState(8)
    for e in range(n):
        for s in range(P.n_slots):
            v0, nv = L.o_verts + 2 * int(P.slot_voff[s]), int(q[e, L.o_nverts + s])
            if nv <= 0:
                continue
            v = f[e, v0:v0 + 2 * nv].reshape(nv, 2)
            ctr = v.mean(axis=0)
            kind = (e + s) % 8
            scale = (1.0, 1.0, 2.5, 1.0, 9.0, 30.0, 0.3, 1.0)[kind]
            target = ((0.0, rs.rand()), (0.0, 0.0), (rs.rand(), 1.0), (1.0, 1.0), rs.rand(2), rs.rand(2), (0.999, 0.001),
                      rs.uniform(-1.3, 2.3, size=2))[kind]
            v[:] = (v - ctr) * scale + np.asarray(target, np.float64)
            if e % 9 == 4 and s % 2 == 0:
                v[:, rs.randint(2)] = (np.nan, 3.0e9, -7.0e11, np.inf)[(e // 9 + s) % 4]
    upload(env, f, q)
    o = helpers.OracleEnv(env.compiled, n_envs=n, seed=3)
    o.f64[:], o.i32[:] = f, q
    img = env.observation()['image'].cpu().numpy()
    ref = o.render()
    bad = np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0]
    assert bad.size == 0, ('frames differ', bad[:8].tolist(), int(bad.size))


@pytest.mark.gpu
def test_fit_layer_capacity_is_result_neutral():
    """fit_layer_capacity() re-creates the engine with the appendable layers sized by their high-water marks (a smaller record:
    more envs per CU): time steps and frames of every later call equal those of the engine that keeps its roomy layers, the
    layers still grow on demand afterwards, and the sprites are the same layer by layer."""
    import torch
    from moog import environment
    from moog_demos import example_configs
    name, n = 'first_person_predators_prey', 256
    cfg = example_configs.load(name)
    big = environment.BatchedEnvironment(num_envs=n, seed=5, layer_capacity={'prey': 32, 'predators': 96}, **cfg)
    fit = environment.BatchedEnvironment(num_envs=n, seed=5, layer_capacity={'prey': 32, 'predators': 96}, **cfg)
    big.reset()
    fit.reset()
    g = torch.Generator(device='cpu').manual_seed(3)
    for k in range(90):   # (the roomy layers themselves overflow after ~150 calls of a random policy)
        if k == 30:
            before = fit.layout.f64_per_env
            caps = fit.fit_layer_capacity(headroom=0.1, min_room=1)   # tight: the layers have to grow again later in the run
            assert caps and fit.layout.f64_per_env < before, (caps, before, fit.layout.f64_per_env)
        a = torch.rand((n, 2), generator=g, dtype=torch.float64) * 2 - 1
        x, y = big.step(a), fit.step(a)
        assert torch.equal(x.step_type, y.step_type), k
        assert torch.equal(torch.nan_to_num(x.reward, nan=-7.), torch.nan_to_num(y.reward, nan=-7.)), k
        assert torch.equal(x.observation['image'], y.observation['image']), 'frames differ at call %d' % k
    assert len(fit.capacity_growths) >= 2, fit.capacity_growths   # the fit, then at least one growth on demand
    use = fit.layer_usage()
    assert all(u['dropped'] == 0 for u in use.values()), use
    for env_i in (0, n // 2, n - 1):
        sa, sb = big.sprites(env_i), fit.sprites(env_i)
        for layer in sa:
            assert len(sa[layer]) == len(sb[layer]), layer
            for p, q in zip(sa[layer], sb[layer]):
                assert p['x'] == q['x'] and p['y'] == q['y'] and np.array_equal(p['vertices'], q['vertices'])
    big.close()
    fit.close()


def test_auto_capacity_fits_by_itself():
    """layer_capacity={'auto': True, ...}: the layers are fitted to the batch's high-water marks after `fit_after` calls without
    anybody asking (a smaller record from then on), keep growing on demand, and nothing observable changes: time steps and
    frames equal those of the engine with fixed roomy layers."""
    import torch
    from moog import environment
    from moog_demos import example_configs
    name, n = 'first_person_predators_prey', 128
    cfg = example_configs.load(name)
    big = environment.BatchedEnvironment(num_envs=n, seed=7, layer_capacity={'prey': 32, 'predators': 96}, **cfg)
    auto = environment.BatchedEnvironment(num_envs=n, seed=7, layer_capacity={'auto': True, 'fit_after': 25, 'prey': 32, 'predators': 96}, **cfg)
    big.reset()
    auto.reset()
    before = auto.layout.f64_per_env
    g = torch.Generator(device='cpu').manual_seed(4)
    for k in range(60):
        a = torch.rand((n, 2), generator=g, dtype=torch.float64) * 2 - 1
        x, y = big.step(a), auto.step(a)
        assert torch.equal(x.step_type, y.step_type), k
        assert torch.equal(x.observation['image'], y.observation['image']), 'frames differ at call %d' % k
        assert (auto.layout.f64_per_env < before) == (k >= 24), (k, auto.layout.f64_per_env, before)
    assert auto.capacity_growths, 'the automatic fit did not happen'
    assert all(u['dropped'] == 0 for u in auto.layer_usage().values())
    big.close()
    auto.close()
