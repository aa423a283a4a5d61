"""Differential fuzz: randomly assembled configs (layers, shapes, forces, corrective physics,
rules incl. run-time sprite creation and traced lambdas, tasks, action spaces) stepped by the
HIP engine and by the CPU oracle on the same Philox streams.  Every feature exists twice (C and
HIP); this looks for divergence in combinations the recorded reference fixtures do not cover.
Integer records bit-exact, floats <= 1e-9, rewards / step types / final frames identical."""
import collections

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

SHAPES = ['circle', 'square', 'triangle', 'pentagon', 'star_5', 'spoke_4', 'hexagon']


def random_config(seed):
    from moog import action_spaces, game_rules, observers, physics as physics_lib, shapes, tasks
    from moog.state_initialization import distributions as distribs, sprite_generators
    rs = np.random.RandomState(seed)

    def factors(lo, hi, moving, color):
        parts = [distribs.Continuous('x', lo, hi), distribs.Continuous('y', lo, hi),
                 distribs.Discrete('shape', list(rs.choice(SHAPES, size=rs.randint(1, 4), replace=False))),
                 distribs.Continuous('scale', 0.05, 0.11)]
        if moving:
            parts += [distribs.Continuous('x_vel', -0.03, 0.03), distribs.Continuous('y_vel', -0.03, 0.03)]
        if rs.rand() < 0.4:
            parts.append(distribs.Continuous('angle', 0., 6.28))
        if rs.rand() < 0.3:
            parts.append(distribs.Continuous('angle_vel', -0.1, 0.1))
        return distribs.Product(parts, c0=color, c1=1., c2=float(rs.uniform(0.4, 1.0)),
                                mass=float(rs.choice([1., 2., 0.5])))
    n_a, n_b = int(rs.randint(2, 6)), int(rs.randint(1, 5))
    rs2 = np.random.RandomState(7000 + seed)   # (later additions draw from their own stream: the older choices stay put)
    give_up = bool(rs2.rand() < 0.4)           # generators that give up after a few rejections and return what they have
    host_draws = bool(rs2.rand() < 0.5)        # sprites computed by the initializer from its own np.random draws
    n_extra, corners = int(rs2.randint(1, 4)), int(rs2.randint(3, 7))
    gen_a = sprite_generators.generate_sprites(factors(0.15, 0.85, True, 0.05), num_sprites=n_a,
                                               **(dict(max_recursion_depth=int(rs2.randint(1, 4)), fail_gracefully=True)
                                                  if give_up else {}))
    gen_b = sprite_generators.generate_sprites(factors(0.2, 0.8, bool(rs.rand() < 0.5), 0.6), num_sprites=n_b)
    gen_new = sprite_generators.generate_sprites(factors(0.2, 0.8, True, 0.8), num_sprites=1)
    gen_agent = sprite_generators.generate_sprites(
        distribs.Product([distribs.Continuous('x', 0.3, 0.7), distribs.Continuous('y', 0.3, 0.7)],
                         shape='circle', scale=0.07, c0=0.33, c1=1., c2=0.66), num_sprites=1)
    thickness = float(rs.choice([0.02, 0.05]))
    disjoint = bool(rs.rand() < 0.5)

    def state_initializer():
        walls = shapes.border_walls(visible_thickness=thickness, c0=0., c1=0., c2=0.5)
        agent = gen_agent(without_overlapping=walls)
        a = gen_a(disjoint=disjoint, without_overlapping=walls + agent)
        b = gen_b(without_overlapping=walls)
        extra = []
        if host_draws:   # as parallelogram_catch.py:34-68 / multi_tracking_with_feature.py:136-141 do
            from moog import sprite
            turn = np.random.uniform(0, 2)
            angles = np.pi * (2. * np.arange(corners) / corners + turn)
            outline = np.stack((np.sin(angles), np.cos(angles)), axis=1)
            outline *= np.array([[1.] if k % 2 else [np.random.uniform(0.5, 1.)] for k in range(corners)])
            centres = 0.3 * outline[:n_extra] + np.array([0.5, 0.5])
            flips = 0.5 * np.pi * np.random.binomial(1, 0.5, (n_extra))
            for c, f in zip(centres, flips):
                extra.append(sprite.Sprite(x=c[0], y=c[1], shape=0.06 * outline, angle=f, c0=0.45, c1=1., c2=0.9,
                                           x_vel=np.random.uniform(-0.02, 0.02), y_vel=0.01 * np.cos(f)))
            extra.append(sprite.Sprite(x=extra[0].x, y=1. - extra[0].y, shape='triangle', scale=0.05, c0=0.5, c1=1.,
                                       c2=0.5, x_vel=-1 * extra[0].x_vel))
        return collections.OrderedDict([('walls', walls), ('a', a), ('b', b), ('spawn', []), ('extra', extra),
                                        ('agent', agent)])

    forces = [(physics_lib.Drag(coeff_friction=float(rs.choice([0.05, 0.25]))), 'agent')]
    pairs = [('a', 'walls'), ('agent', 'walls'), ('a', 'a'), ('a', 'b'), ('agent', 'b'), ('b', 'walls'),
             (['a', 'spawn'], 'walls')]
    for i in rs.choice(len(pairs), size=rs.randint(2, 5), replace=False):
        la, lb = pairs[i]
        forces.append((physics_lib.Collision(elasticity=float(rs.choice([0., 0.5, 1.])),
                                             symmetric=bool(la == lb or rs.rand() < 0.3),
                                             update_angle_vel=bool(rs.rand() < 0.6)), la, lb))
    if rs.rand() < 0.3:
        forces.append((physics_lib.DownGravity(g=-0.001), 'a'))
    if rs.rand() < 0.3:
        forces.append((physics_lib.RandomForce(max_force_magnitude=0.01), 'b'))
    if rs.rand() < 0.3:
        forces.append((physics_lib.DistanceForce(physics_lib.linear_force_fn(
            zero_intercept=-0.002, slope=0.0005)), 'agent', 'a'))
    corrective = []
    c = rs.rand()
    if c < 0.2:
        corrective.append(physics_lib.ConstantSpeed('a', speed=0.02))
    elif c < 0.4:
        corrective.append(physics_lib.Tether('b', update_angle_vel=bool(rs.rand() < 0.5)))
    physics = physics_lib.Physics(*forces, updates_per_env_step=int(rs.choice([1, 3, 5])),
                                  corrective_physics=corrective)

    rules = []
    if rs.rand() < 0.5:
        rules.append(game_rules.VanishOnContact(vanishing_layer='b', contacting_layer='agent'))
    if rs.rand() < 0.4:
        rules.append(game_rules.ConditionalRule(
            condition=lambda state: np.random.binomial(1, p=0.3),
            rules=game_rules.CreateSprites('spawn', gen_new, without_overlapping=('walls', 'agent'))))
    if rs.rand() < 0.4:
        rules.append(game_rules.TimedRule((3, 5), game_rules.VanishByFilter('spawn', lambda s: s.x > 0.5)))
    if rs.rand() < 0.4:
        def _dim(s):
            s.c2 = s.c2 * 0.9
        rules.append(game_rules.ModifySprites(['a'], _dim, sample_one=bool(rs.rand() < 0.5),
                                              filter_fn=lambda s: s.c2 > 0.5))
    if rs.rand() < 0.4:
        def _kick(s):
            s.velocity = s.velocity * 0.5
        rules.append(game_rules.ModifyOnContact('a', 'agent', modifier_0=_kick))
    if rs.rand() < 0.3:
        rules.append(game_rules.ModifySprites(
            ['a', 'b'], lambda s: setattr(s, 'position', np.remainder(s.position, 1))))

    subtasks = [tasks.ContactReward(1., layers_0='agent', layers_1='b')]
    if rs.rand() < 0.5:
        subtasks.append(tasks.ContactReward(lambda a, s: -1. * s.scale, layers_0='agent', layers_1='a',
                                            condition=lambda a, s: s.mass > 0.75,
                                            reset_steps_after_contact=int(rs.choice([0, 2]))))
    if rs.rand() < 0.3:
        subtasks.append(tasks.StayAlive(reward_period=3, reward_value=0.5))
    task = tasks.CompositeTask(*subtasks, timeout_steps=int(rs.choice([6, 12])))
    if rs.rand() < 0.7:
        action_space = action_spaces.Joystick(scaling_factor=0.02, action_layers='agent')
    else:
        action_space = action_spaces.Grid(scaling_factor=0.02, action_layers='agent',
                                          control_velocity=bool(rs.rand() < 0.5))
    color = 'hsv_to_rgb' if rs.rand() < 0.7 else None
    return dict(state_initializer=state_initializer, physics=physics, task=task, action_space=action_space,
                observers={'image': observers.PILRenderer(image_size=(64, 64), color_to_rgb=color)},
                game_rules=tuple(rules))


@pytest.mark.parametrize('seed', range(64))
def test_random_config_engine_vs_oracle(seed):
    from moog import environment
    n, steps = 48, 25
    env = environment.BatchedEnvironment(num_envs=n, seed=100 + seed, env_index0=7 * seed,
                                         **random_config(seed))
    env.check_faults = False
    o = helpers.OracleEnv(env.compiled, n_envs=n, seed=100 + seed, env_index0=7 * seed)
    env.reset()
    o.reset(render=False)
    rs = np.random.RandomState(seed)
    import torch
    for k in range(steps):
        a = rs.randint(0, 5, size=n) if env._is_grid else rs.uniform(-1, 1, size=(n, 2))
        out = env.step(a)
        o.step(a, render=False)
        torch.cuda.synchronize()
        f, q = env.state_f64.cpu().numpy(), env.state_i32.cpu().numpy()
        assert np.array_equal(q, o.i32), 'seed %d: int state differs at step %d' % (seed, k)
        with np.errstate(invalid='ignore'):
            err = np.abs(f - o.f64)
        err = np.where(np.isnan(f) & np.isnan(o.f64), 0, err)
        err = np.where(f == o.f64, 0, err)
        assert float(np.max(err)) <= 1e-9, (seed, k, float(np.max(err)))
        assert np.array_equal(out.step_type.cpu().numpy(), o.step_type)
        assert helpers.same_or_nan(out.reward.cpu().numpy(), o.reward)
        o.f64[:], o.i32[:] = f, q   # lock step: removes 1-ulp libm / ocml drift
    assert np.array_equal(out.observation['image'].cpu().numpy(), o.render())
