"""Long lock-step run of the engine against the oracle at full size, across an episode boundary (auto-reset):
python tools/dbg/long_lockstep.py [config [n_envs [steps]]].  Integer records exact, floats <= 1e-9, frames every 25 steps."""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(R, 'moog.github.io_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, helpers
from moog import environment
from moog_demos import example_configs
name = sys.argv[1] if len(sys.argv) > 1 else 'colliding_predators_32'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 230
# (the recipes' own LAYER_CAPACITY values are the ones the reference fixtures were recorded with: too small for thousands of random-policy envs)
ROOM = {'first_person_predators_prey': {'prey': 32, 'predators': 96}, 'rules_zoo_l1': {'prey': 24, 'predators': 24}}
env = environment.BatchedEnvironment(num_envs=n, seed=23, layer_capacity=ROOM.get(name, example_configs.capacity(name)), **example_configs.load(name))
o = helpers.OracleEnv(env.compiled, n_envs=n, seed=23)
env.reset(); o.reset(render=False)
rs = np.random.RandomState(4); t0 = time.time(); worst = 0.0; resets = 0
for k in range(steps):
    a = rs.randint(0, 5, size=n) if env._is_grid else rs.uniform(-1, 1, size=(n, 2))
    out = env.step(a); o.step(a, render=False)
    torch.cuda.synchronize()
    f, q = env.state_f64.cpu().numpy(), env.state_i32.cpu().numpy()
    assert np.array_equal(q, o.i32), 'int state differs at step %d' % k
    with np.errstate(invalid='ignore'):
        err = np.where((f == o.f64) | (np.isnan(f) & np.isnan(o.f64)), 0, np.abs(f - o.f64))
    worst = max(worst, float(err.max())); assert worst <= 1e-9, (k, worst)
    st = out.step_type.cpu().numpy(); assert np.array_equal(st, o.step_type); resets += int((st == 0).sum())
    assert helpers.same_or_nan(out.reward.cpu().numpy(), o.reward)
    o.f64[:], o.i32[:] = f, q
    if k % 25 == 24:
        assert np.array_equal(out.observation['image'].cpu().numpy(), o.render()), 'frames differ at step %d' % k
print('%s: %d envs x %d steps in lock step with the oracle: records exact, max float diff %.3g, %d auto-resets, frames exact every 25 steps (%.0f s)' % (
    name, n, steps, worst, resets, time.time() - t0))
