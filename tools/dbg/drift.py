"""How fast do engine and oracle drift apart when nothing re-synchronises them? (window length of the
un-resynced full-size test)"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(R, 'moog.github.io_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
import helpers
from moog import environment
from moog_demos import example_configs
n, steps = 4096, 64
env = environment.BatchedEnvironment(num_envs=n, seed=23, **example_configs.load('colliding_predators_32'))
o = helpers.OracleEnv(env.compiled, n_envs=n, seed=23)
env.reset(); o.reset(render=False)
rs = np.random.RandomState(4)
for k in range(steps):
    a = rs.uniform(-1, 1, size=(n, 2))
    env.step(a); o.step(a, render=False)
    torch.cuda.synchronize()
    f, q = env.state_f64.cpu().numpy(), env.state_i32.cpu().numpy()
    with np.errstate(invalid='ignore'):
        err = np.where(f == o.f64, 0, np.abs(f - o.f64))
    err = np.where(np.isnan(f) & np.isnan(o.f64), 0, err)
    per_env = err.max(axis=1)
    print('step %2d  max %.3g  envs > 1e-9: %d  > 1e-5: %d  int mismatches: %d' % (
        k, per_env.max(), int((per_env > 1e-9).sum()), int((per_env > 1e-5).sum()), int((q != o.i32).any(axis=1).sum())))
