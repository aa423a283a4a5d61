import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(R, 'moog.github.io_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, helpers
from helpers import compiled, fixture, records_from_fixture
import test_gpu_parity as T
name, seed = 'callables_zoo', 1
c, fx = compiled(name), fixture(name, seed)
n = len(fx['step_type']); ts = list(range(1, n))
env = T.make_env(name, len(ts)); L = c.layout
f64 = np.zeros((len(ts), L.f64_per_env)); i32 = np.zeros((len(ts), L.i32_per_env), np.int32)
for i, t in enumerate(ts): records_from_fixture(fx, t - 1, c, f64, i32, env=i)
T.upload(env, f64, i32)
actions = np.stack([helpers.action_of(fx, t) for t in ts])
env.check_faults = False
out = env.step(actions, injected_uniforms=T.padded_uniforms(fx, ts))
o = helpers.OracleEnv(c, n_envs=len(ts), seed=0)
o.f64[:], o.i32[:] = f64, i32
o.step(actions, uniforms=T.padded_uniforms(fx, ts), render=False)
r = out.reward.cpu().numpy()
for i, t in enumerate(ts):
    if not helpers.same_or_nan(float(r[i]), fx['reward'][t]):
        P = c.program
        print('call', t, 'engine', r[i], 'reference', fx['reward'][t], 'oracle', o.reward[i], 'task counters before', f64[i, L.o_task:L.o_task + 3], 'rule state', f64[i, L.o_rule:L.o_rule + P.n_rules])
        print('   alive before', i32[i, L.o_flags:L.o_flags + L.S] & 1, 'agent x', f64[i, L.o_pos + 2 * 8])
