import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(R, 'tests')); sys.path.insert(0, os.path.join(R, 'moog.github.io_amd'))
import numpy as np, torch, helpers
from helpers import *
from test_gpu_parity import make_env, upload, download, padded_uniforms
name, t = sys.argv[1], int(sys.argv[2])
c = compiled(name); fx = fixture(name, 0); L = c.layout; P = c.program
print('born_rule', P.born_rule, 'persist', [int(P.slot_persist[s]) for s in range(P.n_slots)], 'n_rules', P.n_rules)
o = OracleEnv(c)
records_from_fixture(fx, t - 1, c, o.f64, o.i32)
f0, q0 = o.f64.copy(), o.i32.copy()
o.step(helpers.action_of(fx, t), uniforms=uniforms_of(fx, t))
env = make_env(name, 1)
upload(env, f0, q0)
env.check_faults = False
out = env.step(np.stack([helpers.action_of(fx, t)]), injected_uniforms=padded_uniforms(fx, [t]))
f, q = download(env)
S = P.n_slots
print('flags oracle', o.i32[0, L.o_flags:L.o_flags + S] & 1)
print('flags device', q[0, L.o_flags:L.o_flags + S] & 1)
print('ref alive   ', fx['alive'][t])
print('rng oracle', o.i32[0, L.o_rng:L.o_rng + 4], 'device', q[0, L.o_rng:L.o_rng + 4])
print('rule oracle', o.f64[0, L.o_rule:L.o_rule + P.n_rules], 'device', f[0, L.o_rule:L.o_rule + P.n_rules])
print('uniforms', uniforms_of(fx, t)[:12])
