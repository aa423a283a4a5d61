import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'moog.github.io_amd'))
import numpy as np, torch
import helpers
from helpers import *
name, seed = 'sampler_zoo_l2', 0
c, fx = compiled(name), fixture(name, seed)
L = c.layout
from moog import environment
from moog_demos import example_configs
for t in (25, 50):
    o = OracleEnv(c)
    records_from_fixture(fx, t - 1, c, o.f64, o.i32)
    u = uniforms_of(fx, t)
    o.step(fx['action'][t], uniforms=u)
    env = environment.BatchedEnvironment(num_envs=1, layer_capacity=example_configs.capacity(name), **example_configs.load(name))
    f64, i32 = np.zeros_like(o.f64), np.zeros_like(o.i32)
    records_from_fixture(fx, t - 1, c, f64, i32)
    env.state_f64.copy_(torch.as_tensor(f64)); env.state_i32.copy_(torch.as_tensor(i32))
    env.step(np.asarray(fx['action'][t]).reshape(1, -1), injected_uniforms=np.asarray(u).reshape(1, -1))
    torch.cuda.synchronize()
    q = env.state_i32.cpu().numpy()
    print('t', t, 'uniforms', np.round(u[-4:], 4), 'n', len(u))
    print(' ref    nverts', fx['nverts'][t][4:10])
    print(' oracle nverts', o.i32[0, L.o_nverts + 4:L.o_nverts + 10], 'flags', o.i32[0, L.o_flags + 4:L.o_flags + 10])
    print(' engine nverts', q[0, L.o_nverts + 4:L.o_nverts + 10], 'flags', q[0, L.o_flags + 4:L.o_flags + 10], 'fault', q[0, L.o_fault])
    env.close()
