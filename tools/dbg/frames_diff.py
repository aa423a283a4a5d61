"""Finds a frame that differs from the oracle renderer and saves the state + both frames (debug aid)."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(R, 'moog.github.io_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, helpers
from moog import environment
from moog_demos import example_configs
name = sys.argv[1]; n = 256
env = environment.BatchedEnvironment(num_envs=n, seed=21, env_index0=300, layer_capacity=example_configs.capacity(name), **example_configs.load(name))
o = helpers.OracleEnv(env.compiled, n_envs=n, seed=21, env_index0=300)
env.reset(); rs = np.random.RandomState(8)
for k in range(12):
    a = rs.randint(0, 5, size=n) if env._is_grid else rs.uniform(-1, 1, size=(n, 2))
    out = env.step(a); torch.cuda.synchronize()
    o.f64[:], o.i32[:] = env.state_f64.cpu().numpy(), env.state_i32.cpu().numpy()
    img = out.observation['image'].cpu().numpy(); ref = o.render()
    bad = np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0]
    if bad.size:
        e = int(bad[0])
        ys, xs = np.nonzero((img[e] != ref[e]).any(axis=2))
        print('step', k, 'env', e, 'pixels (row,col) in the flipped frame:', list(zip(ys.tolist(), xs.tolist()))[:20])
        np.savez(os.path.join(R, 'gpurun_out', 'dbg_frame.npz'), f64=o.f64[e], i32=o.i32[e], img=img[e], ref=ref[e])
        break
else:
    print('no difference')
