"""Debug aid: frames with / without the rasteriser's static-prefix cache against the oracle renderer."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(R, 'moog.github.io_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
import helpers
from moog import environment
from moog_demos import example_configs
name = sys.argv[1] if len(sys.argv) > 1 else 'pong'
n = 8
env = environment.BatchedEnvironment(num_envs=n, seed=11, env_index0=1000, **example_configs.load(name))
o = helpers.OracleEnv(env.compiled, n_envs=n, seed=11, env_index0=1000)
ns, pic = env.static_prefix()
print('static prefix slots', ns)
if pic is not None:
    pic = pic.cpu().numpy()
    print('picture: non-bg pixels', int((pic != pic[32, 32]).any(axis=2).sum()), 'rows with any', np.nonzero((pic != pic[32,32]).any(axis=(1,2)))[0][[0,-1]].tolist() if (pic != pic[32,32]).any() else None)
    for r in (0, 1, 2, 3, 4, 32, 60, 61, 62, 63):
        print('  row', r, ''.join('#' if (pic[r, x] != pic[32, 32]).any() else '.' for x in range(pic.shape[1])))
out = env.reset()
torch.cuda.synchronize()
o.f64[:], o.i32[:] = env.state_f64.cpu().numpy(), env.state_i32.cpu().numpy()
img = out.observation['image'].cpu().numpy(); ref = o.render()
print('after reset: differing envs', np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0].tolist())
for i in (1,):
    print('row 7 got', img[i, 7, :6].tolist(), 'want', ref[i, 7, :6].tolist(), 'row 8 right got', img[i, 8, 58:].tolist(), 'want', ref[i, 8, 58:].tolist())
    for r in range(0):
        print('%2d ' % r + ''.join(('X' if (img[i, r, x] != ref[i, r, x]).any() else ('#' if (ref[i, r, x] != 0).any() else '.')) for x in range(64)))
rs = np.random.RandomState(5)
for k in range(4):
    a = rs.randint(0, 5, size=n) if env._is_grid else rs.uniform(-1, 1, size=(n, 2))
    out = env.step(a)
    torch.cuda.synchronize()
    o.f64[:], o.i32[:] = env.state_f64.cpu().numpy(), env.state_i32.cpu().numpy()
    img = out.observation['image'].cpu().numpy(); ref = o.render()
    bad = np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0]
    print('step', k, 'differing envs', bad.tolist())
    if bad.size:
        i = bad[0]
        ys, xs = np.nonzero((img[i] != ref[i]).any(axis=2))
        print('  env', i, 'pixels', len(ys), 'rows', ys.min(), ys.max(), 'cols', xs.min(), xs.max())
        print('  got', img[i, ys[0], xs[0]], 'want', ref[i, ys[0], xs[0]])
        break
