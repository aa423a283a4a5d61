"""Pillow corpus through the wave rasteriser vs the workgroup rasteriser: which polygons differ, and where."""
import os, sys, collections
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'moog.github.io_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
import torch
import helpers
from moog import environment, action_spaces, observers, physics as physics_lib, sprite, tasks
z = dict(np.load(helpers.GOLDEN + '/raster.npz'))
W = int(sys.argv[1]) if len(sys.argv) > 1 else 64
idx = np.nonzero(z['size'] == W)[0]
def run(wave):
    os.environ['MOOG_RASTER_WAVE'] = wave
    cfg = dict(
        state_initializer=lambda: collections.OrderedDict(
            [('a', [sprite.Sprite(shape='circle', c0=200, c1=100, c2=50, opacity=128)]), ('agent', [])]),
        physics=physics_lib.Physics(updates_per_env_step=1), task=tasks.CompositeTask(),
        action_space=action_spaces.Grid(action_layers='agent'),
        observers={'image': observers.PILRenderer(image_size=(W, W), bg_color=tuple(z['bg']))})
    env = environment.BatchedEnvironment(num_envs=len(idx), **cfg)
    env.reset()
    torch.cuda.synchronize()
    f, q = env.state_f64.cpu().numpy(), env.state_i32.cpu().numpy()
    L, P = env.layout, env.compiled.program
    for i, k in enumerate(idx):
        nv = int(z['nv'][k]); xy = z['xy'][k, :nv].astype(np.float64)
        v = (xy + np.where(xy >= 0, 0.5, -0.5)) / W
        q[i, L.o_nverts] = nv
        f[i, L.o_verts:L.o_verts + 2 * nv] = v.ravel()
    env.state_f64.copy_(torch.from_numpy(f)); env.state_i32.copy_(torch.from_numpy(q))
    return env.observation()['image'].cpu().numpy()
a = run('1'); b = run('0')
bad = [i for i in range(len(idx)) if not np.array_equal(a[i], b[i])]
print('differ:', len(bad))
for i in bad[:6]:
    k = idx[i]; nv = int(z['nv'][k])
    print('poly', k, 'nv', nv, z['xy'][k, :nv].tolist())
    d = np.argwhere((a[i] != b[i]).any(axis=2))
    print(' diff pixels (row from top, col):', d[:12].tolist(), 'wave:', a[i][tuple(d[0])].tolist(), 'wg:', b[i][tuple(d[0])].tolist())
