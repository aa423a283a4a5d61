import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'moog.github.io_amd'))
import numpy as np, torch
from moog import environment
from moog_demos import example_configs
env = environment.BatchedEnvironment(num_envs=4096, seed=1, **example_configs.load('colliding_predators_32'))
env.reset()
for _ in range(5): env.step(env.random_action())
os.environ['MOOG_RASTER_STOP'] = '99'
for _ in range(3): img = env.observation()['image']
torch.cuda.synchronize()
raw = img.cpu().numpy().reshape(4096, -1)[:, :4 * 64].copy().view(np.int64).reshape(4096, 4, 8)
names = ['1 verts', '2 edges', '3 scan', '4 masks', '4b queue', '5 compose', 'total']
for w in range(4):
    print('wave %d: ' % w + '  '.join('%s %.0f' % (n, raw[:, w, i].mean()) for i, n in enumerate(names)))
print('max total over blocks', raw[:, :, 6].max(), ' mean', raw[:, :, 6].mean())
