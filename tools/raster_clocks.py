"""Phase clocks of waves 0 and 3 of the raster kernel (MOOG_RASTER_STOP=11 writes them instead of the frame)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'moog.github.io_amd'))
import torch
from moog import environment
from moog_demos import example_configs
name = sys.argv[1] if len(sys.argv) > 1 else "colliding_predators_32"
kw = dict(image_size=(int(sys.argv[2]),) * 2) if len(sys.argv) > 2 else {}
env = environment.BatchedEnvironment(num_envs=4096, seed=1, **(__import__("moog_demos.example_configs." + name, fromlist=["x"]).get_config(0, **kw) if kw else example_configs.load(name)))
env.reset()
for _ in range(3):
    env.step(env.random_action())
os.environ['MOOG_RASTER_STOP'] = '11'
for _ in range(3):
    img = env.observation()['image']
torch.cuda.synchronize()
c = img.reshape(4096, -1)[:, :64].contiguous().view(torch.int32).float()
names = ['setup (0-2)', 'push 3a', '3b + barrier', 'main rows', 'rare rows', 'barrier wait', 'total', 'tips (lane 0)']
for wv, off in (('wave 0', 0), ('wave 3', 8)):
    print(wv, ' '.join('%s=%.0f' % (n, c[:, off + i].mean().item()) for i, n in enumerate(names)))
