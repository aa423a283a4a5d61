"""Per-frame work counters of the raster kernel (MOOG_RASTER_STOP=10 writes them instead of the frame)."""
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
os.environ['MOOG_RASTER_STOP'] = '10'
img = env.observation()['image']
torch.cuda.synchronize()
c = img.reshape(4096, -1)[:, :8].float()
for i, nm in enumerate(['listed edges', 'long edges', 'generic rows', 'multi-head rows', 'very long edges', 'rows / 4']):
    print('%-16s mean %.2f max %d' % (nm, c[:, i].mean().item(), int(c[:, i].max().item())))
