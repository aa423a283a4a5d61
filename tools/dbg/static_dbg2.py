"""Do the leading constant sprites keep their bits across steps?  (the rasteriser's static prefix)"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(R, 'moog.github.io_amd'))
import numpy as np, torch
from moog import environment, _abi
from moog_demos import example_configs
name = sys.argv[1] if len(sys.argv) > 1 else 'colliding_predators_32'
n = 4096
env = environment.BatchedEnvironment(num_envs=n, seed=11, **example_configs.load(name))
env.reset()
L = env.layout
ns, _ = env.static_prefix()
nsv = int(env.compiled.program.slot_voff[ns])
v0 = env.state_f64[:, L.o_verts:L.o_verts + 2 * nsv].clone()
def t():
    env.set_timing(True); env.kernel_time(_abi.MOOG_K_RASTER)
    for _ in range(20): env.observation()
    ms, k = env.kernel_time(_abi.MOOG_K_RASTER); env.set_timing(False)
    return ms / k * 1e3
print('slots', ns, 'vertex slots', nsv, 'raster after reset %.1f us' % t())
for k in range(5):
    env.step(env.random_action())
v1 = env.state_f64[:, L.o_verts:L.o_verts + 2 * nsv]
same_bits = (v0.view(torch.int64) == v1.view(torch.int64)).all(dim=1)
print('envs whose prefix vertices kept their bits:', int(same_bits.sum()), 'of', n, ' equal by value:', int((v0 == v1).all(dim=1).sum()))
bad = (v0.view(torch.int64) != v1.view(torch.int64)).nonzero()
if len(bad): print('first changed', bad[0].tolist(), v0[bad[0][0], bad[0][1]].item(), v1[bad[0][0], bad[0][1]].item())
print('raster after 5 steps %.1f us' % t())
