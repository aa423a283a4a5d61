import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'moog.github.io_amd'))
import torch
from moog import environment, _abi
from moog_demos import example_configs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = environment.BatchedEnvironment(num_envs=n, seed=1, **example_configs.load('pacman'))
env.reset()
out = []
for k in range(30):
    env.step(env.random_action())
    torch.cuda.synchronize()
    out.append(env.env_prefix_slots)
print(n, out)
env.set_timing(True); env.kernel_time(_abi.MOOG_K_RASTER)
for _ in range(20):
    env.step(env.random_action())
ms, c = env.kernel_time(_abi.MOOG_K_RASTER)
print('raster %.1f us' % (ms / c * 1e3), env.env_prefix_slots)
