"""pacman: step-kernel time per call after a common reset (lock step): calls without resetting envs vs calls with."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'moog.github.io_amd'))
import torch
from moog import environment
from moog_demos import example_configs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = environment.BatchedEnvironment(num_envs=n, seed=1, **example_configs.load('pacman'))
env.check_faults = False
env.reset()
env.set_timing(True)
for k in range(40):
    for j in range(3): env.kernel_time(j)
    resetting = int((env.state_i32[:, env.layout.o_reset_next] != 0).sum().item())
    env.step(env.random_action())
    torch.cuda.synchronize()
    t, r = env.kernel_time(0), env.kernel_time(1)
    if k % 3 == 0 or resetting:
        print('call %2d: envs resetting in it %4d   step kernel %7.0f us   raster %6.0f us' % (k, resetting, t[0] * 1e3, r[0] * 1e3), flush=True)
