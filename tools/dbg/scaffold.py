"""Step-kernel time with parts switched off (MOOG_STEP_DEBUG bits: 1 skip collisions, 2 skip integrate): what the sub-step
loop costs when it does nothing."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'moog.github.io_amd'))
import torch
from moog import environment
from moog_demos import example_configs
name = sys.argv[1] if len(sys.argv) > 1 else 'colliding_predators_32'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
for dbg in (0, 1, 2, 3):
    env = environment.BatchedEnvironment(num_envs=n, seed=1, layer_capacity=example_configs.capacity(name), **example_configs.load(name))
    env.check_faults = False
    env.reset()
    for _ in range(30):
        env.step(env.random_action())
    env.set_debug(dbg, 0)
    env.set_timing(True)
    for k in range(3): env.kernel_time(k)
    for _ in range(20):
        env.step(env.random_action())
    torch.cuda.synchronize()
    t = env.kernel_time(0)
    print('%s dbg %d: step kernel %.1f us' % (name, dbg, t[0] / max(t[1], 1) * 1e3), flush=True)
    env.close()
