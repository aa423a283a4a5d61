"""Times the raster kernel truncated after each phase (MOOG_RASTER_STOP)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'moog.github.io_amd'))
import torch
from moog import environment, _abi
from moog_demos import example_configs
name = sys.argv[1] if len(sys.argv) > 1 else "colliding_predators_32"
kw = dict(image_size=(int(sys.argv[2]),) * 2) if len(sys.argv) > 2 else {}
env = environment.BatchedEnvironment(num_envs=4096, seed=1, **(__import__("moog_demos.example_configs." + name, fromlist=["x"]).get_config(0, **kw) if kw else example_configs.load(name)))
env.reset()
for _ in range(int(os.environ.get('MOOG_WARM_STEPS', '3'))):
    env.step(env.random_action())
    torch.cuda.synchronize()   # (the per-env prefix settles from what the host has seen of earlier calls)
print('per-env prefix slots:', env.env_prefix_slots)
for stop in [int(x) for x in os.environ.get('MOOG_RASTER_STOPS', '1,2,3,4,5,0').split(',')]:
    env.set_debug(0, stop)
    for _ in range(3):
        env.observation()
    env.set_timing(True); env.kernel_time(_abi.MOOG_K_RASTER)
    for _ in range(20):
        env.observation()
    ms, n = env.kernel_time(_abi.MOOG_K_RASTER)
    env.set_timing(False)
    print('stop after phase %d: %.1f us' % (stop, ms / n * 1e3))
