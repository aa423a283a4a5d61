"""A few raster launches of the headline workload (for rocprofv3 runs that only want that kernel)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'moog.github.io_amd'))
import torch
from moog import environment
from moog_demos import example_configs
name = sys.argv[1] if len(sys.argv) > 1 else "colliding_predators_32"
env = environment.BatchedEnvironment(num_envs=4096, seed=1, **example_configs.load(name))
env.reset()
for _ in range(10):
    env.step(env.random_action())
for _ in range(8):
    env.observation()
torch.cuda.synchronize()
