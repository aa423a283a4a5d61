"""Fused step + raster launch (moog_engine_set_fused): result neutrality against the separate launches."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'moog.github.io_amd'))
import numpy as np
import torch
from moog import environment
from moog_demos import example_configs

def run(fused, n, steps, name='colliding_predators_32'):
    env = environment.BatchedEnvironment(num_envs=n, seed=5, **example_configs.load(name))
    got = env.enable_cost_schedule(fused=fused)
    assert got == fused, got
    env.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    frames = []
    for k in range(steps):
        a = torch.empty((n, 2), dtype=torch.float64, device='cuda').uniform_(-1, 1, generator=g)
        ts = env.step(a)
        if k % 5 == 0 or k == steps - 1:
            frames.append((ts.observation['image'].clone(), ts.reward.clone(), ts.step_type.clone()))
    torch.cuda.synchronize()
    env.raise_faults()
    return env.state_f64.clone(), env.state_i32.clone(), frames

for n, steps in ((64, 40), (1000, 80), (4096, 60)):
    a = run(False, n, steps)
    b = run(True, n, steps)
    ok = torch.equal(a[1], b[1]) and torch.equal(a[0].nan_to_num(), b[0].nan_to_num())
    bad_frames = 0
    for x, y in zip(a[2], b[2]):
        bad_frames += int((x[0] != y[0]).flatten(1).any(1).sum())
        ok = ok and torch.equal(x[1].nan_to_num(), y[1].nan_to_num()) and torch.equal(x[2], y[2])
    print('n', n, 'state/outputs', 'identical' if ok else 'DIFFERENT', 'frames differing', bad_frames)
