"""Soak test of moog_engine_set_fused: the frames of EVERY call of a long run against the separate launches
(position-weighted checksum per env and call, exact integer arithmetic), plus time steps and the final state.
usage: python tools/dbg/fused_soak.py [config] [envs] [steps]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'moog.github.io_amd'))
import numpy as np
import torch
from moog import environment
from moog_demos import example_configs

name = sys.argv[1] if len(sys.argv) > 1 else 'colliding_predators_32'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 600

def run(fused):
    env = environment.BatchedEnvironment(num_envs=n, seed=77, layer_capacity=example_configs.capacity(name), **example_configs.load(name))
    assert env.enable_cost_schedule(fused=fused) == fused
    env.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(3)
    sums = torch.zeros((steps, n), dtype=torch.int64, device='cuda')
    types = torch.zeros((steps, n), dtype=torch.int32, device='cuda')
    w = None
    for k in range(steps):
        if env._is_grid:
            a = torch.randint(0, 5, (n,), generator=g, dtype=torch.int32, device='cuda')
        else:
            a = torch.empty((n, 2), dtype=torch.float64, device='cuda').uniform_(-1, 1, generator=g)
        ts = env.step(a)
        img = ts.observation['image']
        if w is None:
            w = (torch.arange(img[0].numel(), device='cuda', dtype=torch.int64) % 8191) + 1
        sums[k] = (img.reshape(n, -1).to(torch.int64) * w).sum(1)
        types[k] = ts.step_type
    torch.cuda.synchronize()
    env.raise_faults()
    return sums, types, env.state_f64.clone(), env.state_i32.clone()

a, b = run(False), run(True)
bad = (a[0] != b[0])
print('%s: %d envs x %d calls = %d frames; frames that differ: %d; step types equal: %s; final state equal: %s' % (
    name, n, steps, n * steps, int(bad.sum()), bool(torch.equal(a[1], b[1])),
    bool(torch.equal(a[3], b[3]) and torch.equal(a[2].nan_to_num(), b[2].nan_to_num()))))
if bad.any():
    idx = bad.nonzero()[:10].tolist()
    print('first mismatches (call, env):', idx)
