"""colliding_predators_32: step-kernel time with one Collision force left out at a time (lock-step episode, 4096 envs)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'moog.github.io_amd'))
import torch
from moog import environment, physics as physics_lib
from moog_demos import example_configs
def run(label, keep):
    cfg = example_configs.load('colliding_predators_32')
    ph = cfg['physics']
    forces = [f for i, f in enumerate(ph._forces) if keep(i, f)]
    cfg['physics'] = physics_lib.Physics(*forces, updates_per_env_step=ph._updates_per_env_step)
    env = environment.BatchedEnvironment(num_envs=4096, seed=1, **cfg)
    env.check_faults = False
    env.reset()
    for _ in range(30):
        env.step(env.random_action())
    env.set_timing(True)
    for k in range(3): env.kernel_time(k)
    for _ in range(20):
        env.step(env.random_action())
    torch.cuda.synchronize()
    t = env.kernel_time(0)
    print('%-44s step kernel %.0f us' % (label, t[0] / max(t[1], 1) * 1e3), flush=True)
    env.close()
cfg = example_configs.load('colliding_predators_32')
for i, f in enumerate(cfg['physics']._forces):
    print(i, type(f[0]).__name__, f[1:], flush=True)
run('all forces', lambda i, f: True)
for i in range(len(cfg['physics']._forces)):
    run('without force %d' % i, lambda j, f, i=i: j != i)
