"""Who sets the step kernel's duration in the stationary mix: the slowest STEPPING env or the slowest RESETTING env (auto-reset runs
inside the same launch)?  Per-env cycles of a call come from the cost words of the launch schedule.  python tools/dbg/reset_tail.py"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(R, 'moog.github.io_amd')); sys.path.insert(0, R)
import numpy as np, torch
from moog import environment
from moog_demos import example_configs
import bench
name = sys.argv[1] if len(sys.argv) > 1 else 'colliding_predators_32'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = environment.BatchedEnvironment(num_envs=n, seed=2024, layer_capacity=example_configs.capacity(name), **example_configs.load(name))
env.enable_cost_schedule()
env.reset()
T = int(env.compiled.program.timeout_steps)
bench.stagger_episodes(env, T, 0)
for _ in range(T + 1):
    env.step(env.random_action())
rows = []
for k in range(40):
    ts = env.step(env.random_action())
    torch.cuda.synchronize()
    c = env._cost.cpu().numpy().astype(np.float64)
    first = ts.step_type.cpu().numpy() == 0
    rows.append((first.sum(), c[first].max() if first.any() else 0, c[first].mean() if first.any() else 0, c[~first].max(), c[~first].mean()))
r = np.array(rows)
print('%s, %d envs, 40 calls of the stationary mix: resetting envs per call %.1f' % (name, n, r[:, 0].mean()))
print('  cycles of the slowest RESETTING env per call: mean %.0f  max %.0f   (mean reset %.0f)' % (r[:, 1].mean(), r[:, 1].max(), r[:, 2].mean()))
print('  cycles of the slowest STEPPING env per call:  mean %.0f  max %.0f   (mean step %.0f)' % (r[:, 3].mean(), r[:, 3].max(), r[:, 4].mean()))
print('  calls in which a resetting env was the slowest of the launch: %d of 40' % int((r[:, 1] > r[:, 3]).sum()))
