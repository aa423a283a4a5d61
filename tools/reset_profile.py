"""Cycle breakdown of a reset that runs inside the step kernel (profiling build: tools/step_profile_variant.sh builds it):
MOOG_HIP_LIB=gpurun_out/libmoog_hip_prof.so python tools/reset_profile.py pacman"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'moog.github.io_amd'))
import numpy as np, torch
from moog import environment
from moog_demos import example_configs
name = sys.argv[1] if len(sys.argv) > 1 else 'pacman'
n = int(os.environ.get('MOOG_PROFILE_ENVS', 256))
SECTIONS = [(14, 'maze generator'), (15, 'cell selection'), (16, 'sample_op_factors'), (13, 'create_sprite'), (11, 'rule reset + first rule step')]
res = {}
for sel in [0] + [s for s, _ in SECTIONS]:
    env = environment.BatchedEnvironment(num_envs=n, seed=1, **example_configs.load(name))
    env.check_faults = False
    env.reset()
    for k in range(3):
        env.step(env.random_action())
    env.state_i32[:, env.layout.o_reset_next] = 1      # every env resets inside the next step call
    env.set_debug(128 | (sel << 8), 0)
    ts = env.step(env.random_action())
    res[sel] = (ts.discount.cpu().numpy().copy(), ts.reward.cpu().numpy().copy())
    env.close()
tot = res[0][0]
print('%s: reset inside the step kernel, %d envs: cycles mean %.0f  max %.0f' % (name, n, tot.mean(), tot.max()))
acc = 0
for sel, nm in SECTIONS:
    v = res[sel][1]
    acc += v.mean()
    print('  %-32s mean %9.0f (%4.1f%%)' % (nm, v.mean(), 100 * v.mean() / tot.mean()))
print('  %-32s mean %9.0f (%4.1f%%)' % ('everything else', tot.mean() - acc, 100 * (tot.mean() - acc) / tot.mean()))
