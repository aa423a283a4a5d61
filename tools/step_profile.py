"""Cycle breakdown of the step kernel for the slowest envs (profiling build: -DMOOG_PROFILE).
Run through tools/step_profile.sh, which builds lib/libmoog_hip_prof.so on the GPU box."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'moog.github.io_amd'))
import numpy as np, torch
from moog import environment
from moog_demos import example_configs
NAMES = ['path tests', 'contact search', 'make_disjoint', 'resolve', 'broad-phase scan', 'integrate', 'apply_physics (all)', 'collision loops (all)']
res = {}
for sel in range(len(NAMES) + 1):
    os.environ['MOOG_STEP_DEBUG'] = str(128 | (sel << 8))
    env = environment.BatchedEnvironment(num_envs=4096, seed=1, **example_configs.load('colliding_predators_32'))
    env.check_faults = False
    env.reset()
    for k in range(40):
        ts = env.step(env.random_action())
    res[sel] = (ts.discount.cpu().numpy().copy(), ts.reward.cpu().numpy().copy())
    env.close()
tot = res[0][0]
order = np.argsort(-tot)
heavy = order[:40]
print('step 40: total cycles  mean %.0f  heaviest-40 mean %.0f  max %.0f' % (tot.mean(), tot[heavy].mean(), tot.max()))
cnt = res[0][1]
print('  path tests / contact searches: mean %.1f / %.1f, heaviest-40 %.1f / %.1f' % (
    (cnt % 100000).mean(), (cnt // 100000).mean(), (cnt[heavy] % 100000).mean(), (cnt[heavy] // 100000).mean()))
acc_all = acc_heavy = 0
for sel, name in enumerate(NAMES, 1):
    v = res[sel][1]
    print('  %-18s mean %9.0f (%4.1f%%)   heaviest-40 %9.0f (%4.1f%%)' % (
        name, v.mean(), 100 * v.mean() / tot.mean(), v[heavy].mean(), 100 * v[heavy].mean() / tot[heavy].mean()))
phys, coll = res[7][1], res[8][1]
integ = res[6][1]
for name, v in (('collision loop overhead', coll - sum(res[k][1] for k in range(1, 6))),
                ('forces + scaffolding', phys - coll - integ),
                ('outside apply_physics', tot - phys)):
    print('  %-18s mean %9.0f (%4.1f%%)   heaviest-40 %9.0f (%4.1f%%)' % (
        name, v.mean(), 100 * v.mean() / tot.mean(), v[heavy].mean(), 100 * v[heavy].mean() / tot[heavy].mean()))
