"""Cycle breakdown of the step kernel for the mean and the slowest envs (profiling build: -DMOOG_PROFILE).
Run through tools/step_profile.sh, which builds gpurun_out/libmoog_hip_prof.so on the GPU box."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'moog.github.io_amd'))
import numpy as np, torch
from moog import environment
from moog_demos import example_configs
name = sys.argv[1] if len(sys.argv) > 1 else 'colliding_predators_32'
SECTIONS = [(1, 'single path tests'), (9, 'candidate batches (4 x 16 lanes)'), (2, 'contact search'), (3, 'make_disjoint'), (4, 'resolve'),
            (5, 'broad-phase scan'), (6, 'integrate'), (10, 'record load + boxes'), (11, 'rules + action'), (12, 'task reward')]
res = {}
for sel in [0, 7, 8] + [s for s, _ in SECTIONS]:
    env = environment.BatchedEnvironment(num_envs=int(os.environ.get('MOOG_PROFILE_ENVS', 4096)), seed=1, **example_configs.load(name))
    env.check_faults = False
    env.reset()
    env.set_debug(128 | (sel << 8), 0)
    for k in range(int(os.environ.get("MOOG_PROFILE_STEPS", 40))):
        ts = env.step(env.random_action())
    res[sel] = (ts.discount.cpu().numpy().copy(), ts.reward.cpu().numpy().copy())
    env.close()
tot = res[0][0]
order = np.argsort(-tot)
heavy = order[:40]
print('%s, step %s: total cycles  mean %%.0f  heaviest-40 mean %%.0f  max %%.0f' % (name, os.environ.get('MOOG_PROFILE_STEPS', '40')) % ( tot.mean(), tot[heavy].mean(), tot.max()))
cnt = res[0][1] % 1e10
print('  single path tests / contact searches: mean %.1f / %.1f, heaviest-40 %.1f / %.1f' % (
    (cnt % 100000).mean(), (cnt // 100000).mean(), (cnt[heavy] % 100000).mean(), (cnt[heavy] // 100000).mean()))
def line(nm, v):
    print('  %-34s mean %9.0f (%4.1f%%)   heaviest-40 %9.0f (%4.1f%%)' % (
        nm, v.mean(), 100 * v.mean() / tot.mean(), v[heavy].mean(), 100 * v[heavy].mean() / tot[heavy].mean()))
acc = np.zeros_like(tot)
for sel, nm in SECTIONS:
    line(nm, res[sel][1]); acc = acc + res[sel][1]
phys, coll = res[7][1], res[8][1]
inside_coll = sum(res[s][1] for s in (1, 9, 2, 3, 4, 5))
line('collision loop control', coll - inside_coll)
line('forces + substep scaffolding', phys - coll - res[6][1])
line('store + launch prologue / epilogue', tot - phys - res[10][1] - res[11][1] - res[12][1])
