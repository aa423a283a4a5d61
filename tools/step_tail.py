"""Per-env cycle counts of the step kernel (MOOG_STEP_DEBUG=128 writes clock64 deltas into `discount`)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'moog.github.io_amd'))
os.environ['MOOG_STEP_DEBUG'] = '128'
import numpy as np, torch
from moog import environment
from moog_demos import example_configs
NAME = sys.argv[1] if len(sys.argv) > 1 else 'colliding_predators_32'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 60
env = environment.BatchedEnvironment(num_envs=N, seed=1, **example_configs.load(NAME))
env.check_faults = False
env.reset()
for k in range(STEPS):
    ts = env.step(env.random_action())
    if k % 10 == 9:
        c = ts.discount.cpu().numpy()
        r = ts.reward.cpu().numpy()
        keep = ts.step_type.cpu().numpy() != 0
        c, r = c[keep], r[keep]
        npath, nresp = r % 100000, r // 100000
        A = np.stack([np.ones_like(c), npath, nresp], 1)
        coef = np.linalg.lstsq(A, c, rcond=None)[0]
        order = np.argsort(-c)[:5]
        print('  fit cycles = %.0f + %.0f*path_tests + %.0f*contact_searches; mean path %.1f resp %.1f; slowest:' % (
            coef[0], coef[1], coef[2], npath.mean(), nresp.mean()),
            [(int(c[i]), int(npath[i]), int(nresp[i])) for i in order])
        q = np.percentile(c, [0, 10, 50, 90, 99, 99.9, 100])
        print('step %d cycles/env: min %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f  mean %.0f  slowest/mean %.2f' % ((k,) + tuple(q) + (c.mean(), c.max() / c.mean())))
