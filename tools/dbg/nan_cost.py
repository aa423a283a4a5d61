"""falling_balls_64: are the slowest envs the ones that hold non-finite sprites?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'moog.github.io_amd'))
import numpy as np, torch
from moog import environment
from moog_demos import example_configs
n = 1024
env = environment.BatchedEnvironment(num_envs=n, seed=1, **example_configs.load('falling_balls_64'))
env.check_faults = False
env.enable_cost_schedule()
env.reset()
for k in range(120):
    env.step(env.random_action())
torch.cuda.synchronize()
L = env.layout
f = env.state_f64.cpu().numpy()
q = env.state_i32.cpu().numpy()
S = env.compiled.program.n_slots
pos = f[:, L.o_pos:L.o_pos + 2 * S].reshape(n, S, 2)
alive = (q[:, L.o_flags:L.o_flags + S] & 1) != 0
bad = (~np.isfinite(pos).all(2)) & alive
c = env._cost.cpu().numpy()
nb = bad.sum(1)
o = np.argsort(-c)
print('envs with a non-finite live sprite: %d of %d' % ((nb > 0).sum(), n))
print('cost of the 10 slowest envs (Mcycles):', np.round(c[o[:10]] / 1e6, 1), ' non-finite sprites there:', nb[o[:10]])
print('mean cost with / without non-finite sprites: %.2f / %.2f Mcycles' % (c[nb > 0].mean() / 1e6 if (nb > 0).any() else 0, c[nb == 0].mean() / 1e6))
# pile density of the slow envs: pairs of balls whose centres are closer than half a diameter
for e in o[:5]:
    p = pos[e][alive[e]]
    d = np.linalg.norm(p[:, None] - p[None], axis=2)
    iu = np.triu_indices(len(p), 1)
    print('env %d: cost %.1f M, live %d, centre pairs closer than 0.01 / 0.02 / 0.04: %d / %d / %d, min y %.3f' % (
        e, c[e] / 1e6, alive[e].sum(), (d[iu] < 0.01).sum(), (d[iu] < 0.02).sum(), (d[iu] < 0.04).sum(), np.nanmin(p[:, 1])))
