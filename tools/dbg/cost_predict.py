"""How well does an env's cost in step t predict its cost in step t + 1?  (moog_engine_set_pipeline's premise)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'moog.github.io_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np
import torch
from moog import environment
from moog_demos import example_configs
import bench

n = 4096
env = environment.BatchedEnvironment(num_envs=n, seed=2024, **example_configs.load('colliding_predators_32'))
env.enable_cost_schedule()
env.reset()
P = env.compiled.program
bench.stagger_episodes(env, P.timeout_steps, 0)
g = torch.Generator(device='cuda'); g.manual_seed(1)
def step():
    env.step(torch.empty((n, 2), dtype=torch.float64, device='cuda').uniform_(-1, 1, generator=g))
for _ in range(int(P.timeout_steps) + 20):
    step()
prev = None
rows = []
for k in range(40):
    step()
    torch.cuda.synchronize()
    c = env._cost.cpu().numpy().copy()
    rn = (env.state_i32[:, env.layout.o_reset_next] == 1).cpu().numpy()
    if prev is not None:
        pc, prn = prev
        score = np.where(prn, np.inf, pc)           # resetting envs are put first
        order = np.argsort(-score, kind='stable')
        mx = c.max()
        r = [np.corrcoef(pc, c)[0, 1]]
        for H in (256, 512, 1024, 2048):
            r.append(c[order[H:]].max() / mx)
        r.append(np.sort(c)[-41:-1].mean() / mx)
        r.append(c.mean() / mx)
        rows.append(r)
    prev = (c, rn)
rows = np.array(rows)
print('corr(cost t, cost t+1) %.3f' % rows[:, 0].mean())
for i, H in enumerate((256, 512, 1024, 2048)):
    print('slowest env outside the predicted top %4d: %.3f of the step maximum (mean over steps), min %.3f max %.3f' % (H, rows[:, 1 + i].mean(), rows[:, 1 + i].min(), rows[:, 1 + i].max()))
print('heaviest-40 mean / max %.3f   mean / max %.3f' % (rows[:, 5].mean(), rows[:, 6].mean()))
