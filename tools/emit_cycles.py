"""Cycles the draw-record emitter takes inside the step kernel (MOOG step debug bit 256: the step type of every env is replaced by
the emitter's shader-clock cycles; bit 128: the discount by the env's whole step), by launch rank."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'moog.github.io_amd'))
import numpy as np
import torch
from moog import environment
from moog_demos import example_configs
name = sys.argv[1] if len(sys.argv) > 1 else 'colliding_predators_32'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = environment.BatchedEnvironment(num_envs=n, seed=1, layer_capacity=example_configs.capacity(name), **example_configs.load(name))
env.check_faults = False
env.enable_cost_schedule()
env.reset()
for _ in range(30):
    env.step(env.random_action())
env.set_debug(step_debug=256)
emit, total = [], []
for _ in range(10):
    ts = env.step(env.random_action())
    torch.cuda.synchronize()
    emit.append(env.step_type.cpu().numpy().astype(np.float64))
    d, r = env.discount.cpu().numpy(), env.reward.cpu().numpy()
    total.append(np.stack([np.mod(d, 65536.0), np.floor(d / 65536.0), np.mod(r, 65536.0), np.floor(r / 65536.0)]))
env.set_debug(step_debug=0)
emit, ph = np.stack(emit), np.stack(total)
ok = np.isfinite(ph).all(axis=1)
print('%s, %d envs, step kernel %s: emitter cycles per env-step  mean %.0f  p50 %.0f  p99 %.0f  max %.0f' % (
    name, n, env.step_kernel(), emit.mean(), np.percentile(emit, 50), np.percentile(emit, 99), emit.max()))
print('   by phase (mean cycles; shader clock):  prefix check %.0f | slots: colours, liveness, scan %.0f | vertex slots: points, bounds %.0f | items: rows, records %.0f' % tuple(
    float(np.nanmean(np.where(ok, ph[:, k], np.nan))) for k in range(4)))
