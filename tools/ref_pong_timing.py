"""BASELINE config 1 on the CPU: the reference's pong example config, one env, stepped by the reference itself the way
its tests/runtime_benchmark.py:64-157 does (20 resets x 20 calls per phase; that script needs absl / tqdm, which this image
lacks, so the same loops are restated here).  Build container only (imports /root/reference):
    PYTHONPATH=oracle/shim:/root/reference MPLBACKEND=Agg python tools/ref_pong_timing.py"""
import importlib.util
import time

import numpy as np

from moog import environment

spec = importlib.util.spec_from_file_location('ref_pong', '/root/reference/moog_demos/example_configs/pong.py')
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)


def phase(name, fn, env):
    ts = []
    for _ in range(20):
        env.reset()
        t0 = time.time()
        for _ in range(20):
            fn()
        ts.append((time.time() - t0) / 20.0)
    ms = 1e3 * np.array(ts)
    print('  %-28s %8.3f ms/step  (stddev %.3f)  %8.1f steps/s' % (name, ms.mean(), ms.std(), 1e3 / ms.mean()))


cfg = mod.get_config(0)
env = environment.Environment(**cfg)
space = env.action_space
print('reference pong, 1 env, CPU (this container), 20 resets x 20 calls per phase')
phase('step + 64x64 PIL render', lambda: env.step(space.random_action()), env)
t0 = time.time()
for _ in range(20):
    env.reset()
print('  %-28s %8.3f ms/reset' % ('reset only', 1e3 * (time.time() - t0) / 20))
phase('physics only', lambda: env.physics.step(env.state), env)
phase('render only', lambda: env.observation(), env)
cfg2 = mod.get_config(0)
cfg2['observers'] = {}
env2 = environment.Environment(**cfg2)
phase('step, observers disabled', lambda: env2.step(space.random_action()), env2)
