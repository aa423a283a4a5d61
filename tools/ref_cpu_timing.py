"""The reference itself (the imported Python package, one env, one core) on the two scaled BASELINE workloads: config 3
(colliding_predators scaled to 32 sprites) and config 5 (falling_balls scaled to 64 sprites), with its 64 x 64 PIL
observer, the way BASELINE.md section 2 measured them -- random actions, wall clock over the calls after a reset.  The
scaled recipes are this repo's example_configs run against the reference package (as tests/golden/make_golden.py does).
Build container only (imports /root/reference; it does not exist on the GPU box):
    PYTHONPATH=oracle/shim:/root/reference MPLBACKEND=Agg python tools/ref_cpu_timing.py > profiles/r05_ref_cpu.txt"""
import importlib
import importlib.util
import os
import platform
import sys
import time

import numpy as np

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
CFG_DIR = os.path.join(REPO, 'moog.github.io_amd', 'moog_demos', 'example_configs')

from moog import environment  # noqa: E402  (the reference package: PYTHONPATH)

assert '/root/reference' in os.path.abspath(environment.__file__), 'run with PYTHONPATH=oracle/shim:/root/reference'
spec = importlib.util.spec_from_file_location('amd_configs', os.path.join(CFG_DIR, '__init__.py'),
                                              submodule_search_locations=[CFG_DIR])
pkg = importlib.util.module_from_spec(spec)
sys.modules['amd_configs'] = pkg
spec.loader.exec_module(pkg)


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or '?'


def run(name, episodes, calls):
    cfg = importlib.import_module('amd_configs.' + name).get_config(0)
    env = environment.Environment(**cfg)
    np.random.seed(0)
    t_reset, t_step, n = 0.0, 0.0, 0
    sprites = 0
    for _ in range(episodes):
        t0 = time.perf_counter()
        env.reset()
        t_reset += time.perf_counter() - t0
        sprites = sum(len(v) for v in env.state.values())
        t0 = time.perf_counter()
        for _ in range(calls):
            ts = env.step(env.action_space.random_action())
            n += 1
            if ts.last():
                break
        t_step += time.perf_counter() - t0
    print('  %-26s %3d sprites  %8.2f env steps/s  (%7.1f ms per step, %d calls)   reset %7.1f ms'
          % (name, sprites, n / t_step, 1e3 * t_step / n, n, 1e3 * t_reset / episodes))


print('reference Python on the scaled BASELINE workloads: one env, one core, 64 x 64 PIL observer, random actions of the config.s own action space')
print('host: %s; python %s, numpy %s' % (cpu_model(), platform.python_version(), np.__version__))
run('colliding_predators_32', episodes=3, calls=40)
run('falling_balls_64', episodes=2, calls=40)
