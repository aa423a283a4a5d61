"""Times the step kernel with parts disabled (MOOG_STEP_DEBUG bitmask; results are wrong, timing only)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'moog.github.io_amd'))
import torch
from moog import environment, _abi
from moog_demos import example_configs
name = sys.argv[1] if len(sys.argv) > 1 else 'colliding_predators_32'
env = environment.BatchedEnvironment(num_envs=4096, seed=1, **example_configs.load(name))
env.check_faults = False
env.reset()
for _ in range(10):
    env.step(env.random_action())
snap = (env.state_f64.clone(), env.state_i32.clone())
for dbg in (0, 1, 2, 3, 4, 8, 16):
    env.set_debug(dbg, 0)
    env.state_f64.copy_(snap[0]); env.state_i32.copy_(snap[1])
    for _ in range(2):
        env.physics_step()
    env.set_timing(True); env.kernel_time(_abi.MOOG_K_STEP)
    for _ in range(10):
        env.state_f64.copy_(snap[0]); env.state_i32.copy_(snap[1])
        env.physics_step()
    ms, n = env.kernel_time(_abi.MOOG_K_STEP)
    env.set_timing(False)
    print('dbg %d: physics kernel %.0f us' % (dbg, ms / n * 1e3))
