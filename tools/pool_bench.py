"""Reset pool on / off for the configs whose state_initializer plays physics forward: python tools/pool_bench.py [N ...]
(wall-clock env-steps/s of `steps` calls with random actions after a warm-up, frames drawn; the pool's own counters)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'moog.github.io_amd'))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')   # (the reset pool's fills and nothing else here want more than the runtime's four; set before HIP initialises)
import torch
from moog import environment
from moog_demos import example_configs


def run(name, n, pool, steps, warm):
    env = environment.BatchedEnvironment(num_envs=n, seed=1, layer_capacity=example_configs.capacity(name), reset_pool=pool,
                                         **example_configs.load(name))
    env.check_faults = False
    env.reset()
    for _ in range(warm):
        env.step(env.random_action())
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps):
        env.step(env.random_action())
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    st = env.reset_pool
    faults = int((env.state_i32[:, env.layout.o_fault] != 0).sum().item())
    print('%-30s N=%5d pool=%-5s %9.0f env-steps/s  %7.1f us/call  faults %d  %s  %s' % (
        name, n, pool, n * steps / dt, dt / steps * 1e6, faults, st, env.reset_pool_refusal or ''), flush=True)
    env.close()


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'only':   # only <config> <envs>: the pooled run alone (tools/prof_pool.sh)
        run(sys.argv[2], int(sys.argv[3]), 'auto', 600, 100)
        sys.exit(0)
    sizes = [int(a) for a in sys.argv[1:]] or [1024, 4096]
    for name in ('bounce_box_contact_prediction', 'red_green_l1'):
        for n in sizes:
            run(name, n, False, 60, 10)
            run(name, n, 'auto', 600, 100)
