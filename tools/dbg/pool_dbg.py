"""Timeline of the reset pool's fills: python tools/dbg/pool_dbg.py [config] [N]"""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'moog.github.io_amd'))
import torch
from moog import environment
from moog_demos import example_configs
name = sys.argv[1] if len(sys.argv) > 1 else 'bounce_box_contact_prediction'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
env = environment.BatchedEnvironment(num_envs=n, seed=1, layer_capacity=example_configs.capacity(name), reset_pool=True, **example_configs.load(name))
env.check_faults = False
torch.cuda.synchronize(); t = time.perf_counter()
env.reset()
torch.cuda.current_stream().synchronize(); t1 = time.perf_counter()
print('reset call (main stream): %.1f ms' % ((t1 - t) * 1e3))
torch.cuda.synchronize(); t2 = time.perf_counter()
print('first fill (device idle otherwise): %.1f ms more' % ((t2 - t1) * 1e3), env.reset_pool)
for k in range(40):
    t = time.perf_counter()
    env.step(env.random_action())
    torch.cuda.current_stream().synchronize(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    st = env.reset_pool
    print('call %2d: main stream %.2f ms, device idle after %.2f ms more  %s' % (k, (t1 - t) * 1e3, (t2 - t1) * 1e3, st), flush=True)
# free running
torch.cuda.synchronize(); t = time.perf_counter()
for k in range(300):
    env.step(env.random_action())
torch.cuda.current_stream().synchronize(); t1 = time.perf_counter()
print('300 calls free running: %.2f ms per call' % ((t1 - t) / 300 * 1e3), env.reset_pool)
