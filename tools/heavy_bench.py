"""Latency of the step kernel's slowest envs, replayed in isolation.

The step kernel lasts as long as its slowest env, so this is the number to optimise: the pre-step records
and actions of the heaviest envs of a few calls are captured once (`capture`), and `bench` replays exactly those steps --
tiled over a small batch (one wavefront per CU: the env's own dependent chain) and over a full batch (3 waves per SIMD) --
and prints their cycle counts (s_memtime at entry / exit of the wavefront, MOOG_STEP_DEBUG=128).  The work replayed is
identical from run to run (the step path draws no random numbers on this workload), so builds can be compared to 0.1 %.

    python tools/heavy_bench.py capture [workload]      -> gpurun_out/heavy_<workload>.npz  (copy to tools/ubench/)
    python tools/heavy_bench.py bench [workload] [--sections]
"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(R, 'moog.github.io_amd'))
import numpy as np, torch
from moog import environment
from moog_demos import example_configs

mode = sys.argv[1] if len(sys.argv) > 1 else 'bench'
name = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else 'colliding_predators_32'
TOP = 64


def path(d):
    return os.path.join(R, d, '%s_%s.npz' % ('sample' if 'RANDOM_SAMPLE' in os.environ else 'heavy', name))


def make(n):
    env = environment.BatchedEnvironment(num_envs=n, seed=1, layer_capacity=example_configs.capacity(name),
                                         **example_configs.load(name))
    env.check_faults = False
    return env


if mode == 'capture':
    n = int(os.environ.get('CAPTURE_ENVS', 4096))
    env = make(n)
    env.reset()
    env.set_debug(128, 0)
    F, Q, A, C = [], [], [], []
    AT = tuple(int(x) for x in os.environ.get('CAPTURE_AT', '25,40,50,60').split(','))
    for k in range(max(AT) + 2):
        a = env.random_action()
        if k in AT:
            f0, q0 = env.state_f64.clone(), env.state_i32.clone()
        ts = env.step(a)
        if k in AT:
            c = ts.discount.clone()
            c[ts.step_type == 0] = 0
            top = torch.argsort(-c)[:TOP] if 'RANDOM_SAMPLE' not in os.environ else torch.randperm(n, device=c.device)[:TOP]
            F.append(f0[top].cpu().numpy()); Q.append(q0[top].cpu().numpy()); A.append(a[top].cpu().numpy())
            C.append(c[top].cpu().numpy())
            print('call %d: heaviest %d envs: cycles mean %.0f max %.0f (batch mean %.0f)' % (k, TOP, C[-1].mean(), C[-1].max(), c.mean().item()))
    os.makedirs(os.path.join(R, 'gpurun_out'), exist_ok=True)
    np.savez_compressed(path('gpurun_out'), f64=np.concatenate(F), i32=np.concatenate(Q), act=np.concatenate(A), cycles=np.concatenate(C))
    print('saved', path('gpurun_out'))
    sys.exit(0)

src = next(path(d) for d in ('tools/ubench', 'tools/ubench/build', 'gpurun_out') if os.path.exists(path(d)))   # (big captures: build/, untracked)
d = np.load(src)
m = d['f64'].shape[0]
sections = '--sections' in sys.argv
for n in ([int(os.environ['HEAVY_ONLY'])] if os.environ.get('HEAVY_ONLY') else (256, 3072)):
    env = make(n)
    env.reset()
    idx = np.arange(n) % m
    f = torch.from_numpy(d['f64'][idx]).cuda(); q = torch.from_numpy(d['i32'][idx]).cuda(); a = torch.from_numpy(d['act'][idx]).cuda()
    sel = [0] + (list(range(1, 17)) if sections else [])
    out = {}
    for s in sel:
        env.set_debug(128 | (s << 8), 0)
        cyc = []
        for rep in range(3):
            env.state_f64.copy_(f); env.state_i32.copy_(q)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ts = env.step(a)
            e1.record()
            torch.cuda.synchronize()
            cyc.append((ts.discount.cpu().numpy().copy(), ts.reward.cpu().numpy().copy(), e0.elapsed_time(e1) * 1e3))
        out[s] = cyc[-1]
    c, r, us = out[0]
    cnt = r % 1e10
    print('%s x %d envs (%d distinct heavy envs): cycles mean %.0f  p50 %.0f  max %.0f   call %.0f us   path tests %.1f searches %.1f' % (
        name, n, m, c.mean(), np.median(c), c.max(), us, (cnt % 100000).mean(), (cnt // 100000).mean()))
    if os.environ.get('COUNT_PREFIX'):
        d3 = r // 1e10
        print('   batches per env-step %.1f, candidates in them %.1f, rejected symmetrically %.1f' % ((d3 % 1000).mean(), ((d3 // 1000) % 1000).mean(), (d3 // 1000000).mean()))
    if os.environ.get('MOOG_WATCH') == '1':   # section samples of the watcher wavefronts (a -DMOOG_WATCH build of the step kernel)
        import ctypes
        SECN = ['prologue', 'rules + action', 'force loop', 'same layer: broad rounds', 'same layer: list', 'same layer: batch formation',
                'narrow batch (4 x 16 lanes)', 'path test', 'search: containment', 'search: motion matrix', 'search: crossing rows',
                'search: finish', 'resolve_contact', 'make_disjoint', 'same layer: re-test after a contact', 'layer pair: scan',
                'layer pair: consume', 'integrate: poses', 'integrate: long lists', 'integrate: vertices', 'integrate: boxes',
                'task reward', 'store', 'collision_step control', 'same layer: consume control', 'search: select']
        buf = (ctypes.c_int32 * (n * 32))()
        rc = env._lib.moog_engine_read_watch(env._handle, buf, 1)
        assert rc == 0, rc
        w = np.frombuffer(buf, dtype=np.int32).reshape(n, 32).astype(np.float64)
        tot = w.sum()
        print('   watcher samples: %.0f per env-step and env; share of the samples by section:' % (w.sum(1).mean() / 3))
        for k in np.argsort(-w.sum(0)):
            if w[:, k].sum() > 0:
                print('      %-36s %5.1f %%' % (SECN[k] if k < len(SECN) else 'section %d' % k, 100 * w[:, k].sum() / tot))
    if len(sel) > 1:
        names = {1: 'single path tests', 2: 'contact search', 3: 'make_disjoint', 4: 'resolve', 5: 'broad-phase scan', 6: 'integrate',
                 7: 'physics total', 8: 'collisions total', 9: 'candidate batches', 10: 'record load + boxes', 11: 'rules + action',
                 12: 'task reward', 13: ' search: containment', 14: ' search: motion matrix', 15: ' search: crossing rows'}
        names[16] = ' search: finish'
        for s in sel[1:]:
            v = out[s][1]
            print('   %-22s %9.0f (%4.1f %%)' % (names[s], v.mean(), 100 * v.mean() / c.mean()))
    env.close()
