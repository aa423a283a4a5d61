"""Cycles per call of single device functions of the step path on the captured heavy envs (tools/heavy_bench.py capture):
    bash tools/fn_bench.sh  builds tools/ubench/build/libfn_bench.so (CPU) ; python tools/fn_bench.py runs it (GPU)."""
import ctypes, os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(R, 'moog.github.io_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
import helpers
name = 'colliding_predators_32'
lib = ctypes.CDLL(os.environ.get('FN_BENCH_LIB', os.path.join(R, 'tools', 'ubench', 'build', 'libfn_bench.so')))
c = helpers.compiled(name)
d = np.load(os.path.join(R, 'tools', 'ubench', '%s_%s.npz' % ('sample' if 'RANDOM_SAMPLE' in os.environ else 'heavy', name)))
m = d['f64'].shape[0]
FNS = [(9, 'empty loop (wsync)', 2000), (0, 'get_collision_vectors', 200), (1, 'path test (overlaps, prechecked)', 200),
       (2, 'narrow_reject_prefix (4 candidates)', 200), (4, 'integrate_all', 50), (5, 'apply_physics (one substep)', 10),
       (14, 'force loop headers alone', 100), (15, 'Drag on one layer alone', 100), (13, 'substep: force loop only', 10), (10, 'substep: forces + integrate only', 10), (11, 'substep: + broad phase, lists', 10), (12, 'substep: + narrow phase, no search', 10)]
ONLY = [int(x) for x in os.environ['FN_ONLY'].split(',')] if os.environ.get('FN_ONLY') else None
if ONLY:
    FNS = [f for f in FNS if f[0] in ONLY]
for n in ([int(os.environ['FN_ENVS'])] if os.environ.get('FN_ENVS') else (256, 3072)):
    idx = np.arange(n) % m
    f = torch.from_numpy(d['f64'][idx]).cuda(); q = torch.from_numpy(d['i32'][idx]).cuda()
    pairs = torch.full((n, 2), -1, dtype=torch.int32, device='cuda')
    out = torch.zeros((2 * n,), dtype=torch.float64, device='cuda')
    def run(which, iters):
        ff, qq = f.clone(), q.clone()
        rc = lib.moog_fn_bench(ctypes.byref(c.program), ctypes.c_void_p(ff.data_ptr()), ctypes.c_void_p(qq.data_ptr()), n, which, iters,
                               ctypes.c_void_p(pairs.data_ptr()), ctypes.c_void_p(out.data_ptr()))
        assert rc == 0, rc
        return out[:n].cpu().numpy().copy()
    run(-1, 1)
    has = (pairs[:, 0] >= 0).cpu().numpy()
    print('%d envs (%d with an overlapping predator pair)' % (n, has.sum()))
    for which, nm, iters in FNS:
        run(which, 2)
        v = run(which, iters)
        v = v[has] if which in (0, 1, 2) else v
        print('  %-40s cycles per call: mean %8.0f  p10 %8.0f  p90 %8.0f' % (nm, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
