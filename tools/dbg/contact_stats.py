"""What do the contact searches of a call find?  The oracle logs every search of Collision.step (collisions.py:494-584) with its
outcome: no contact vectors (-> _make_disjoint), a contact (-> pop-out + impulse), or a contact in the FUTURE (no-op).  A FUTURE
pair keeps overlapping, so both of its ordered visits search again in every sub-step without changing anything.
CPU only: OMP_NUM_THREADS=1 python tools/dbg/contact_stats.py [envs [warm-up steps]]"""
import ctypes, os, sys
os.environ['OMP_NUM_THREADS'] = '1'
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(R, 'tests')); sys.path.insert(0, os.path.join(R, 'moog.github.io_amd'))
import numpy as np, helpers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 40
name = sys.argv[3] if len(sys.argv) > 3 else 'colliding_predators_32'
c = helpers.compiled(name)
o = helpers.OracleEnv(c, n_envs=n, seed=5)
o.reset(render=False)
rs = np.random.RandomState(1)
from moog import _abi
grid = c.program.action.kind == _abi.MOOG_ACTION_GRID
act = lambda: rs.randint(0, 5, size=n) if grid else rs.uniform(-1, 1, size=(n, 2))
for _ in range(warm):
    o.step(act(), render=False)
lib = helpers.oracle()
buf = np.zeros(16_000_000, np.int32)
lib.oracle_contact_log(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), len(buf))
o.step(act(), render=False)
m = lib.oracle_contact_log_count()
lib.oracle_contact_log(None, 0)
log = buf[:m].reshape(-1, 2)
stats = {}   # env -> [searches, none, ok, future, deep, mirrored-future-repeat]
cur = None
last_future = None
for a, b in log:
    if a == -1:
        cur = stats.setdefault(int(b), [0, 0, 0, 0, 0, 0, 0, 0, 0])
        last_future = set()
        last_none = set()
    elif a == -2:
        cur[4] += 1
    elif a in (-20, -21):   # _make_disjoint after a search without vectors: -20 left the state alone, -21 moved the pair
        s0, s1, depth = b & 255, (b >> 8) & 255, b >> 16
        if depth == 0:
            cur[6 + (a == -21)] += 1
            if a == -20:
                if (s1, s0) in last_none:
                    cur[8] += 1
                last_none.add((s0, s1))
        if a == -21:
            last_none = {p for p in last_none if s0 not in p and s1 not in p}
    elif a <= -3:
        st = -3 - a   # CV_NONE 0, CV_OK 1, CV_FUTURE 2
        s0, s1 = b & 255, (b >> 8) & 255
        cur[0] += 1
        cur[1 + st] += 1
        if st == 2:
            if (s1, s0) in last_future:
                cur[5] += 1
            last_future.add((s0, s1))
        else:   # a contact changed s0 / s1: their cached searches are stale
            last_future = {p for p in last_future if s0 not in p and s1 not in p}
            if st == 1:
                last_none = {p for p in last_none if s0 not in p and s1 not in p}
A = np.array(list(stats.values()), float)
order = np.argsort(-A[:, 0])
heavy = order[:max(1, n // 100)]
print('%s, %d envs, one call after %d warm-up calls' % (name, n, warm))
for nm, sel in (('all envs', slice(None)), ('heaviest 1 %', heavy)):
    v = A[sel].mean(0)
    print('  %-13s searches %.1f: no vectors %.1f, contact %.1f, FUTURE %.1f; at depth > 0: %.1f; FUTURE repeats of the mirrored pair with both sprites untouched: %.1f' % (
        (nm,) + tuple(v[:6])))
    print('  %-13s first searches (depth 0) without vectors: _make_disjoint left the state alone %.1f, moved the pair %.1f; of the former, repeats of the mirrored pair with both sprites untouched: %.1f' % (
        nm, v[6], v[7], v[8]))
