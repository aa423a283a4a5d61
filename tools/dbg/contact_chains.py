"""How much concurrency is there among the contacts of one sub-step?  ("resolve sprite-disjoint contacts
concurrently".)  The oracle logs every contact it resolves, in order (oracle_contact_log); contacts that share no sprite commute,
so the depth of the dependency chains (a contact waits for every earlier contact of the same sub-step that shares a sprite with
it) is the number of ROUNDS a perfectly parallel resolver would still need.  contacts / rounds bounds the speed-up of contact
resolution.  CPU only: OMP_NUM_THREADS=1 python tools/dbg/contact_chains.py [envs [warm-up steps]]"""
import ctypes, os, sys
os.environ['OMP_NUM_THREADS'] = '1'
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(R, 'tests')); sys.path.insert(0, os.path.join(R, 'moog.github.io_amd'))
import numpy as np, helpers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 40
c = helpers.compiled('colliding_predators_32')
o = helpers.OracleEnv(c, n_envs=n, seed=5)
o.reset(render=False)
rs = np.random.RandomState(1)
for _ in range(warm):
    o.step(rs.uniform(-1, 1, size=(n, 2)), render=False)
lib = helpers.oracle()
buf = np.zeros(8_000_000, np.int32)
lib.oracle_contact_log(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), len(buf))
o.step(rs.uniform(-1, 1, size=(n, 2)), render=False)
m = lib.oracle_contact_log_count()
lib.oracle_contact_log(None, 0)
log = buf[:m].reshape(-1, 2)
per_env = {}
cur, sub = None, None
for a, b in log:
    if a == -1:
        cur = per_env.setdefault(int(b), [])
        cur.append([])
    elif a < -1:
        continue   # (outcome records of tools/dbg/contact_stats.py)
    elif cur is not None:
        cur[-1].append((int(a), int(b)))
contacts, rounds = [], []
for env, subs in per_env.items():
    nc = nr = 0
    for lst in subs:
        level = {}
        depth = 0
        for s0, s1 in lst:
            l = 1 + max(level.get(s0, 0), level.get(s1, 0))
            level[s0] = level[s1] = l
            depth = max(depth, l)
        nc += len(lst); nr += depth
    contacts.append(nc); rounds.append(nr)
contacts, rounds = np.array(contacts), np.array(rounds)
order = np.argsort(-contacts)
heavy = order[:max(1, n // 100)]
print('colliding_predators_32, %d envs, one call after %d warm-up calls (K = 10 sub-steps)' % (n, warm))
print('contacts resolved per env and call: mean %.1f, heaviest 1 %% %.1f, max %d' % (contacts.mean(), contacts[heavy].mean(), contacts.max()))
print('rounds a perfectly parallel resolver needs (chains through shared sprites): mean %.1f, heaviest 1 %% %.1f' % (rounds.mean(), rounds[heavy].mean()))
print('contacts / rounds: all envs %.2f, heaviest 1 %% %.2f  (= upper bound on the speed-up of contact resolution alone)' % (
    contacts.sum() / max(rounds.sum(), 1), contacts[heavy].sum() / max(rounds[heavy].sum(), 1)))
