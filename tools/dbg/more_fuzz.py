"""Runs tests/test_gpu_fuzz.py's differential fuzz (random configs, engine vs oracle) on seeds beyond the committed 0..63."""
import os, sys, traceback
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(R, 'moog.github.io_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import test_gpu_fuzz as t
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(lo, hi):
    try:
        t.test_random_config_engine_vs_oracle(seed)
    except Exception as e:   # noqa: BLE001
        bad.append((seed, type(e).__name__, str(e)[:120]))
print('seeds %d..%d: %d clean, failures: %s' % (lo, hi - 1, hi - lo - len(bad), bad[:10]))
