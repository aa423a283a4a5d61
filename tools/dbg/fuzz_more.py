"""More seeds of tests/test_gpu_fuzz.py's engine-vs-oracle fuzz (not part of the suite): python tools/dbg/fuzz_more.py 64 400"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(R, 'tests')); sys.path.insert(0, os.path.join(R, 'moog.github.io_amd'))
import test_gpu_fuzz as t
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    try:
        t.test_random_config_engine_vs_oracle(seed)
    except NotImplementedError as exc:   # (a random config the lowering refuses is not a failure)
        print('seed %d refused: %s' % (seed, str(exc)[:80]))
    except Exception as exc:
        bad += 1
        print('seed %d FAILED: %r' % (seed, exc))
print('seeds %d..%d: %d failures' % (lo, hi - 1, bad))
