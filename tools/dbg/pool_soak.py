"""Longer runs of the reset pool's result-neutrality test (tests/test_gpu_parity.py) than the suite affords:
python tools/dbg/pool_soak.py [envs [calls]]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(R, 'moog.github.io_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import test_gpu_parity as t
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
for name in ('bounce_box_contact_prediction', 'red_green_l1', 'lookahead_zoo', 'combo_zoo'):
    t.test_reset_pool_is_result_neutral(name, n, calls, 1)
    print('%s: %d envs x %d calls with the pool equal the run without it' % (name, n, calls), flush=True)
