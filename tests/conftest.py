"""pytest setup: import paths and the `gpu` marker.

`-m "not gpu"` tests run on CPU only (oracle vs golden vectors, host lowering,
C-ABI symbol checks); `-m gpu` tests are the HIP-vs-oracle parity tests.
"""
import os
import sys

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
PKG = os.path.join(REPO, 'moog.github.io_amd')
for p in (PKG, os.path.dirname(os.path.abspath(__file__)), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
