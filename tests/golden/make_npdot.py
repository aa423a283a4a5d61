"""Pins how the numpy the fixtures were recorded with rounds the reference's 2-vector dot products and norms
(moog/physics/collisions.py:223-228, 286, 322-335, 424-431, 641, 699-741; sprite.py:436, 464; ...):
  np.dot(a, b), float64          -> cblas_ddot: fma(a1, b1, a0 * b0)      ('dot64')
  np.linalg.norm(a), 1-D float64 -> sqrt(a.dot(a))                        ('norm64')
  np.linalg.norm(A, axis=1)      -> sqrt(a0 * a0 + a1 * a1), plain        ('norm64_axis')
  np.dot / norm of float32       -> plain float32 sums                    ('dot32', 'norm32')
Run where the golden recordings are made:  python tests/golden/make_npdot.py"""
import os
import numpy as np

rs = np.random.RandomState(20261002)
n = 4000
a = rs.normal(size=(n, 2)) * 10 ** rs.uniform(-4, 2, size=(n, 1))
b = rs.normal(size=(n, 2)) * 10 ** rs.uniform(-4, 2, size=(n, 1))
a32, b32 = a.astype(np.float32), b.astype(np.float32)
out = dict(
    a=a, b=b,
    dot64=np.array([np.dot(a[i], b[i]) for i in range(n)]),
    norm64=np.array([np.linalg.norm(a[i]) for i in range(n)]),
    norm64_axis=np.linalg.norm(a, axis=1),
    dot32=np.array([np.dot(a32[i], b32[i]) for i in range(n)], dtype=np.float32),
    norm32=np.array([np.linalg.norm(a32[i]) for i in range(n)], dtype=np.float32),
    numpy_version=np.array(np.__version__))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'npdot.npz'), **out)
print('written', n, 'cases, numpy', np.__version__)
