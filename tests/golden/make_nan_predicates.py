"""Generates tests/golden/predicates_nan.npz: matplotlib's Path.intersects_path(filled=True) on polygons without a
finite vertex (a sprite whose position / angle went NaN or inf: every vertex is non-finite, PathNanRemover drops them
all and path_in_path() of an empty path is True) against finite polygons near and far, both ways round.

    MPLBACKEND=Agg python tests/golden/make_nan_predicates.py          (matplotlib 3.10.8)
"""
import os
import numpy as np
from matplotlib.path import Path

HERE = os.path.dirname(os.path.abspath(__file__))


def closed(v):
    v = np.asarray(v, float)
    return Path(np.vstack([v, v[:1]]))


def main():
    rs = np.random.RandomState(11)
    va, vb, na, nb, hit = [], [], [], [], []
    cap = 30

    def circle(n, cx, cy, r):
        t = np.linspace(0, 2 * np.pi, n, endpoint=False)
        return np.stack([cx + r * np.cos(t), cy + r * np.sin(t)], 1)

    def bad(n, kind):
        v = circle(n, rs.uniform(0, 1), rs.uniform(0, 1), 0.05)
        if kind == 0: v[:] = np.nan
        elif kind == 1: v[:, 0] = np.nan                 # x NaN, y finite: still no finite VERTEX
        elif kind == 2: v[:] = np.inf
        elif kind == 3: v[:, 1] = -np.inf
        else: v[:, 0] = np.nan; v[:, 1] = np.inf
        return v

    for k in range(400):
        n1, n2 = rs.choice([4, 5, 8, 18, 30]), rs.choice([4, 5, 8, 18, 30])
        fin = circle(n2, rs.uniform(-1, 2), rs.uniform(-1, 2), rs.uniform(0.02, 0.4))
        b = bad(n1, k % 5)
        mode = k % 3
        if mode == 0: a_, b_ = b, fin
        elif mode == 1: a_, b_ = fin, b
        else: a_, b_ = b, bad(n2, (k // 3) % 5)
        h = closed(a_).intersects_path(closed(b_), filled=True)
        pa = np.zeros((cap, 2)); pb = np.zeros((cap, 2))
        pa[:len(a_)] = a_; pb[:len(b_)] = b_
        va.append(pa); vb.append(pb); na.append(len(a_)); nb.append(len(b_)); hit.append(h)
    np.savez_compressed(os.path.join(HERE, 'predicates_nan.npz'), va=np.array(va), vb=np.array(vb), na=np.array(na, np.int32),
                        nb=np.array(nb, np.int32), hit=np.array(hit, np.uint8))
    print('cases', len(hit), 'hits', int(np.sum(hit)))


if __name__ == '__main__':
    main()
