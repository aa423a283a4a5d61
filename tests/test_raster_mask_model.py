"""Host model of the mask rasteriser (moog.github.io_amd/csrc/moog_raster_mask_core.h): the kernel's own phase
functions compiled with g++ and run thread by thread (tests/csrc/raster_mask_model.cpp), so that the algorithm --
Pillow's polygon fill without crossing lists -- is checked on the CPU: against the Pillow corpus, against the oracle's
restatement of ImagingDrawPolygon on random and degenerate polygons, and as whole frames (several passes per frame,
static prefix) against the oracle renderer.  The GPU tests (tests/test_gpu_parity.py) check the kernel itself."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import helpers
from moog import _abi

SRC = os.path.join(helpers.REPO, 'tests', 'csrc', 'raster_mask_model.cpp')
CORE = os.path.join(helpers.REPO, 'moog.github.io_amd', 'csrc', 'moog_raster_mask_core.h')
BUILD = os.path.join(helpers.REPO, 'tests', '_build')
SO = os.path.join(BUILD, 'libraster_mask_model.so')
_P = ctypes.POINTER


@pytest.fixture(scope='module')
def model():
    os.makedirs(BUILD, exist_ok=True)
    hdr = os.path.join(helpers.REPO, 'include', 'moog_engine.h')
    draw = os.path.join(os.path.dirname(CORE), 'moog_draw_record.h')
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(p) for p in (SRC, CORE, draw, hdr)):
        tmp = SO + '.%d.tmp' % os.getpid()   # (xdist workers may build at the same time)
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-Wall',
                               '-Wno-unused-function', SRC, '-o', tmp])
        os.replace(tmp, SO)
    return ctypes.CDLL(SO)


def model_polygon(m, xy, W, H, mode, stats):
    xy = np.ascontiguousarray(xy, np.int32)
    cov = np.zeros((H, W), np.uint8)
    rc = m.rm_model_polygon(xy.ctypes.data_as(_P(ctypes.c_int)), len(xy), W, H, cov.ctypes.data_as(_P(ctypes.c_uint8)),
                            mode, stats.ctypes.data_as(_P(ctypes.c_longlong)))
    assert rc == 0
    return cov


def oracle_polygon(xy, W, H):
    xy = np.ascontiguousarray(xy, np.int32)
    img = np.zeros((H, W, 3), np.uint8)
    ink = (ctypes.c_uint8 * 4)(255, 255, 255, 255)
    helpers.oracle().oracle_draw_polygon(img.ctypes.data_as(_P(ctypes.c_uint8)), W, H, len(xy),
                                         xy.ctypes.data_as(_P(ctypes.c_int)), ink)
    return (img[:, :, 0] != 0).astype(np.uint8)


def test_model_vs_pillow_corpus(model):
    """The 3000 polygons Pillow itself filled (tests/golden/raster.npz): fast row routine and generic routine."""
    z = dict(np.load(os.path.join(helpers.GOLDEN, 'raster.npz')))
    st = np.zeros(4, np.int64)
    for mode in (0, 1):
        bad = []
        for k in range(len(z['nv'])):
            W, nv = int(z['size'][k]), int(z['nv'][k])
            cov = model_polygon(model, z['xy'][k, :nv], W, W, mode, st)
            if not np.array_equal(cov, (z['red'][k, :W, :W] != z['bg'][0]).astype(np.uint8)):
                bad.append(k)
        assert not bad, ('polygons that differ from Pillow', mode, bad[:10], len(bad))


def random_polygon(rs, W, kind):
    nv = int(rs.randint(1, 33)) if kind != 8 else int(rs.randint(3, 33))
    if kind <= 2:      # lattices: coinciding vertices, zero-width spikes, runs of horizontal edges, several fix-ups per row
        step = (1, 2, 3)[kind]
        ox, oy = rs.randint(-4, W - 4, size=2)
        xy = np.stack([ox + step * rs.randint(0, 5, size=nv), oy + step * rs.randint(0, 5, size=nv)], 1)
    elif kind == 3:    # comb: many crossings per row
        t = np.arange(nv)
        xy = np.stack([rs.randint(0, W // 2) + 2 * t,
                       np.where(t % 2 == 0, rs.randint(0, W // 2), rs.randint(W // 2, W)) + rs.randint(-2, 3, size=nv)], 1)
    elif kind == 4:    # far off-canvas vertices mixed with near ones
        xy = rs.randint(-10, W + 10, size=(nv, 2))
        far = rs.rand(nv) < 0.3
        xy[far] = rs.choice([-30000, -17000, 17000, 30000], size=(int(far.sum()), 2))
    elif kind == 5:    # all vertices on one or two rows: rows of nothing but horizontal heads
        xy = np.stack([rs.randint(-5, W + 5, size=nv), rs.randint(0, W) + rs.randint(0, 2, size=nv)], 1)
    elif kind == 6:    # straddling a canvas border
        c = rs.choice([-2, 0, W - 3, W])
        xy = np.stack([c + rs.randint(-6, 7, size=nv), rs.choice([-2, 0, W - 3, W]) + rs.randint(-6, 7, size=nv)], 1)
    elif kind == 7:
        xy = rs.randint(-8, W + 8, size=(nv, 2))
    elif kind == 8:    # small regular polygons and stars at random angles, like the sprites of the workloads
        r = rs.uniform(1.0, 6.0)
        c = rs.uniform(-3, W + 3, size=2)
        th = rs.uniform(0, 2 * np.pi) + 2 * np.pi * np.arange(nv) / nv
        rad = r * (np.where(np.arange(nv) % 2 == 0, 1.0, rs.uniform(0.3, 1.0)) if rs.rand() < 0.5 else 1.0)
        xy = np.trunc(np.stack([c[0] + rad * np.cos(th) * rs.uniform(0.6, 1.4), c[1] + rad * np.sin(th)], 1)).astype(np.int64)
    else:              # a tiny lattice with many repeated points (thin arms)
        ox, oy = rs.randint(-2, W - 2, size=2)
        xy = np.stack([ox + rs.randint(0, 4, size=nv), oy + rs.randint(0, 8, size=nv)], 1)
    return np.ascontiguousarray(xy, np.int32)


@pytest.mark.parametrize('seed', [0, 1])
def test_model_vs_oracle_fuzz(model, seed):
    """Random and degenerate polygons (the families of test_raster_degenerate_polygon_fuzz and more) against the oracle's
    ImagingDrawPolygon, row routine and generic routine alike; the generic routine must stay the rare path."""
    rs = np.random.RandomState(4242 + seed)
    st, stg = np.zeros(4, np.int64), np.zeros(4, np.int64)
    bad = []
    for it in range(30000):
        W = 64 if rs.rand() < 0.6 else (128 if rs.rand() < 0.7 else int(rs.choice([16, 48, 80, 112])))
        xy = random_polygon(rs, W, it % 10)
        ref = oracle_polygon(xy, W, W)
        if not np.array_equal(model_polygon(model, xy, W, W, 0, st), ref):
            bad.append((it % 10, W, xy.tolist()))
        if it % 7 == 0 and not np.array_equal(model_polygon(model, xy, W, W, 1, stg), ref):
            bad.append(('generic', it % 10, W, xy.tolist()))
    assert not bad, (bad[:3], len(bad))
    assert st[1] < 0.01 * st[0], ('rows sent to the generic routine', int(st[1]), int(st[0]))


def model_frames(m, c, f64, i32, cap_rows, static=None, threads=128, compact=0):
    P = c.program
    n = f64.shape[0]
    W, H = (P.render.width + 15) & ~15, P.render.height
    img = np.zeros((n, H, W, 3), np.uint8)
    st = np.zeros(16, np.int64)
    ns, nsv, sf, sq, sbg = static if static is not None else (0, 0, None, None, None)
    dp, ip, bp = _P(ctypes.c_double), _P(ctypes.c_int32), _P(ctypes.c_uint8)
    rc = m.rm_model_frames(ctypes.byref(P), f64.ctypes.data_as(dp), i32.ctypes.data_as(ip), n, img.ctypes.data_as(bp), threads,
                           cap_rows, ns, nsv, None if sf is None else sf.ctypes.data_as(dp),
                           None if sq is None else sq.ctypes.data_as(ip), None if sbg is None else sbg.ctypes.data_as(bp),
                           None, st.ctypes.data_as(_P(ctypes.c_longlong)), compact)
    assert rc == 0, rc
    return img[:, :, :P.render.width], st


@pytest.mark.parametrize('name,cap_rows', [('colliding_predators_32', 192), ('colliding_predators_32', 64), ('functional_maze', 128),
                                           ('falling_balls_64', 100), ('pong', 192), ('cleanup', 128),
                                           ('chase_avoid_torus', 192), ('chase_avoid_torus', 64), ('match_to_sample_l3', 192),
                                           ('parallelogram_catch', 96), ('multi_tracking_with_feature_l3', 192), ('first_person_predators_prey', 192)])
def test_model_frames_vs_oracle(model, name, cap_rows):
    """Whole frames through the kernel's phases (the row records capped so that frames take several passes),
    against the oracle renderer: states of a few steps of the oracle's own simulation.  chase_avoid_torus: nine copies per
    sprite (polygon_modifiers.py:88-97), the visible ones become items."""
    c = helpers.compiled(name)
    n = 12
    o = helpers.OracleEnv(c, n_envs=n, seed=11)
    o.reset()
    rs = np.random.RandomState(2)
    P = c.program
    grid = P.action.kind == _abi.MOOG_ACTION_GRID and P.n_actions <= 1
    shape = (n, P.n_actions, 2) if P.n_actions > 1 else (n, 2)
    passes = 0
    for k in range(4):
        o.step(rs.randint(0, 5, size=n) if grid else rs.uniform(-1, 1, size=shape), render=False)
        ref = o.render().copy()
        img, st = model_frames(model, c, o.f64, o.i32, cap_rows)
        passes += int(st[2])
        bad = np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0]
        assert bad.size == 0, ('frames differ at step %d' % k, bad[:8].tolist())
        img, st = model_frames(model, c, o.f64, o.i32, cap_rows, compact=1)   # 4-byte edge records (RmEdgesCompact)
        bad = np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0]
        assert bad.size == 0, ('frames differ at step %d with compact edge records' % k, bad[:8].tolist())
    if cap_rows < 192 and name not in ('chase_avoid_torus', 'parallelogram_catch'):
        assert passes > 4 * n, 'the capped records were meant to force several passes per frame'


def test_model_static_prefix(model):
    """Frames composed on top of a cached picture of the leading sprites (the walls) when those equal the reference
    record bit for bit; a frame whose wall was moved by one ulp / recoloured / removed draws everything itself."""
    c = helpers.compiled('colliding_predators_32')
    P, L = c.program, c.layout
    n = 16
    o = helpers.OracleEnv(c, n_envs=n, seed=5)
    o.reset()
    o.step(np.zeros((n, 2)), render=False)
    ns = 4
    nsv = int(P.slot_voff[ns])
    sf, sq = o.f64[0].copy(), o.i32[0].copy()   # the reference record and its picture: env 0 with every other sprite gone
    o2 = helpers.OracleEnv(c, n_envs=1)
    o2.f64[0], o2.i32[0] = sf, sq
    o2.i32[0, L.o_flags + ns:L.o_flags + L.S] &= ~1
    sbg = np.ascontiguousarray(o2.render()[0].copy())
    f, q = o.f64, o.i32
    v0 = L.o_verts + 2 * int(P.slot_voff[1])
    f[1, v0] = np.nextafter(f[1, v0], 2.0)
    f[2, L.o_color + 3 * 2 + 2] = 0.9
    q[3, L.o_flags + 0] &= ~1
    q[4, L.o_opacity + 3] = 100
    ref = o.render().copy()
    img, st = model_frames(model, c, f, q, 192, static=(ns, nsv, sf, sq, sbg))
    bad = np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0]
    assert bad.size == 0, ('frames differ', bad.tolist())


def test_model_torus_copies(model):
    """Torus frames with the sprites moved to where the copies matter: across an edge, across a corner, far outside (every
    copy off the canvas but one), exactly on the edge, and blown up beyond the canvas (all nine copies visible, overlapping
    each other in copy order); some sprites have a coordinate that is NaN, infinite or beyond the range of an int in every vertex."""
    c = helpers.compiled('chase_avoid_torus')
    P, L = c.program, c.layout
    n = 64
    o = helpers.OracleEnv(c, n_envs=n, seed=3)
    o.reset()
    o.step(np.zeros((n,) + ((P.n_actions, 2) if P.n_actions > 1 else (2,))), render=False)
    rs = np.random.RandomState(8)
    f = o.f64
    for e in range(n):
        for s in range(P.n_slots):
            v0, nv = L.o_verts + 2 * int(P.slot_voff[s]), int(o.i32[e, L.o_nverts + s])
            if nv <= 0:
                continue
            v = f[e, v0:v0 + 2 * nv].reshape(nv, 2)
            ctr = v.mean(axis=0)
            kind = (e + s) % 8
            scale = (1.0, 1.0, 2.5, 1.0, 9.0, 30.0, 0.3, 1.0)[kind]
            target = ((0.0, rs.rand()), (0.0, 0.0), (rs.rand(), 1.0), (1.0, 1.0), rs.rand(2), rs.rand(2), (0.999, 0.001),
                      rs.uniform(-1.3, 2.3, size=2))[kind]
            v[:] = (v - ctr) * scale + np.asarray(target, np.float64)
            if e % 9 == 4 and s % 2 == 0:     # coordinates Pillow's (int) cannot hold / NaNs: INT_MIN, not monotone.  (Every
                # vertex of the sprite: a polygon with SOME such points is outside what the engine reproduces -- Pillow
                # subtracts INT_MIN coordinates with overflow; the engine clamps to +-32000 -- with or without a torus.)
                v[:, rs.randint(2)] = (np.nan, 3.0e9, -7.0e11, np.inf)[(e // 9 + s) % 4]
    ref = o.render().copy()
    img, st = model_frames(model, c, o.f64, o.i32, 192)
    bad = np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0]
    assert bad.size == 0, ('frames differ', bad[:8].tolist())
    img, st = model_frames(model, c, o.f64, o.i32, 64)   # several passes
    bad = np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0]
    assert bad.size == 0, ('frames differ (capped rows)', bad[:8].tolist())
    assert st[2] > n


def long_polygon(rs, W, nv, kind):
    """Integer canvas points of a polygon with nv > 32 vertices."""
    if kind == 0:      # lattice: repeated points, spikes, runs of horizontal edges, corners sharing rows
        step = int(rs.randint(1, 4))
        ox, oy = rs.randint(-4, W - 4, size=2)
        xy = np.stack([ox + step * rs.randint(0, 9, size=nv), oy + step * rs.randint(0, 9, size=nv)], 1)
    elif kind == 1:    # comb: dozens of crossings per row
        t = np.arange(nv)
        xy = np.stack([rs.randint(-3, 4) + (W * t) // nv, np.where(t % 2 == 0, rs.randint(0, W // 2), rs.randint(W // 2, W)) + rs.randint(-2, 3, size=nv)], 1)
    elif kind == 2:    # annulus: outer ring one way, inner ring back, joined by a seam (shapes.annulus_vertices)
        n0 = nv // 2
        n1 = nv - n0
        c = rs.uniform(0.2 * W, 0.8 * W, size=2)
        r0, r1 = rs.uniform(0.2 * W, 0.7 * W), rs.uniform(0.02 * W, 0.18 * W)
        th0, th1 = np.linspace(0, 2 * np.pi, n0), np.linspace(2 * np.pi, 0, n1)
        xy = np.concatenate([np.stack([c[0] + r0 * np.cos(th0), c[1] + r0 * np.sin(th0)], 1),
                             np.stack([c[0] + r1 * np.cos(th1), c[1] + r1 * np.sin(th1)], 1)])
    elif kind == 3:    # star
        th = rs.uniform(0, 2 * np.pi) + 2 * np.pi * np.arange(nv) / nv
        rad = np.where(np.arange(nv) % 2 == 0, rs.uniform(0.3, 0.6) * W, rs.uniform(0.05, 0.3) * W)
        c = rs.uniform(0.3 * W, 0.7 * W, size=2)
        xy = np.stack([c[0] + rad * np.cos(th), c[1] + rad * np.sin(th)], 1)
    elif kind == 4:    # random scatter straddling the canvas
        xy = rs.randint(-10, W + 10, size=(nv, 2))
    else:              # all on two rows: heads only
        xy = np.stack([rs.randint(-5, W + 5, size=nv), rs.randint(0, W) + rs.randint(0, 2, size=nv)], 1)
    return np.trunc(xy).astype(np.int64)


def test_model_long_polygons(model):
    """Polygons of 33 .. 102 vertices (the reference's 102-vertex annuli and worse) take the cooperative row routine
    (rm_p4_big): planted in the big slots of match_to_sample's frames beside the ordinary sprites, against the oracle."""
    c = helpers.compiled('match_to_sample_l3')
    P, L = c.program, c.layout
    n = 240
    o = helpers.OracleEnv(c, n_envs=n, seed=9)
    o.reset()
    W = P.render.width
    big = [s for s in range(P.n_slots) if P.slot_vcap[s] > 32]
    assert big
    rs = np.random.RandomState(77)
    for e in range(n):
        for s in big:
            nv = int(rs.randint(33, P.slot_vcap[s] + 1))
            pts = long_polygon(rs, W, nv, (e + s) % 6)
            # world coordinates whose scaled (int) is the wanted point: (p + 0.5) / W for p >= 0, (p - 0.5) / W below zero
            wv = (pts + np.where(pts >= 0, 0.5, -0.5)) / float(W)
            v0 = L.o_verts + 2 * int(P.slot_voff[s])
            o.f64[e, v0:v0 + 2 * nv] = wv.reshape(-1)
            o.i32[e, L.o_nverts + s] = nv
            o.i32[e, L.o_flags + s] |= 1
            o.i32[e, L.o_opacity + s] = (255, 128)[e % 2]
    ref = o.render().copy()
    img, st = model_frames(model, c, o.f64, o.i32, 192)
    bad = np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0]
    assert bad.size == 0, ('frames differ', bad[:8].tolist(), int(bad.size))
    img, st = model_frames(model, c, o.f64, o.i32, 64)
    bad = np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0]
    assert bad.size == 0, ('frames differ (capped rows)', bad[:8].tolist(), int(bad.size))
    img, st = model_frames(model, c, o.f64, o.i32, 192, compact=1)
    bad = np.nonzero((img != ref).reshape(n, -1).any(axis=1))[0]
    assert bad.size == 0, ('frames differ (compact edge records)', bad[:8].tolist(), int(bad.size))
