"""Seeded ray batches shared by the CPU and GPU parity tests."""
import numpy as np

import oracle_lib as O


def primary_rays(scene, width, height, step=1, seed=7):
    """Jittered camera rays through the oracle's camera (main.cpp:88-95)."""
    cam = scene.flat.contents.camera
    rng = np.random.default_rng(seed)
    org, dirs = [], []
    for i in range(0, height, step):
        for j in range(0, width, step):
            o, d = O.camera_ray(cam, width, height, i, j, float(np.float32(rng.random())), float(np.float32(rng.random())))
            org.append(o)
            dirs.append(d)
    return np.array(org, np.float32), np.array(dirs, np.float32)


def random_rays(n, lo, hi, seed=11):
    """Incoherent rays: origins uniform in the box [lo,hi], directions uniform on the sphere."""
    rng = np.random.default_rng(seed)
    org = (rng.random((n, 3)) * (np.asarray(hi) - np.asarray(lo)) + np.asarray(lo)).astype(np.float32)
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return org, v.astype(np.float32)


def scene_bounds(scene):
    v = scene.arrays()["tri_v"].reshape(-1, 3)
    return v.min(0), v.max(0)


def adversarial_rays(scene, n, seed=31):
    """Rays that stress the slab test's corner cases: direction components that are exactly 0 (1/d = inf,
    0 * inf = NaN when the origin sits on a box plane), origins exactly on vertex coordinates (the planes
    of the tight boxes) and on wall planes, and directions along the axes."""
    rng = np.random.default_rng(seed)
    v = scene.arrays()["tri_v"].reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    org = (rng.random((n, 3)) * (hi - lo) + lo).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    k = np.arange(n)
    # a third: origin coordinates copied from random vertices (exactly on box planes)
    pick = v[rng.integers(0, len(v), n)]
    on = (k % 3) == 0
    ax = rng.integers(0, 3, n)
    org[on, ax[on]] = pick[on, ax[on]]
    # zero out one or two direction components, some with -0.0
    z1 = (k % 2) == 0
    a1 = rng.integers(0, 3, n)
    d[z1, a1[z1]] = np.where(rng.random(z1.sum()) < 0.5, np.float32(0.0), np.float32(-0.0))
    z2 = (k % 5) == 0
    a2 = (a1 + 1) % 3
    d[z2, a2[z2]] = 0.0
    # the zeroed axis of an on-plane origin: make them coincide for half of those
    both = on & z1 & (rng.random(n) < 0.5)
    org[both, a1[both]] = pick[both, a1[both]]
    bad = np.abs(d).sum(1) == 0
    d[bad] = (1.0, 0.0, 0.0)
    return org, d


def non_finite_rays(scene, n, seed=3):
    """Rays a caller of trt_trace_closest can hand over but no camera produces: NaN, +-inf, +-1e38, a denormal or a signed zero in one coordinate of the origin
    (every third ray) or of the direction (every second), and the zero vector as a direction (every fiftieth)."""
    lo, hi = scene_bounds(scene)
    org, d = random_rays(n, lo, hi, seed=seed)
    rng = np.random.default_rng(seed)
    vals = np.array([np.nan, np.inf, -np.inf, 1e38, -1e38, 3e-39, 0.0, -0.0], np.float32)
    k = np.arange(n)
    for arr, m in ((org, 3), (d, 2)):
        sel = (k % m) == 0
        arr[sel, rng.integers(0, 3, sel.sum())] = vals[rng.integers(0, len(vals), sel.sum())]
    d[k % 50 == 0] = 0.0
    return org, d


def grazing_rays(flat, n, seed=5):
    """Rays that meet a random triangle of the scene at 1e-5 .. 1e-2 rad from its plane, from 0.1 .. 16 units away: the rays for which a
    Moller-Trumbore distance is least accurate (the case behind the rule 'a hit in front of its own leaf's box does not count')."""
    f = flat.contents
    nt = f.n_tris
    tv = np.ctypeslib.as_array(f.tri_v, shape=(nt * 9,)).reshape(nt, 3, 3).astype(np.float64)
    rng = np.random.default_rng(seed)
    ti = rng.integers(0, nt, n)
    b = rng.random((n, 3))
    b /= b.sum(1, keepdims=True)
    P = (tv[ti] * b[:, :, None]).sum(1)
    e1 = tv[ti, 1] - tv[ti, 0]
    e2 = tv[ti, 2] - tv[ti, 0]
    N = np.cross(e1, e2)
    N /= np.linalg.norm(N, axis=1, keepdims=True) + 1e-300
    tang = e1 * rng.normal(size=(n, 1)) + e2 * rng.normal(size=(n, 1))
    tang /= np.linalg.norm(tang, axis=1, keepdims=True) + 1e-300
    ang = 10.0 ** rng.uniform(-5.2, -2.0, (n, 1)) * rng.choice([-1.0, 1.0], (n, 1))
    d = tang + N * ang
    d /= np.linalg.norm(d, axis=1, keepdims=True) + 1e-300
    o = P - d * 10.0 ** rng.uniform(-1, 1.2, (n, 1))
    ok = np.isfinite(o).all(1) & np.isfinite(d).all(1) & (np.abs(d).sum(1) > 0)
    return np.ascontiguousarray(o[ok].astype(np.float32)), np.ascontiguousarray(d[ok].astype(np.float32))
