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
