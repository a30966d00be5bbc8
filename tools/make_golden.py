#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (oracle/liboracle.so) in this container.

The reference holds no golden vectors for this path and cannot be run, so the
fixtures are outputs of the build's own CPU restatement: small float images and ray/hit batches
on the three shipped scenes.  Both the HIP path (tests -m gpu) and the hostsim (CPU) are compared
against them bit for bit.  Re-run only when the formulation changes.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
import raygen  # noqa: E402
import tinyraytracing_amd as T  # noqa: E402

CASES = [("back", 48, 48, 16, T.SEED_BACK), ("veach-mis", 64, 36, 8, 0x5EED0002), ("staircase", 64, 36, 8, T.SEED_STAIRCASE)]

out_dir = os.path.join(ROOT, "tests", "golden")
os.makedirs(out_dir, exist_ok=True)
for name, w, h, spp, seed in CASES:
    s = T.Scene.named(name, w, h)
    p = T.make_params(w, h, spp, seed)
    img, st = O.render(s.flat, p)
    lo, hi = raygen.scene_bounds(s)
    o1, d1 = raygen.primary_rays(s, w, h, step=2)
    o2, d2 = raygen.random_rays(1024, lo, hi)
    org, dirs = np.vstack([o1, o2]), np.vstack([d1, d2])
    t, tri, uv = O.trace(s.flat, org, dirs)
    np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), image=img, width=w, height=h, spp=spp, seed=seed,
                        rays=np.array([st.rays_camera, st.rays_shadow, st.rays_indirect], np.uint64),
                        org=org, dir=dirs, t=t, tri=tri, uv=uv)
    print(name, img.shape, img.mean(), len(t), (tri >= 0).mean())
