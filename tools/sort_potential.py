#!/usr/bin/env python3
"""How much would sorting the bounce queues buy?  Incoherent secondary rays of a scene (origins = hit points of random rays;
directions toward one light point = shadow rays, or uniformly random = diffuse bounces) traced by k_trace_closest through
trt_trace_closest in arrival order, and again sorted by the Morton code of the origin (and the direction octant for bounces).
Prints the kernel times.  usage: tools/sort_potential.py <scene> [n_rays]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinyraytracing_amd as T  # noqa: E402


def morton3(q):  # q: (n, 3) uint32 with 10 bits each
    def spread(x):
        x = x.astype(np.uint64) & 0x3FF
        x = (x | (x << 16)) & 0x30000FF
        x = (x | (x << 8)) & 0x300F00F
        x = (x | (x << 4)) & 0x30C30C3
        x = (x | (x << 2)) & 0x9249249
        return x
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def main():
    name = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000
    kw = {"n": 1_000_000} if name == "soup" else ({"n": 2_000_000} if name == "blob" else {})
    s = T.Scene.named(name, 64, 36, **kw)
    r = T.Renderer(s, 0)
    v = s.arrays()["tri_v"].reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    rng = np.random.default_rng(1)
    org = (rng.random((n, 3)) * (hi - lo) + lo).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t, tri, uv = r.trace_closest(org, d)
    hit = tri >= 0
    P = (org[hit] + d[hit] * t[hit, None]).astype(np.float32)
    m = len(P)
    light = (lo + (hi - lo) * np.array([0.5, 0.95, 0.5])).astype(np.float32)
    cases = {}
    ds = light[None, :] - P
    ds /= np.linalg.norm(ds, axis=1, keepdims=True)
    cases["shadow rays toward one point"] = (P, ds.astype(np.float32), False)
    db = rng.normal(size=(m, 3)).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    cases["diffuse bounce rays"] = (P, db, True)
    q = np.clip(((P - lo) / (hi - lo) * 1023.0), 0, 1023).astype(np.uint32)
    code = morton3(q)
    for label, (o, dd, use_oct) in cases.items():
        key = code.copy()
        if use_oct:
            octant = ((dd[:, 0] < 0).astype(np.uint64) | ((dd[:, 1] < 0).astype(np.uint64) << 1) | ((dd[:, 2] < 0).astype(np.uint64) << 2))
            key = (octant << 30) | code
        order = np.argsort(key, kind="stable")
        res = {}
        for tag, idx in (("arrival order", None), ("sorted", order)):
            oo, d2 = (o, dd) if idx is None else (o[idx], dd[idx])
            best = 1e9
            for _ in range(3):
                _, _, _, st = r.trace_closest(oo, d2, want_stats=True)
                best = min(best, st.kernel_ms[1])
            util = (st.inner_visits[0] / (64.0 * st.wave_steps[0]) if st.wave_steps[0] else 0, st.tri_tests[0] / (64.0 * st.wave_steps[1]) if st.wave_steps[1] else 0)
            res[tag] = best
            print(f"{name}: {label:32s} {tag:14s} {m} rays  {best:8.3f} ms  {m / best / 1e3:8.1f} Mrays/s  lanes/step {util[0]:.2f}/{util[1]:.2f}", flush=True)
        print(f"{name}: {label:32s} speed-up from sorting {res['arrival order'] / res['sorted']:.2f}x", flush=True)


if __name__ == "__main__":
    main()
