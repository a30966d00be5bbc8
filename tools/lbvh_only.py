#!/usr/bin/env python3
"""Builds the config-5 scene's BVH on the GPU and nothing else (for rocprofv3 --kernel-trace --stats: tools/lbvh_prof.sh).  usage: tools/lbvh_only.py [triangles]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinyraytracing_amd as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = os.path.join(T.SCENES_DIR, "back")
s = T.Scene.load(os.path.join(d, "back.xml"), os.path.join(d, "back.obj"), os.path.join(d, "back.mtl"), d, 64, 64)
s._check(s._lib.trth_scene_drop_tris(s._h, 6, 12))
s._check(s._lib.trth_scene_add_blob(s._h, T.SEED_BLOB, n))
for rep in range(3):  # the first call pays for loading the code objects
    s.build_bvh(2, "lbvh")
    print(f"build {rep}: {s.info['n_triangles']} triangles, first kernel to last {s.build_ms[0]:.1f} ms, call {s.build_ms[1]:.1f} ms", flush=True)
