#!/usr/bin/env python3
"""Start-up cost of a big scene, per phase: generating + building the BVH on the host (tinyraytracing_amd/host), trt_create (validation, the
4-wide and 8-wide collapses on the host, uploads), first render.  usage: tools/create_cost.py [triangles] (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinyraytracing_amd as T  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    import tinyraytracing_amd as TT
    d = os.path.join(TT.SCENES_DIR, "back")
    t = time.time()
    s = T.Scene.load(os.path.join(d, "back.xml"), os.path.join(d, "back.obj"), os.path.join(d, "back.mtl"), d, 3840, 2160)
    s._check(s._lib.trth_scene_drop_tris(s._h, 6, 12))
    s._check(s._lib.trth_scene_add_blob(s._h, TT.SEED_BLOB, n))
    t_gen = time.time() - t
    t = time.time()
    s.build_bvh(TT.default_leaf("blob", s.info["n_triangles"]))
    t_build = time.time() - t
    t = time.time()
    f = s.flat.contents
    t_flat = time.time() - t
    print(f"blob {f.n_tris} triangles, {f.n_nodes} nodes: generated in {t_gen:.2f} s, BVH built (and the triangles reordered) in {t_build:.2f} s, flattened in {t_flat:.2f} s", flush=True)
    os.environ["TRT_DEBUG"] = "1"
    for nk in ("1", "0"):
        os.environ["TRT_NODE_KIND"] = nk
        t = time.time()
        r = T.Renderer(s, 0)
        t_create = time.time() - t
        t = time.time()
        r.render(T.make_params(3840, 2160, 1, 1))
        t_first = time.time() - t
        print(f"node kind {nk}: trt_create {t_create:.2f} s, first 1-spp 4K render {t_first:.2f} s", flush=True)
        r.close()
    os.environ["TRT_NODE_KIND"] = "1"
    for members in (1, 2):
        t = time.time()
        g = T.GroupRenderer(s, [0] * members)
        print(f"trt_group_create, {members} member(s) on device 0: {time.time() - t:.2f} s (the host half runs once, the uploads side by side)", flush=True)
        g.close()


if __name__ == "__main__":
    main()
