#!/usr/bin/env python3
"""GPU LBVH builder (include/trt_build.h) against the host SAH builder on the big scenes: build time, node visits and triangle tests per ray
(COUNT kernels), render time of the same workload on either tree.  usage: tools/lbvh_cost.py [scene:triangles:width:height:spp ...]  (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinyraytracing_amd as T  # noqa: E402


def scene(name, n, w, h, builder):
    d = os.path.join(T.SCENES_DIR, "back")
    s = T.Scene.load(os.path.join(d, "back.xml"), os.path.join(d, "back.obj"), os.path.join(d, "back.mtl"), d, w, h)
    s._check(s._lib.trth_scene_drop_tris(s._h, 6, 12))
    s._check((s._lib.trth_scene_add_soup if name == "soup" else s._lib.trth_scene_add_blob)(s._h, T.SEED_SOUP if name == "soup" else T.SEED_BLOB, n))
    leaf = T.default_leaf(name, s.info["n_triangles"])
    t = time.time()
    s.build_bvh(leaf, builder)
    return s, time.time() - t


def main():
    specs = sys.argv[1:] or ["soup:1000000:1920:1080:16", "blob:2000000:1920:1080:16", "blob:10000000:3840:2160:16"]
    for spec in specs:
        name, n, w, h, spp = spec.split(":")
        n, w, h, spp = int(n), int(w), int(h), int(spp)
        for builder in ("auto", "lbvh", "radix"):
            if builder == "radix":
                os.environ["TRT_LBVH_CLUSTER"] = "0"  # the radix tree as it is, no SAH top
            s, t_build = scene(name, n, w, h, "lbvh" if builder == "radix" else builder)
            os.environ.pop("TRT_LBVH_CLUSTER", None)
            f = s.flat.contents
            t = time.time()
            r = T.Renderer(s, 0)
            t_create = time.time() - t
            _, st = r.render(T.make_params(w, h, spp, 11, flags=T.TRT_FLAG_COUNT))
            rays = st.rays_camera + st.rays_shadow + st.rays_indirect
            r.render(T.make_params(w, h, spp, 11))
            t = time.time()
            r.render(T.make_params(w, h, spp, 11))
            ms = (time.time() - t) * 1e3
            extra = f" (device {s.build_ms[0]:.1f} ms, call {s.build_ms[1]:.0f} ms, rest = vertices out of the host scene, its flattening through the returned order)" if builder != "auto" else ""
            print(f"{name} {f.n_tris} triangles, {builder:5s}: build {t_build:.2f} s{extra}, {f.n_nodes} nodes depth {f.bvh_depth}, trt_create {t_create:.2f} s, "
                  f"visits/ray {(st.inner_visits[0] + st.inner_visits[1]) / rays:.2f} tests/ray {(st.tri_tests[0] + st.tri_tests[1]) / rays:.2f}, "
                  f"{w}x{h} {spp} spp: {ms:.1f} ms = {rays / ms / 1e3:.0f} Mrays/s", flush=True)
            r.close()
            s.close()


if __name__ == "__main__":
    main()
