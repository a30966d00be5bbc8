#!/usr/bin/env python3
"""Tree quality of the GPU builder against the size of its clusters (the maximal radix subtrees under the host's SAH top; TRT_LBVH_CLUSTER): node visits and
triangle tests per ray (COUNT kernels) relative to the host SAH builder's tree of the same scene, build time of the call.  GPU box.
usage: tools/lbvh_cluster_sweep.py [scene[:triangles] ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinyraytracing_amd as T  # noqa: E402


def measure(name, n, builder, cluster):
    if cluster is not None:
        os.environ["TRT_LBVH_CLUSTER"] = str(cluster)
    else:
        os.environ.pop("TRT_LBVH_CLUSTER", None)
    w, h, spp = 480, 270, 8
    s = T.Scene.named(name, w, h, leaf_num=T.default_leaf(name, 10**6), builder=builder, n=n)
    os.environ.pop("TRT_LBVH_CLUSTER", None)
    r = T.Renderer(s, 0)
    _, st = r.render(T.make_params(w, h, spp, 11, flags=T.TRT_FLAG_COUNT))
    rays = st.rays_camera + st.rays_shadow + st.rays_indirect
    r.render(T.make_params(w, h, spp, 11))
    t = time.perf_counter()
    r.render(T.make_params(w, h, spp, 11))
    ms = (time.perf_counter() - t) * 1e3
    out = ((st.inner_visits[0] + st.inner_visits[1]) / rays, (st.tri_tests[0] + st.tri_tests[1]) / rays, rays / ms / 1e3, getattr(s, "build_ms", None), s.info["n_triangles"])
    r.close()
    s.close()
    return out


def main():
    specs = sys.argv[1:] or ["staircase", "veach-mis", "blob:150000", "blob:2000000", "soup:1000000"]
    for spec in specs:
        name, _, n = spec.partition(":")
        n = int(n) if n else None
        v0, t0, m0, _, nt = measure(name, n, "auto", None)
        print(f"{name} ({nt} triangles): host SAH: {v0:.2f} visits {t0:.2f} tests per ray, {m0:.0f} Mrays/s", flush=True)
        for cl in (None, 0, 16, 32, 64, 128, 256, 512, 2048):
            v, t, m, b, _ = measure(name, n, "lbvh", cl)
            print(f"   cluster {'default' if cl is None else cl:>7}: visits {v:6.2f} ({(v / v0 - 1) * 100:+5.1f} %)  tests {t:6.2f} ({(t / t0 - 1) * 100:+5.1f} %)  {m:6.0f} Mrays/s ({(m / m0 - 1) * 100:+5.1f} %)  "
                  f"build: device {b[0]:.1f} ms, call {b[1]:.1f} ms", flush=True)


if __name__ == "__main__":
    main()
