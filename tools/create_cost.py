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
    t = time.time()
    s = T.Scene.named("blob", 3840, 2160, n=n)
    t_scene = time.time() - t
    f = s.flat.contents
    print(f"blob {f.n_tris} triangles, {f.n_nodes} nodes: generated and built on the host in {t_scene:.2f} s", flush=True)
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


if __name__ == "__main__":
    main()
