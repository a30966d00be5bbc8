#!/usr/bin/env python3
"""Host BVH build of the config-5 scene against the number of builder threads (TRT_HOST_THREADS).  usage: tools/build_cost.py [triangles] [threads ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinyraytracing_amd as T  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    counts = [int(x) for x in sys.argv[2:]] or [16, 8, 32, 64]
    d = os.path.join(T.SCENES_DIR, "back")
    print(f"logical CPUs {os.cpu_count()}, usable {len(os.sched_getaffinity(0))}", flush=True)
    for th in counts:
        os.environ["TRT_HOST_THREADS"] = str(th)
        s = T.Scene.load(os.path.join(d, "back.xml"), os.path.join(d, "back.obj"), os.path.join(d, "back.mtl"), d, 64, 64)
        s._check(s._lib.trth_scene_drop_tris(s._h, 6, 12))
        t = time.time()
        s._check(s._lib.trth_scene_add_blob(s._h, T.SEED_BLOB, n))
        tg = time.time() - t
        t = time.time()
        s.build_bvh(T.default_leaf("blob", s.info["n_triangles"]))
        print(f"{th:3d} builder threads: generated in {tg:.2f} s, BVH built and triangles reordered in {time.time() - t:.2f} s ({s.flat.contents.n_nodes} nodes)", flush=True)
        s.close()


if __name__ == "__main__":
    main()
