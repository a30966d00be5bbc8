#!/usr/bin/env python3
"""Randomised parity soak: random tiles / row interleaves / spp / seeds / flags / depth caps / memory budgets of the three cg22
scenes (and a 50 k-triangle soup) rendered by the HIP path through the C-ABI — default handles, handles on trees of the GPU builder, on the reference's own leaf size, on hostile
trees (boxes that do not nest, +-inf, inverted; non-finite vertices and normals), and handles on the optional code paths
(quantised nodes, speculative scheduler, per-lane traversal of the tiny scene) — and by the oracle; every image and every ray count
must be identical.  usage: tools/fuzz_parity.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import scene_util as SU  # noqa: E402
import tinyraytracing_amd as T  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    sizes = {"back": (257, 131), "veach-mis": (320, 180), "staircase": (192, 108), "soup": (160, 90)}
    scenes, renderers, alt, scenes_gpu_tree, on_gpu_tree, scenes_leaf8, on_leaf8, scenes_foreign, on_foreign, on_default_no_tail = {}, {}, {}, {}, {}, {}, {}, {}, {}, {}
    for name, (w, h) in sizes.items():
        scenes[name] = T.Scene.named(name, w, h, **({"n": 50000} if name == "soup" else {}))
        renderers[name] = T.Renderer(scenes[name], 0)
        # a third handle per scene on the tree of the GPU builder (include/trt_build.h): the oracle walks that tree for it
        scenes_gpu_tree[name] = T.Scene.named(name, w, h, builder="lbvh", **({"n": 50000} if name == "soup" else {}))
        on_gpu_tree[name] = T.Renderer(scenes_gpu_tree[name], 0)
        # a fourth handle per scene on a tree with the REFERENCE's leaf size (main.cpp:76 builds with 8): leaves of 4..8 triangles laid out as several slots of the
        # 8-wide nodes (round 4); the tiny scene per lane, so that it walks those nodes too
        # (the shipped scenes: the tree of the REFERENCE's builder as the oracle restates it, bvh.cpp:16-144; the soup: this repository's builder with leaf 8)
        scenes_leaf8[name] = T.Scene.named(name, w, h, leaf_num=8, n=50000) if name == "soup" else SU.load_with_reference_tree(name, w, h, 8)
        os.environ["TRT_TRACE_IMPL"] = "3"
        on_leaf8[name] = T.Renderer(scenes_leaf8[name], 0)
        del os.environ["TRT_TRACE_IMPL"]
        # a fifth handle per scene on a HOSTILE tree (what the C-ABI accepts and no builder emits): boxes pulled in so that they no longer contain what lies below
        # them, +-inf / +-1e38 box coordinates (some of them inverting a box), NaN / inf / 1e38 vertices and normals — the kernels then walk the exact 4-wide
        # nodes without distance culling (trt_wide.h boxesNested); the oracle walks the same arrays
        scenes_foreign[name] = T.Scene.named(name, w, h, leaf_num=int(rng.choice([2, 8])), **({"n": 50000} if name == "soup" else {}))
        if scenes_foreign[name].flat.contents.n_nodes > 40:
            SU.shrink_some_boxes(scenes_foreign[name], max(4, scenes_foreign[name].flat.contents.n_nodes // 40), seed=int(rng.integers(0, 1000)))
        SU.poison_geometry(scenes_foreign[name], seed=int(rng.integers(0, 1000)), every=200, boxes=True)
        on_foreign[name] = T.Renderer(scenes_foreign[name], 0)
        # a second handle per scene on the OTHER node kind of the traversal kernels (exact 4-wide nodes where the default is the 8-wide
        # compressed ones), and per-lane traversal instead of the uniform walk for the tiny scene
        os.environ["TRT_NODE_KIND"] = "0"
        os.environ["TRT_TRACE_IMPL"] = "3"
        os.environ["TRT_TAIL_N"] = "64"  # ... and no early hand-over to k_tail: tiles this small would leave every bounce after the first to it
        alt[name] = T.Renderer(scenes[name], 0)
        del os.environ["TRT_NODE_KIND"], os.environ["TRT_TRACE_IMPL"]
        on_default_no_tail[name] = T.Renderer(scenes[name], 0)  # the default kernels, all bounces in the queue kernels
        del os.environ["TRT_TAIL_N"]
    t0 = time.time()
    t_print = t0
    n = 0
    while time.time() - t0 < budget:
        if time.time() - t_print > 60:
            print(f"... {n} configurations so far, all identical ({time.time() - t0:.0f} s)", flush=True)
            t_print = time.time()
        name = list(sizes)[int(rng.integers(0, len(sizes)))]
        w, h = sizes[name]
        x0 = int(rng.integers(0, w - 1)); x1 = int(rng.integers(x0 + 1, min(w, x0 + 40) + 1))
        y0 = int(rng.integers(0, h - 1)); y1 = int(rng.integers(y0 + 1, min(h, y0 + 24) + 1))
        spp = int(rng.choice([1, 2, 3, 5, 8, 17, 33]))
        seed = int(rng.integers(0, 2 ** 32))
        flags = 0
        if rng.random() < 0.3: flags |= T.TRT_FLAG_FIXED_NEE
        if rng.random() < 0.2: flags |= T.TRT_FLAG_FIXED_PIXELS
        if rng.random() < 0.4: flags |= T.TRT_FLAG_OVERLAP
        if rng.random() < 0.3: flags |= T.TRT_FLAG_COUNT
        if rng.random() < 0.3: flags |= T.TRT_FLAG_RAY_OFFSET
        if rng.random() < 0.25: flags |= T.TRT_FLAG_SPECULAR_KS
        md = int(rng.choice([0, 0, 0, 1, 2, 5]))
        rows = None
        if rng.random() < 0.4:
            rb = int(rng.choice([1, 2, 8])); rm = int(rng.choice([2, 3, 8])); rows = (rb, rm, int(rng.integers(0, rm)))
        npix = (x1 - x0) * (y1 - y0)
        budget_b = 0 if rng.random() < 0.6 else int(npix * (2 * 48 + 32 + 48 * 8) * int(rng.choice([1, 2, 3])) + 4096 * 16 * 16)
        p = T.make_params(w, h, spp, seed, tile=(x0, y0, x1, y1), rows=rows, max_depth=md, flags=flags, mem_budget=budget_b)
        if not T.rows_selected(p):
            continue
        pick = rng.random()
        use = on_gpu_tree if pick < 0.2 else (alt if pick < 0.4 else (on_leaf8 if pick < 0.6 else (on_foreign if pick < 0.7 else (on_default_no_tail if pick < 0.85 else renderers))))
        try:
            img, st = use[name].render(p)
        except T.TrtError as e:
            if "mem_budget too small" in str(e):
                continue
            raise
        ref, ost = O.render((scenes_gpu_tree if use is on_gpu_tree else (scenes_leaf8 if use is on_leaf8 else (scenes_foreign if use is on_foreign else scenes)))[name].flat, p)
        ok = np.array_equal(img, ref) and (st.rays_camera, st.rays_shadow, st.rays_indirect) == (ost.rays_camera, ost.rays_shadow, ost.rays_indirect)
        n += 1
        if not ok:
            print("MISMATCH", name, "foreign" if use is on_foreign else ("leaf8" if use is on_leaf8 else ("lbvh" if use is on_gpu_tree else ("alt" if use is alt else ("no_tail" if use is on_default_no_tail else "default")))), dict(tile=(x0, y0, x1, y1), spp=spp, seed=seed, flags=flags, max_depth=md, rows=rows, mem_budget=budget_b), flush=True)
            sys.exit(1)
    print(f"fuzz parity: {n} random configurations, all bit-identical to the oracle ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
