#!/usr/bin/env python3
"""Where does k_trace_closest spend its time on the oct nodes?  Ray batches through trt_trace_closest and depth-limited renders,
node kind 0 against 1, with the redo counter.  usage: tools/oct_debug.py [scene ...]"""
import os
import sys
import time

import numpy as np  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raygen  # noqa: E402
import tinyraytracing_amd as T  # noqa: E402

KW = {"soup": {"n": 1_000_000}, "blob": {"n": 2_000_000}}


def main():
    for name in sys.argv[1:] or ["staircase", "blob"]:
        s = T.Scene.named(name, 1920, 1080, **KW.get(name, {}))
        lo, hi = raygen.scene_bounds(s)
        o1, d1 = raygen.primary_rays(s, 1920, 1080)
        o2, d2 = raygen.random_rays(2_000_000, lo, hi, seed=3)
        for nk in ("0", "1"):
            os.environ["TRT_NODE_KIND"] = nk
            r = T.Renderer(s, 0)
            for label, (o, d) in (("primary", (o1, d1)), ("random", (o2, d2))):
                r.trace_closest(o, d)
                _, tri, _, st = r.trace_closest(o, d, want_stats=True)
                print(f"{name} nk{nk} trace_closest {label:8s} {len(o)} rays: {st.kernel_ms[1]:9.3f} ms  visits/ray {st.inner_visits[0] / len(o):6.2f} tests/ray {st.tri_tests[0] / len(o):5.2f} "
                      f"lanes {st.inner_visits[0] / max(64 * st.wave_steps[0], 1):.2f}/{st.tri_tests[0] / max(64 * st.wave_steps[1], 1):.2f} redo {st.redo_rays} hits {(tri >= 0).mean():.3f}", flush=True)
            for md in (1, 2, 0):
                p = T.make_params(1920, 1080, 4, 77, max_depth=md, flags=T.TRT_FLAG_TIMING)
                r.render(p)
                t0 = time.time()
                _, st = r.render(p)
                print(f"{name} nk{nk} render max_depth {md}: {(time.time() - t0) * 1e3:8.2f} ms  closest {st.kernel_ms[1]:8.3f} ({st.launches[1]} launches) shade {st.kernel_ms[2]:7.3f} shadow {st.kernel_ms[3]:8.3f} "
                      f"tail {st.kernel_ms[5]:7.3f} redo {st.redo_rays} rays {st.rays}", flush=True)
            r.close()
        s.close()


if __name__ == "__main__":
    main()
