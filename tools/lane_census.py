#!/usr/bin/env python3
"""Where are the idle SIMD lanes of the persistent traversal kernels?  trt_stats.lane_census of a counting render: of all lane slots
(64 x wave iterations) the share waiting for a node step, for a triangle step, holding a finished ray that waits for the refill
batch, and holding nothing — next to the share that actually worked (visits + tests).  usage: tools/lane_census.py [scene[:spp] ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinyraytracing_amd as T  # noqa: E402

KW = {"soup": {"n": 1_000_000}, "blob": {"n": 2_000_000}}


def main():
    for arg in sys.argv[1:] or ["veach-mis:64", "staircase:32", "soup:16", "blob:32"]:
        name, spp = (arg.split(":") + ["32"])[:2]
        s = T.Scene.named(name, 1920, 1080, **KW.get(name, {}))
        for rf in (os.environ.get("CENSUS_REFILLS") or "default").split(","):
            if rf != "default":
                os.environ["TRT_REFILL_MIN"] = rf
            else:
                os.environ.pop("TRT_REFILL_MIN", None)
            r = T.Renderer(s, 0)
            _, st = r.render(T.make_params(1920, 1080, int(spp), 77, flags=T.TRT_FLAG_COUNT | T.TRT_FLAG_TIMING))
            r.close()
            c = st.lane_census
            slots = 64.0 * max(c[3], 1)
            work = (st.inner_visits[0] + st.inner_visits[1] + st.tri_tests[0] + st.tri_tests[1]) / slots
            print(f"{name:10s} {spp:>3} spp refill {rf:7s}: lane slots {slots:.3e}  at node {c[0] / slots:.3f}  at triangle {c[1] / slots:.3f}  done, waiting {c[2] / slots:.3f}  empty {1 - (c[0] + c[1] + c[2]) / slots:.3f}"
                  f"  | working {work:.3f}  | steps: node {st.wave_steps[0] / max(c[3], 1):.3f} leaf {st.wave_steps[1] / max(c[3], 1):.3f}  lanes/step {(st.inner_visits[0] + st.inner_visits[1]) / max(64 * st.wave_steps[0], 1):.3f}/"
                  f"{(st.tri_tests[0] + st.tri_tests[1]) / max(64 * st.wave_steps[1], 1):.3f}  trace ms {st.kernel_ms[1] + st.kernel_ms[3]:.1f}", flush=True)
        s.close()


if __name__ == "__main__":
    main()
