#!/usr/bin/env python3
"""The noise floor of the block comparison of tests/test_ref_png.py: two renders of the SAME estimator with different seeds, at the
snapshot's resolution and sample count, compared exactly as a render is compared with the reference's snapshot (16x16-block means in
linear space after the 8-bit encode) — and each of them against the snapshot, plus the ratio of the mean encoded radiance.  What
exceeds the floor is systematic.  CPU only (the fast oracle).  usage: tools/two_seed_floor.py [scene ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import test_ref_png as R  # noqa: E402
import tinyraytracing_amd as T  # noqa: E402

CASES = {"back": ("back_image10.png", 10), "back0": ("back_image10-0.png", 10), "veach-mis": ("veach-mis_image10.png", 10), "staircase": ("staircase_image10.png", 10)}


def main():
    for key in sys.argv[1:] or list(CASES):
        fixture, spp = CASES[key]
        scene = "back" if key.startswith("back") else key
        png = R._png(fixture)
        h, w = png.shape[:2]
        s = T.Scene.named(scene, w, h)
        a = O.render(s.flat, T.make_params(w, h, spp, 1001))[0]
        b = O.render(s.flat, T.make_params(w, h, spp, 2002))[0]
        enc = lambda x: R._lin8(T.tonemap(x)).mean()
        floor = R._compare(a, T.tonemap(b))
        va, vb = R._compare(a, png), R._compare(b, png)
        print(f"{fixture}: two seeds at {spp} spp: median block error {floor[0]:.4f} p90 {floor[1]:.4f} corr {floor[2]:.4f} | vs snapshot: {va[0]:.4f} / {vb[0]:.4f} "
              f"| mean encoded radiance render / snapshot: {enc(a) / R._lin8(png).mean():.4f} / {enc(b) / R._lin8(png).mean():.4f}", flush=True)


if __name__ == "__main__":
    main()
