#!/usr/bin/env python3
"""Which light's contribution differs between this build and the reference's staircase snapshot?  Renders the scene once per light
(all other radiances set to zero: the random draws stay the same, so the six images add up to the full render) and fits the snapshot's
16x16... block means as a non-negative-free least-squares combination of the six block images, on the blocks that stay clear of the
8-bit clamp.  A weight of 1 says "this light's light arrives as in the reference".  CPU only (the fast oracle).
Round 3: weights 1.71, 1.71 (two lights of negligible energy), 1.12, 0.88 (leftLight: 72 % of the energy), 0.97, 0.92.
Round 4: superseded by tools/staircase_decomp.py + tools/staircase_fit.py (full resolution, 256 spp, bootstrap intervals) — no per-light factor explains
the snapshot; the weight of SPECULAR bounces does (profiles/r04_staircase_residual.txt).
usage: tools/per_light_fit.py [spp]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import test_ref_png as R  # noqa: E402
import tinyraytracing_amd as T  # noqa: E402


def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    png = R._png("staircase_image256.png")
    H, W = png.shape[:2]
    w, h = W // 2, H // 2
    s = T.Scene.named("staircase", w, h)
    f = s.flat.contents
    nl = f.n_lights
    orig = [tuple(f.lights[k].radiance) for k in range(nl)]
    imgs = []
    for keep in range(nl):
        for k in range(nl):
            for c in range(3):
                v = orig[k][c] if k == keep else 0.0
                f.lights[k].radiance[c] = v
                f.materials[f.lights[k].mat].radiance[c] = v
        imgs.append(O.render(s.flat, T.make_params(w, h, spp, 1001))[0].astype(np.float64))
        print(f"light {keep} ({s.material_name(f.lights[keep].mat)}): mean radiance {imgs[-1].mean():.5f}", flush=True)
    ref = R._lin8(png).reshape(h, 2, w, 2, 3).mean((1, 3))
    B = 40
    blocks = lambda x: x[:h // B * B, :w // B * B].reshape(h // B, B, w // B, B, 3).mean((1, 3))
    A = np.stack([blocks(i).ravel() for i in imgs], 1)
    y = blocks(ref).ravel()
    m = blocks(sum(imgs)).ravel() < 0.6
    c = np.linalg.lstsq(A[m], y[m], rcond=None)[0]
    print("weights of the single-light renders that reproduce the snapshot:", np.round(c, 3))
    print("energy share of each light:", np.round(A[m].sum(0) / A[m].sum(), 3), " this build / snapshot on those blocks:", round(A[m].sum(1).mean() / y[m].mean(), 4))


if __name__ == "__main__":
    main()
