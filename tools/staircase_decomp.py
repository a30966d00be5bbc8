#!/usr/bin/env python3
"""Round 4 (VERDICT r03 task 1): decompositions of the staircase render for the fit against the reference's own converged snapshot
(example-scenes-cg22/staircase/image256.png, kept as tests/golden/ref_png/staircase_image256.png).  Runs on the GPU box through the
product path (the HIP renderer is bit-identical to the oracle, so it serves as a fast oracle here); the fits themselves run on the CPU
(tools/staircase_fit.py) on the small block images this writes.

Every decomposition is EXACT (same seed, same random streams, the images of one family add up to the full render):
  * by light: all radiances but one set to zero (the draws do not depend on radiance);
  * by path vertex: max_depth = k keeps the vertices 0 .. k-1 of every path (trt_params.max_depth), so the difference of two
    renders is the light gathered at one depth — a per-bounce factor rho in the reference would show as weights rho^d;
  * by number of TRANSMISSION events: the Glass material's Tr only weights a path (pathTracing.cpp:95-96), it never steers it, so
    the render is a polynomial in a scalar Tr = tau whose k-th coefficient is the light that crossed k interfaces; renders at
    several tau give the coefficients by a Vandermonde solve.
usage: tools/staircase_decomp.py OUT.npz [spp] [block]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinyraytracing_amd as T  # noqa: E402

W, H = 1280, 720  # the snapshot's size (staircase.xml)


def blocks(img, b):
    h, w, _ = img.shape
    return img[:h // b * b, :w // b * b].reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3)).astype(np.float32)


def main():
    out_path = sys.argv[1]
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    seed = T.SEED_STAIRCASE
    s = T.Scene.named("staircase", W, H)
    f = s.flat.contents
    nl = f.n_lights
    names = [s.material_name(f.lights[k].mat) for k in range(nl)]
    orig = [tuple(f.lights[k].radiance) for k in range(nl)]
    glass = [i for i in range(f.n_materials) if s.material_name(i) == "Glass"][0]
    tr0 = tuple(f.materials[glass].Tr)

    def set_lights(keep):
        for k in range(nl):
            for c in range(3):
                v = orig[k][c] if (keep is None or k == keep) else 0.0
                f.lights[k].radiance[c] = v
                f.materials[f.lights[k].mat].radiance[c] = v

    def set_tr(t):
        for c in range(3):
            f.materials[glass].Tr[c] = t[c]

    res = {"light_names": np.array(names), "spp": spp, "block": B, "seed": seed}
    depths = list(range(1, 13)) + [0]
    t0 = time.time()
    for keep in [None] + list(range(nl)):
        set_lights(keep)
        r = T.Renderer(s, 0)
        tag = "all" if keep is None else f"L{keep}"
        for k in depths:
            img, st = r.render(T.make_params(W, H, spp, seed, max_depth=k))
            res[f"depth_{tag}_{k}"] = blocks(img.astype(np.float64), B)
            if keep is None and k == 0:
                res["full_f16"] = img.astype(np.float16)
                res["full_u8"] = T.tonemap(img)
                res["full_blockmax"] = img[:H // B * B, :W // B * B].reshape(H // B, B, W // B, B, 3).max(axis=(1, 3, 4)).astype(np.float32)
        r.close()
        print(f"{tag}: {len(depths)} renders, {time.time() - t0:.1f} s", flush=True)
    # a second seed of the full render: the noise floor of every statistic below
    set_lights(None)
    r = T.Renderer(s, 0)
    img2, _ = r.render(T.make_params(W, H, spp, seed + 77))
    res["full2_blocks"] = blocks(img2.astype(np.float64), B)
    res["full2_u8"] = T.tonemap(img2)
    r.close()
    # by number of TRANSMISSION events: scalar Tr = tau on the Glass material
    taus = [0.0, 0.25, 0.5, 0.75, 1.0, 1.25, 1.5]
    res["taus"] = np.array(taus)
    for keep in [None, names.index("leftLight")]:
        set_lights(keep)
        tag = "all" if keep is None else f"L{keep}"
        for t in taus:
            set_tr((t, t, t))
            r = T.Renderer(s, 0)
            img, _ = r.render(T.make_params(W, H, spp, seed))
            res[f"tau_{tag}_{t}"] = blocks(img.astype(np.float64), B)
            r.close()
        print(f"tau {tag}: {len(taus)} renders, {time.time() - t0:.1f} s", flush=True)
    set_tr(tr0)
    set_lights(None)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    np.savez_compressed(out_path, **res)
    print("wrote", out_path, os.path.getsize(out_path) // 1024, "KiB", flush=True)


if __name__ == "__main__":
    main()
