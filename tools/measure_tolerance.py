#!/usr/bin/env python3
"""Measures the distance between the reference's own arithmetic (oracle/oracle_literal.cpp, ORACLE_MODE_LITERAL) and the
formulation the HIP kernels share with the fast oracle (Moller-Trumbore, fp32 scalars, trt_prims.h polynomials, iterative
beta form), same counter RNG, same seed: the protocol of SURVEY.md §8(c).  CPU only (both sides are oracles; the HIP image
is bit-identical to the fast one, tests/test_gpu_parity.py).  Prints one JSON line per workload; the frozen bounds live in
tests/test_literal_tolerance.py and DESIGN.md §2.

  python tools/measure_tolerance.py back 1024 1024 256      # BASELINE config 2
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import tinyraytracing_amd as T  # noqa: E402

SEEDS = {"back": T.SEED_BACK, "veach-mis": 0x5EED0002, "staircase": T.SEED_STAIRCASE}


def blocks(img, b=8):
    h, w, _ = img.shape
    return img[:h // b * b, :w // b * b].reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3))


def compare(g, c):
    """g: image under test, c: the literal (reference-arithmetic) image."""
    g = g.astype(np.float64)
    c = c.astype(np.float64)
    dn = np.sqrt(((g - c) ** 2).sum(axis=2))
    cn = np.sqrt((c ** 2).sum(axis=2))
    r = dn / (1.0 + cn)
    gb, cb = blocks(g), blocks(c)
    rb = np.sqrt(((gb - cb) ** 2).sum(axis=2)) / (1.0 + np.sqrt((cb ** 2).sum(axis=2)))
    return {
        "frac_pixels_within": {f"{tau:g}": float((r <= tau).mean()) for tau in (1e-5, 1e-4, 1e-3, 1e-2, 3e-2, 1e-1)},
        "pixel_tau_p99": float(np.percentile(r, 99)),
        "pixel_tau_max": float(r.max()),
        "rRMSE": float(np.sqrt(((g - c) ** 2).mean()) / np.sqrt((c ** 2).mean())),
        "mean_radiance_rel_diff": float(abs(g.mean() - c.mean()) / c.mean()),
        "block8_tau_p99": float(np.percentile(rb, 99)),
        "block8_tau_max": float(rb.max()),
        "frac_pixels_bit_identical": float((dn == 0).mean()),
    }


def main():
    name, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    threads = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    flags = int(sys.argv[6]) if len(sys.argv) > 6 else 0  # e.g. 32 = TRT_FLAG_RAY_OFFSET: the same measurement with Q6 out of the way
    sc = T.Scene.named(name, w, h)
    p = T.make_params(w, h, spp, SEEDS[name], flags=flags)
    t0 = time.time()
    fast, sf = O.render(sc.flat, p, threads=threads)
    t1 = time.time()
    lit, sl = O.render_literal(sc.flat, p, threads=threads)
    t2 = time.time()
    out = {"workload": f"{name} {w}x{h} {spp} spp, seed {SEEDS[name]:#x}, flags {flags}", "fast_s": round(t1 - t0, 1), "literal_s": round(t2 - t1, 1),
           "rays_fast": {"camera": sf.rays_camera, "shadow": sf.rays_shadow, "indirect": sf.rays_indirect},
           "rays_literal": {"camera": sl.rays_camera, "shadow": sl.rays_shadow, "indirect": sl.rays_indirect}}
    out.update(compare(fast, lit))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
