#!/usr/bin/env python3
"""Round 4 (VERDICT r03 task 1): fits of the reference's converged staircase render (tests/golden/ref_png/staircase_image256.png =
example-scenes-cg22/staircase/image256.png) by the exact decompositions of this build's render (tools/staircase_decomp.py).  CPU only.

  usage: tools/staircase_fit.py DECOMP.npz [block multiple]

Everything is done on block means of LINEAR radiance (the PNG decoded with the inverse of imshow()'s transfer, main.cpp:34-36), on the
blocks that stay clear of the 8-bit clamp in both images.  Confidence intervals: bootstrap over blocks (the blocks are resampled
with replacement, 300 times; 2.5 % .. 97.5 %)."""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "tests", "golden", "ref_png", "staircase_image256.png")


def lin8(a):
    return ((a.astype(np.float64) + 0.5) / 255.0) ** 2.2


def blocks(img, b):
    h, w = img.shape[:2]
    return img[:h // b * b, :w // b * b].reshape(h // b, b, w // b, b, -1).mean(axis=(1, 3))


def coarsen(x, m):
    return blocks(x, m) if m > 1 else x


def boot(A, y, n=300, seed=1, nonneg=False):
    rng = np.random.default_rng(seed)
    c0 = np.linalg.lstsq(A, y, rcond=None)[0]
    cs = []
    for _ in range(n):
        i = rng.integers(0, len(y), len(y))
        cs.append(np.linalg.lstsq(A[i], y[i], rcond=None)[0])
    cs = np.array(cs)
    return c0, np.percentile(cs, 2.5, axis=0), np.percentile(cs, 97.5, axis=0)


def main():
    d = np.load(sys.argv[1], allow_pickle=False)
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    B = int(d["block"])
    names = [str(x) for x in d["light_names"]]
    png = np.asarray(Image.open(REF).convert("RGB"))
    ref = blocks(lin8(png), B)
    ref_max8 = png[:png.shape[0] // B * B, :png.shape[1] // B * B].reshape(png.shape[0] // B, B, png.shape[1] // B, B, 3).max(axis=(1, 3, 4))
    ours_u8 = d["full_u8"]
    ours_enc = blocks(lin8(ours_u8), B)          # this build through the same 8-bit encode
    ours2_enc = blocks(lin8(d["full2_u8"]), B)   # a second seed
    ours_lin = d["depth_all_0"].astype(np.float64)
    clear = (ref_max8 < 250) & (d["full_blockmax"] < 0.95)
    print(f"blocks of {B}x{B}: {clear.size}, clear of the clamp in both images: {int(clear.sum())}")

    def stat(a, b, what):
        m = clear
        rel = np.abs(a[m] - b[m]) / (0.02 + b[m])
        print(f"  {what:52s} mean ratio {a[m].mean() / b[m].mean():.4f}  per channel {np.round(a[m].mean(0) / b[m].mean(0), 4)}  median block error {np.median(rel):.4f}")

    print("like with like (both through the 8-bit encode):")
    stat(ours_enc, ref, "this build / reference snapshot")
    stat(ours2_enc, ours_enc, "this build, second seed / first seed (noise floor)")
    stat(ours_lin, ours_enc, "this build linear / through the encode (encode bias)")

    # coarser blocks for the fits
    def prep(x):
        return coarsen(np.where(clear[..., None], x, np.nan), M)

    def valid_of(*xs):
        v = np.ones(xs[0].shape[:2], bool)
        for x in xs:
            v &= np.isfinite(x).all(-1)
        return v

    y_img = prep(ref)
    # ---------------- by light
    L = [prep(d[f"depth_L{k}_0"].astype(np.float64)) for k in range(len(names))]
    v = valid_of(y_img, *L)
    A = np.stack([x[v].ravel() for x in L], 1)
    y = y_img[v].ravel()
    c, lo, hi = boot(A, y)
    share = A.sum(0) / A.sum()
    print(f"\nby light ({int(v.sum())} blocks of {B * M}x{B * M}, 3 channels): weight that reproduces the snapshot [95 % interval], share of this build's energy")
    for k, n in enumerate(names):
        print(f"  {n:14s} {c[k]:7.3f} [{lo[k]:6.3f}, {hi[k]:6.3f}]   share {share[k]:.3f}")
    res = y - A @ c
    print(f"  residual rms / mean: {np.sqrt((res ** 2).mean()) / y.mean():.4f}; with all weights 1: {np.sqrt(((y - A.sum(1)) ** 2).mean()) / y.mean():.4f}")

    # ---------------- by depth
    K = 12
    R = [prep(d[f"depth_all_{k}"].astype(np.float64)) for k in range(1, K + 1)]
    full = prep(ours_lin)
    D = [R[0]] + [R[k] - R[k - 1] for k in range(1, K)] + [full - R[K - 1]]
    v = valid_of(y_img, full, *R)
    Dm = np.stack([x[v].ravel() for x in D], 1)
    y = y_img[v].ravel()
    print(f"\nby path vertex (light gathered at depth d; d = 0 is the camera ray's hit): share of energy, and free weights (groups 0, 1, 2, 3-4, 5-7, 8+)")
    print("  share:", np.round(Dm.sum(0) / Dm.sum(), 4))
    groups = [[0], [1], [2], [3, 4], [5, 6, 7], list(range(8, K + 1))]
    G = np.stack([Dm[:, g].sum(1) for g in groups], 1)
    c, lo, hi = boot(G, y)
    for g, cc, l, h in zip(groups, c, lo, hi):
        print(f"  depth {str(g):22s} weight {cc:7.3f} [{l:6.3f}, {h:6.3f}]")
    # one-parameter models: weight rho^d ; and a * rho^d
    best = None
    for rho in np.arange(0.80, 1.101, 0.005):
        w = rho ** np.arange(K + 1)
        w[-1] = rho ** (K + 1)
        pred = Dm @ w
        a = (pred @ y) / (pred @ pred)
        e1 = np.sqrt(((y - pred) ** 2).mean()) / y.mean()
        e2 = np.sqrt(((y - a * pred) ** 2).mean()) / y.mean()
        if best is None or e1 < best[1]:
            best = (rho, e1)
        if abs(rho - 1.0) < 1e-9 or abs(rho - 0.875) < 1e-9 or abs(rho - 0.9) < 1e-9 or abs(rho - 0.95) < 1e-9:
            print(f"  model weight = rho^d, rho {rho:.3f}: rms/mean {e1:.4f}   (with a free overall scale {a:.4f}: {e2:.4f})")
    print(f"  best rho (no free scale): {best[0]:.3f}  rms/mean {best[1]:.4f}")
    # bootstrap of rho
    rng = np.random.default_rng(3)
    rhos = []
    grid = np.arange(0.80, 1.101, 0.005)
    W = np.stack([np.concatenate([r ** np.arange(K), [r ** (K + 1)]]) for r in grid], 1)
    P = Dm @ W
    for _ in range(200):
        i = rng.integers(0, len(y), len(y))
        e = ((y[i][:, None] - P[i]) ** 2).mean(0)
        rhos.append(grid[np.argmin(e)])
    print(f"  rho 95 % interval: [{np.percentile(rhos, 2.5):.3f}, {np.percentile(rhos, 97.5):.3f}]")

    # ---------------- by light AND depth: per-light rho
    print("\nper light: best rho of weight = rho^d on that light's images (the other lights at weight 1)")
    for k, n in enumerate(names):
        Rk = [prep(d[f"depth_L{k}_{j}"].astype(np.float64)) for j in range(1, K + 1)]
        fk = prep(d[f"depth_L{k}_0"].astype(np.float64))
        Dk = [Rk[0]] + [Rk[j] - Rk[j - 1] for j in range(1, K)] + [fk - Rk[K - 1]]
        vv = valid_of(y_img, full, fk, *Rk)
        Dkm = np.stack([x[vv].ravel() for x in Dk], 1)
        others = (full[vv].ravel() - fk[vv].ravel())
        yy = y_img[vv].ravel()
        errs = [np.sqrt(((yy - others - Dkm @ np.concatenate([r ** np.arange(K), [r ** (K + 1)]])) ** 2).mean()) for r in grid]
        print(f"  {n:14s} rho {grid[int(np.argmin(errs))]:.3f}   mean depth of its energy {(Dkm.sum(0) * np.arange(K + 1)).sum() / Dkm.sum():.2f}")

    # ---------------- by number of TRANSMISSION events
    taus = [float(t) for t in d["taus"]]
    for tag in ("all",) + tuple(f"L{k}" for k in range(len(names)) if f"tau_L{k}_{taus[0]}" in d):
        T = [prep(d[f"tau_{tag}_{t}"].astype(np.float64)) for t in taus]
        vv = valid_of(y_img, *T)
        Tm = np.stack([x[vv] for x in T], 0)            # [tau, block, channel]
        Kt = len(taus) - 1
        V = np.vander(np.array(taus), Kt + 1, increasing=True)
        coef = np.linalg.solve(V, Tm.reshape(len(taus), -1)).reshape(Kt + 1, -1, 3)   # I_k per block and channel
        tr = np.array([0.8, 1.0, 0.95])
        comp = [coef[k] * tr[None, :] ** k for k in range(Kt + 1)]   # what each k contributes at the scene's Tr
        tot = sum(c_.sum() for c_ in comp)
        print(f"\nby TRANSMISSION events ({tag}): share of energy that crossed k glass interfaces, k = 0..{Kt}: {np.round([c_.sum() / tot for c_ in comp], 4)}")
        if tag == "all":
            # free weights on k = 0, 1, 2, 3+
            G = np.stack([comp[0].ravel(), comp[1].ravel(), comp[2].ravel(), sum(comp[3:]).ravel()], 1)
            yy = y_img[vv].ravel()
            c, lo, hi = boot(G, yy)
            for k, (cc, l, h) in enumerate(zip(c, lo, hi)):
                print(f"  k = {k}{'+' if k == 3 else ' '}: weight {cc:7.3f} [{l:6.3f}, {h:6.3f}]")
            for g in (1.0, 0.94, 0.9):
                pred = sum(comp[k] * g ** k for k in range(Kt + 1)).ravel()
                print(f"  model weight = g^k, g {g:.2f}: rms/mean {np.sqrt(((yy - pred) ** 2).mean()) / yy.mean():.4f}")


if __name__ == "__main__":
    main()
