#!/usr/bin/env python3
"""Parity on RANDOM scenes: every other parity test renders the three shipped scenes and two synthetic meshes; this one writes scenes nobody modelled —
1 to 400 random triangles (slivers, points, duplicates, coplanar stacks, axis-aligned quads, far outliers), random per-vertex normals and texture coordinates,
1 to 12 materials drawn from every branch of nextRay() (diffuse, Phong with Ns below / at / above 1, glass with Ni below / at / above 1, black, Kd = Ks = 0,
textured, emissive and textured at once), 0 to 8 lights in any XML order (lights whose material no triangle uses, lights far larger or smaller than lights[0]:
quirk Q3), a random camera (inside the geometry, looking away from it, tiny / huge fovy) — through the loaders, both host builders, leaf sizes 1..15, and renders
them with the oracle and with the device code: the CPU build (tests/hostsim, both node kinds) by default, the kernels through the C-ABI with --gpu.  Every image
bit and every ray count must agree.

usage: tools/fuzz_scenes.py [--seconds 120] [--seed 1] [--gpu] [--keep DIR]      (exit code 0 = all identical)"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import tinyraytracing_amd as T  # noqa: E402


def fmt(x):
    return repr(float(np.float32(x)))


def write_random_scene(d, rng, big=False):
    """returns (description, width, height); big: now and then a few thousand to tens of thousands of triangles (deep trees, spilled stacks, many waves)"""
    n_mat = int(rng.integers(1, 13))
    mats = []
    tex_id = 0
    for m in range(n_mat):
        kind = rng.choice(["diffuse", "phong", "phong_low", "glass", "glass_low", "black", "zero", "textured", "mirrorish"])
        Kd, Ks, Tr, Ns, Ni, tex = rng.random(3), np.zeros(3), np.zeros(3), 1.0, 1.0, None
        if kind == "phong":
            Ks, Ns = rng.random(3), float(rng.choice([1.0000001, 2, 10, 100, 5000]))
        elif kind == "phong_low":
            Ks, Ns = rng.random(3), float(rng.choice([0.0, 0.5, 1.0, -1.0]))
        elif kind == "glass":
            Tr, Ni = rng.random(3), float(rng.choice([1.0000001, 1.3, 1.5, 2.4, 10.0]))
        elif kind == "glass_low":
            Tr, Ni, Ks, Ns = rng.random(3), float(rng.choice([0.5, 0.9, 0.99999994, 1.0, 0.0])), rng.random(3), 20.0
        elif kind == "black":
            Kd = np.zeros(3); Ks = rng.random(3); Ns = 30.0
        elif kind == "zero":
            Kd = np.zeros(3)
        elif kind == "textured":
            tex = f"tex{tex_id}.ppm"
            tex_id += 1
            w, h = int(rng.integers(1, 9)), int(rng.integers(1, 9))
            with open(os.path.join(d, tex), "wb") as f:
                f.write(b"P6\n%d %d\n255\n" % (w, h) + rng.integers(0, 256, w * h * 3, dtype=np.uint8).tobytes())
            if rng.random() < 0.4:
                Ks, Ns = rng.random(3), 40.0
        elif kind == "mirrorish":
            Kd = rng.random(3) * 0.01; Ks = np.ones(3); Ns = 1e6
        mats.append(dict(name=f"m{m}", kind=kind, Kd=Kd, Ks=Ks, Tr=Tr, Ns=Ns, Ni=Ni, tex=tex))
    with open(os.path.join(d, "s.mtl"), "w") as f:
        for m in mats:
            f.write(f"newmtl {m['name']}\nKd {' '.join(fmt(x) for x in m['Kd'])}\nKs {' '.join(fmt(x) for x in m['Ks'])}\nTr {' '.join(fmt(x) for x in m['Tr'])}\n"
                    f"Ns {fmt(m['Ns'])}\nNi {fmt(m['Ni'])}\n" + (f"map_Kd {m['tex']}\n" if m["tex"] else ""))
    # geometry
    n_tri = int(rng.choice([1, 2, 3, 7, 20, 60, 150, 400] + ([3000, 12000, 40000] if big else [])))
    scale = float(rng.choice([0.01, 1.0, 1.0, 1.0, 50.0, 3000.0]))
    V, VN, VT, F = [], [], [], []

    def vert(p):
        V.append(p); return len(V)
    for _ in range(int(rng.integers(1, 6))):
        VN.append(rng.normal(size=3))
    VN.append(np.array([0.0, 0.0, 1.0])); VN.append(np.array([0.0, 1.0, 0.0]))
    for _ in range(int(rng.integers(1, 6))):
        VT.append(rng.random(2) * float(rng.choice([1.0, 1.0, 3.0, -2.0])))
    t = 0
    while t < n_tri:
        kind = rng.choice(["random", "random", "small", "sliver", "point", "dup", "quad", "stack", "far"])
        mat = int(rng.integers(0, n_mat))
        c = rng.uniform(-1, 1, 3) * scale
        tris = []
        if kind == "random":
            tris = [c + rng.uniform(-1, 1, (3, 3)) * scale * 0.7]
        elif kind == "small":
            tris = [c + rng.uniform(-1, 1, (3, 3)) * scale * 0.05]
        elif kind == "sliver":
            a = c; b = c + rng.uniform(-1, 1, 3) * scale
            tris = [np.array([a, b, (a + b) / 2 + rng.uniform(-1, 1, 3) * scale * 1e-4])]
        elif kind == "point":
            tris = [np.array([c, c, c])]
        elif kind == "dup" and F:
            tris = [np.array([V[i - 1] for i in F[int(rng.integers(0, len(F)))][0]])]
        elif kind == "quad":
            ax = int(rng.integers(0, 3)); u, v = [(1, 2), (0, 2), (0, 1)][ax]
            e1 = np.zeros(3); e2 = np.zeros(3); e1[u] = rng.uniform(0.2, 1.5) * scale; e2[v] = rng.uniform(0.2, 1.5) * scale
            tris = [np.array([c, c + e1, c + e1 + e2]), np.array([c, c + e1 + e2, c + e2])]
        elif kind == "stack":
            base = c + rng.uniform(-1, 1, (3, 3)) * scale * 0.5
            tris = [base.copy() for _ in range(int(rng.integers(2, 5)))]
        elif kind == "far":
            tris = [c * 40 + rng.uniform(-1, 1, (3, 3)) * scale]
        for tri in tris:
            idx = tuple(vert(p) for p in tri)
            F.append((idx, mat, tuple(int(rng.integers(1, len(VT) + 1)) for _ in range(3)), tuple(int(rng.integers(1, len(VN) + 1)) for _ in range(3))))
            t += 1
    with open(os.path.join(d, "s.obj"), "w") as f:
        order_vt_first = rng.random() < 0.5  # quirk Q12: which of vt / vn comes first decides how the face slots are read
        blocks = [("vt", VT), ("vn", VN)] if order_vt_first else [("vn", VN), ("vt", VT)]
        for p in V:
            f.write("v " + " ".join(fmt(x) for x in p) + "\n")
        for tag, arr in blocks:
            for p in arr:
                f.write(tag + " " + " ".join(fmt(x) for x in p) + "\n")
        cur = None
        for idx, mat, vt, vn in F:
            if mat != cur:
                f.write(f"usemtl m{mat}\n"); cur = mat
            if order_vt_first:
                f.write("f " + " ".join(f"{idx[k]}/{vt[k]}/{vn[k]}" for k in range(3)) + "\n")
            else:
                f.write("f " + " ".join(f"{idx[k]}/{vn[k]}/{vt[k]}" for k in range(3)) + "\n")
    # lights
    n_light = int(rng.choice([0, 1, 1, 2, 3, 6, 8]))
    n_light = min(n_light, n_mat)
    light_mats = rng.permutation(n_mat)[:n_light]
    used = {m for _, m, _, _ in F}
    lights = "\n".join(f'<light mtlname="m{m}" radiance="{fmt(rng.uniform(0, 20))},{fmt(rng.uniform(0, 20))},{fmt(rng.uniform(0, 20))}"/>' for m in light_mats)
    # camera
    vv = np.array(V)
    centre = vv.mean(0)
    ext = max(float(np.abs(vv - centre).max()), 1e-3)
    mode = rng.choice(["outside", "outside", "inside", "away", "on_vertex"])
    look = centre + rng.uniform(-0.3, 0.3, 3) * ext
    eye = centre + rng.normal(size=3) * ext * (2.5 if mode != "inside" else 0.3)
    if mode == "away":
        look = eye + (eye - centre)
    if mode == "on_vertex":
        eye = vv[int(rng.integers(0, len(vv)))]
    fovy = float(rng.choice([1.0, 20.0, 40.0, 90.0, 170.0]))
    w, h = int(rng.integers(2, 41)), int(rng.integers(2, 31))
    with open(os.path.join(d, "s.xml"), "w") as f:
        f.write(f'<?xml version="1.0" encoding="utf-8"?>\n<camera type="perspective" width="{w}" height="{h}" fovy="{fmt(fovy)}">\n'
                f'\t<eye x="{fmt(eye[0])}" y="{fmt(eye[1])}" z="{fmt(eye[2])}"/>\n\t<lookat x="{fmt(look[0])}" y="{fmt(look[1])}" z="{fmt(look[2])}"/>\n'
                f'\t<up x="0.0" y="1.0" z="0.0"/>\n</camera>\n{lights}\n')
    desc = dict(tris=len(F), mats=[m["kind"] for m in mats], lights=[int(m) for m in light_mats], lights_without_triangles=[int(m) for m in light_mats if m not in used],
                scale=scale, camera=str(mode), fovy=fovy, size=(w, h))
    return desc, w, h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--gpu", action="store_true", help="render with the kernels through the C-ABI as well (an MI355X)")
    ap.add_argument("--keep", default=None, help="copy the files of a mismatching scene here")
    a = ap.parse_args()
    import hostsim_lib as H
    rng = np.random.default_rng(a.seed)
    t0 = t_print = time.time()
    n = n_render = refused = 0
    while time.time() - t0 < a.seconds:
        if time.time() - t_print > 60:
            print(f"... {n} scenes, {n_render} renders so far, all identical ({time.time() - t0:.0f} s)", flush=True)
            t_print = time.time()
        d = tempfile.mkdtemp(prefix="trt_fuzz_scene_")
        try:
            desc, w, h = write_random_scene(d, rng, big=a.gpu)
            leaf = int(rng.choice([1, 2, 2, 3, 4, 8, 8, 15]))
            builder = str(rng.choice(["sweep", "binned", "auto"] + (["lbvh", "lbvh"] if a.gpu else [])))  # with --gpu: the GPU builder's trees too (leaves of <= 2)
            try:
                s = T.Scene.load(os.path.join(d, "s.xml"), os.path.join(d, "s.obj"), os.path.join(d, "s.mtl"), d, 0, 0)
                s.build_bvh(leaf, builder)
            except Exception as e:  # the loader refuses (e.g. a light without area): fine, as long as it says so
                refused += 1
                continue
            n += 1
            for _ in range(2):
                flags = 0
                if rng.random() < 0.3: flags |= T.TRT_FLAG_FIXED_NEE
                if rng.random() < 0.2: flags |= T.TRT_FLAG_FIXED_PIXELS
                if rng.random() < 0.3: flags |= T.TRT_FLAG_RAY_OFFSET
                if rng.random() < 0.25: flags |= T.TRT_FLAG_SPECULAR_KS
                if rng.random() < 0.3: flags |= T.TRT_FLAG_OVERLAP
                p = T.make_params(w, h, int(rng.choice([1, 2, 5, 16])), int(rng.integers(0, 2 ** 32)), max_depth=int(rng.choice([0, 0, 0, 1, 3])), flags=flags)
                if rng.random() < 0.3:  # a tile and a row interleave of it
                    x0 = int(rng.integers(0, w - 1)); y0 = int(rng.integers(0, h - 1))
                    rows = (int(rng.choice([1, 2])), int(rng.choice([2, 3])), 0) if rng.random() < 0.5 else None
                    p = T.make_params(w, h, p.spp, p.seed, tile=(x0, y0, int(rng.integers(x0 + 1, w + 1)), int(rng.integers(y0 + 1, h + 1))), rows=rows, max_depth=p.max_depth, flags=flags)
                    if not T.rows_selected(p):
                        continue
                ref, ost = O.render(s.flat, p)
                want = [ost.rays_camera, ost.rays_shadow, ost.rays_indirect]
                results = []
                for nk in (0, 1):
                    old = H.set_node_kind(nk)
                    try:
                        img, rays = H.render(s.flat, p)
                    finally:
                        H.set_node_kind(old)
                    results.append((f"device code on the CPU, node kind {nk}", img, rays))
                if a.gpu:
                    # (TRT_TAIL_N=16: images this small would otherwise leave every bounce after the first to k_tail; with it the queue kernels run them all)
                    for env in ({}, {"TRT_NODE_KIND": "0", "TRT_TRACE_IMPL": "3"}, {"TRT_NODE_KIND": "1", "TRT_TRACE_IMPL": "3"}, {"TRT_TAIL_N": "16"},
                                {"TRT_NODE_KIND": str(int(rng.integers(0, 2))), "TRT_TRACE_IMPL": "3", "TRT_TAIL_N": "16"}):
                        os.environ.update(env)
                        try:
                            r = T.Renderer(s, 0)
                        finally:
                            for k in env:
                                del os.environ[k]
                        img, st = r.render(p)
                        r.close()
                        results.append((f"kernels {env}", img, [st.rays_camera, st.rays_shadow, st.rays_indirect]))
                if a.gpu and rng.random() < 0.3 and p.spp >= 2:  # the progressive entry point: the same samples in two or three calls, resumed from the sums
                    r = T.Renderer(s, 0)
                    cuts = sorted(set([0, p.spp] + [int(x) for x in rng.integers(1, p.spp, 2)]))
                    acc, img = None, None
                    for b, e in zip(cuts[:-1], cuts[1:]):
                        img, acc, _ = r.render_samples(p, b, e, acc)
                    r.close()
                    results.append((f"trt_render_samples in {len(cuts) - 1} calls", img, want))
                if a.gpu and rng.random() < 0.25 and (p.x0, p.y0, p.x1, p.y1) == (0, 0, w, h) and p.row_mod <= 1:  # a device group of 2-3 handles on the one GPU, interleaved stripes
                    g = T.GroupRenderer(s, [0] * int(rng.choice([2, 3])))
                    pg = T.make_params(w, h, p.spp, p.seed, max_depth=p.max_depth, flags=flags)
                    pg.row_block = int(rng.choice([1, 2, 8]))
                    img, stg, _ = g.render(pg)
                    g.close()
                    results.append((f"device group of {len(g.devices)}, row_block {pg.row_block}", img, [stg.rays_camera, stg.rays_shadow, stg.rays_indirect]))
                for what, img, rays in results:
                    n_render += 1
                    if not (np.array_equal(ref.view(np.uint32), img.view(np.uint32)) and rays == want):
                        bad = int((ref.view(np.uint32) != img.view(np.uint32)).any(-1).sum())
                        print("MISMATCH:", what, "leaf", leaf, builder, "flags", flags, "spp", p.spp, "seed", p.seed, "max_depth", p.max_depth, desc, f"{bad} pixels differ, rays {rays} vs {want}", flush=True)
                        if a.keep:
                            shutil.copytree(d, a.keep, dirs_exist_ok=True)
                        return 1
            s.close()
        finally:
            shutil.rmtree(d, ignore_errors=True)
    print(f"fuzz scenes: {n} random scenes ({refused} refused by the loader), {n_render} renders, all bit-identical to the oracle ({time.time() - t0:.0f} s)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
