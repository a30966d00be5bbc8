"""Parity of the HIP path (through the C-ABI of include/trt.h) with the CPU oracle — MI355X only.

Bar: bit-exact.  The formulation makes every fp32 operation of a path the same on both
sides, so images, hit records and ray counts must be identical; `TOL` below is the stated fp32
per-pixel tolerance of BASELINE.json's north star and stays at zero unless a test says otherwise.
"""
import ctypes as C
import os

import numpy as np
import pytest

import hostsim_lib as H
import oracle_lib as O
import raygen
import scene_util as SU
import tinyraytracing_amd as T
from conftest import get_scene

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 0.0  # max |gpu - oracle| per channel, linear radiance


def assert_same_image(a, b, what=""):
    assert a.shape == b.shape, what
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    assert np.isfinite(a).all(), what
    assert d.max() <= TOL, f"{what}: max abs diff {d.max()} in {int((d.max(-1) > TOL).sum())} pixels"


# ------------------------------------------------------------------ ray batches: traverseBVH
@pytest.mark.parametrize("name", ["back", "veach-mis", "staircase"])
def test_trace_matches_golden(name, renderer_factory):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    s = get_scene(name, int(g["width"]), int(g["height"]))
    r = renderer_factory(s)
    t, tri, uv = r.trace_closest(g["org"], g["dir"])
    assert np.array_equal(tri, g["tri"])
    assert np.array_equal(t, g["t"]) and np.array_equal(uv, g["uv"])


@pytest.mark.parametrize("name,n", [("back", 200000), ("staircase", 200000)])
def test_trace_matches_oracle_on_incoherent_rays(name, n, renderer_factory):
    s = get_scene(name, 64, 36)
    lo, hi = raygen.scene_bounds(s)
    org, dirs = raygen.random_rays(n, lo - 5, hi + 5, seed=99)
    t0, tri0, uv0, ost = O.trace(s.flat, org, dirs, want_stats=True)
    t1, tri1, uv1, st = renderer_factory(s).trace_closest(org, dirs, want_stats=True)
    assert np.array_equal(tri0, tri1) and np.array_equal(t0, t1) and np.array_equal(uv0, uv1)
    # visit counts: the persistent drivers run, per ray, the very steps of the hostsim's per-lane traversal of the same node kind
    # (80-B oct nodes where the tree qualifies, else the exact 4-wide ones); the wave-uniform tiny-tree driver visits the
    # reference's unculled set like the oracle
    old = H.set_node_kind(1 if st.inner_node_bytes == 80 else 0)
    try:
        _, _, _, cnt = H.trace(s.flat, org, dirs)
    finally:
        H.set_node_kind(old)
    assert [st.inner_visits[0], st.tri_tests[0]] in (cnt, [ost.inner_visits[0], ost.tri_tests[0]])


@pytest.mark.parametrize("name", ["back", "veach-mis", "staircase"])
def test_trace_degenerate_rays(name, renderer_factory):
    """Zero direction components (1/d = inf) and origins on box planes (0 * inf = NaN in the slab test):
    the 4-wide nodes (trt_wide.h), their packed-fp32 box tests and the tiny-tree walk must all agree
    with the oracle's binary recursion."""
    s = get_scene(name, 64, 36)
    org, dirs = raygen.adversarial_rays(s, 100000)
    t0, tri0, uv0 = O.trace(s.flat, org, dirs)
    t1, tri1, uv1 = renderer_factory(s).trace_closest(org, dirs)
    assert np.array_equal(tri0, tri1) and np.array_equal(t0, t1) and np.array_equal(uv0, uv1)


def test_trace_foreign_bvh_with_boxes_that_do_not_nest():
    """A tree whose stored boxes do not contain their children's (legal through the C-ABI): the wide collapse must
    keep those nodes, the traversal must not cull by distance, and the hits must equal the oracle's."""
    import scene_util
    s = T.Scene.named("staircase", 64, 36)
    assert scene_util.shrink_some_boxes(s, 400) == 400
    # (2 M rays: culling by distance, which such a tree cannot have — trt_wide.h boxesNested —, went wrong for about two rays in a million; round 4)
    org, dirs = raygen.random_rays(2000000, np.array([-8, -1, -8], np.float32), np.array([8, 8, 8], np.float32), seed=101)
    t0, tri0, uv0 = O.trace(s.flat, org, dirs)
    r = T.Renderer(s, 0)
    t1, tri1, uv1 = r.trace_closest(org, dirs)
    assert np.array_equal(tri0, tri1) and np.array_equal(t0, t1) and np.array_equal(uv0, uv1)
    p = T.make_params(64, 36, 16, 9)  # ... and the shadow rays' search hint and the tail kernel on the same tree
    img, st = r.render(p)
    ref, ost = O.render(s.flat, p)
    r.close()
    assert_same_image(img, ref, "foreign tree")
    assert st.rays == ost.rays


def test_trace_soup_deep_bvh(renderer_factory):
    s = get_scene("soup", 64, 36, n=200000)
    assert s.arrays()["bvh_depth"] > 12
    lo, hi = raygen.scene_bounds(s)
    org, dirs = raygen.random_rays(100000, lo, hi, seed=5)
    o2, d2 = raygen.primary_rays(s, 64, 36)
    org, dirs = np.vstack([org, o2]), np.vstack([dirs, d2])
    t0, tri0, uv0 = O.trace(s.flat, org, dirs)
    t1, tri1, uv1 = renderer_factory(s).trace_closest(org, dirs)
    assert np.array_equal(tri0, tri1) and np.array_equal(t0, t1) and np.array_equal(uv0, uv1)


# The persistent traversal kernels walk one of two node kinds (trt_create picks; TRT_NODE_KIND forces): 0 = exact 4-wide nodes,
# 1 = 8-wide nodes with quantised boxes (trt_oct.h).  TRT_TRACE_IMPL=3 puts a tiny tree on the per-lane driver too.
NODE_KINDS = ["0", "1"]


@pytest.mark.parametrize("nk", NODE_KINDS)
def test_every_node_kind_gives_the_same_image(nk, monkeypatch):
    monkeypatch.setenv("TRT_NODE_KIND", nk)
    s = get_scene("veach-mis", 96, 54)
    r = T.Renderer(s, 0)
    p = T.make_params(96, 54, 8, 77, flags=T.TRT_FLAG_COUNT)
    img, st = r.render(p)
    ref, ost = O.render(s.flat, p)
    r.close()
    assert st.inner_node_bytes == (80 if nk == "1" else 128)
    assert_same_image(img, ref, f"node kind {nk}")
    assert st.rays == ost.rays


def test_tiny_scene_uniform_walk_and_forced_per_ray_traversal(monkeypatch):
    s = get_scene("back", 64, 64)
    p = T.make_params(64, 64, 8, 5)
    ref, _ = O.render(s.flat, p)
    for impl, nk, nbytes in (("0", "0", 64), ("3", "0", 128), ("3", "1", 80)):  # leaves of 8 triangles: several slots of the oct nodes each (round 4)
        monkeypatch.setenv("TRT_TRACE_IMPL", impl)
        monkeypatch.setenv("TRT_NODE_KIND", nk)
        r = T.Renderer(s, 0)
        img, st = r.render(p)
        r.close()
        assert st.inner_node_bytes == nbytes, (impl, nk, st.inner_node_bytes)
        assert_same_image(img, ref, f"back impl {impl} node kind {nk}")
    s2 = T.Scene.named("back", 64, 64, leaf_num=2)  # leaves of 2: the oct tree on the tiny scene
    monkeypatch.setenv("TRT_TRACE_IMPL", "3")
    monkeypatch.setenv("TRT_NODE_KIND", "1")
    r = T.Renderer(s2, 0)
    img, st = r.render(p)
    r.close()
    assert st.inner_node_bytes == 80
    assert_same_image(img, O.render(s2.flat, p)[0], "back, leaves of 2, oct nodes")
    s2.close()


# ------------------------------------------------------------------ images: the whole loop
@pytest.mark.parametrize("name", ["back", "veach-mis", "staircase"])
def test_image_matches_golden(name, renderer_factory):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    w, h, spp, seed = int(g["width"]), int(g["height"]), int(g["spp"]), int(g["seed"])
    s = get_scene(name, w, h)
    img, st = renderer_factory(s).render(T.make_params(w, h, spp, seed))
    assert_same_image(img, g["image"], name)
    assert [st.rays_camera, st.rays_shadow, st.rays_indirect] == g["rays"].tolist()


@pytest.mark.parametrize("name,w,h,spp", [("back", 128, 128, 32), ("veach-mis", 160, 90, 16), ("staircase", 160, 90, 16)])
def test_image_matches_oracle(name, w, h, spp, renderer_factory):
    s = get_scene(name, w, h)
    p = T.make_params(w, h, spp, 0xABCDEF, flags=T.TRT_FLAG_COUNT | T.TRT_FLAG_TIMING)
    img, st = renderer_factory(s).render(p)
    ref, ost = O.render(s.flat, p)
    assert_same_image(img, ref, name)
    assert (st.rays_camera, st.rays_shadow, st.rays_indirect, st.shaded_hits) == (ost.rays_camera, ost.rays_shadow, ost.rays_indirect, ost.shaded_hits)
    assert st.max_bounces == ost.max_bounces
    assert st.inner_visits[0] > 0 and st.tri_tests[0] > 0 and st.inner_visits[1] > 0
    assert st.kernel_ms[1] > 0 and st.render_ms > 0


def test_sample_chunking_is_invisible(renderer_factory):
    """A small mem_budget splits the spp into several passes; the result must not change."""
    s = get_scene("back", 96, 96)
    r = renderer_factory(s)
    big, st_big = r.render(T.make_params(96, 96, 24, 3))
    per_path = 2 * 48 + 16 + 16 + 48 + 4  # two ray queues, hit, Lacc, one shadow queue, redo list
    small, st_small = r.render(T.make_params(96, 96, 24, 3, mem_budget=96 * 96 * 5 * per_path))
    assert st_big.passes == 1 and st_small.passes == 5
    assert np.array_equal(big, small)
    assert st_big.rays == st_small.rays


def test_overlapped_passes_equal_serial_passes():
    """TRT_FLAG_OVERLAP (two passes in flight on two streams) vs one pass at a time: identical image and counters."""
    s = get_scene("veach-mis", 128, 72)
    out = []
    for flags in (T.TRT_FLAG_COUNT | T.TRT_FLAG_OVERLAP, T.TRT_FLAG_COUNT):
        r = T.Renderer(s, 0)
        img, st = r.render(T.make_params(128, 72, 12, 31, flags=flags))
        r.close()
        out.append((img, st.rays, st.shaded_hits, st.inner_visits[0], st.tri_tests[1]))
        assert st.passes >= (2 if flags & T.TRT_FLAG_OVERLAP else 1)
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1:] == out[1][1:]


def test_tiles_and_row_interleave_compose(renderer_factory):
    s = get_scene("back", 100, 60)
    r = renderer_factory(s)
    full, _ = r.render(T.make_params(100, 60, 6, 17))
    parts = np.zeros_like(full)
    for k in range(3):
        p = T.make_params(100, 60, 6, 17, rows=(8, 3, k))
        out, st = r.render(p)
        ys = T.rows_selected(p)
        assert st.rows_rendered == len(ys)
        parts[ys] = out
    assert np.array_equal(full, parts)
    tile, _ = r.render(T.make_params(100, 60, 6, 17, tile=(13, 7, 77, 41)))
    assert np.array_equal(tile, full[7:41, 13:77])
    one, _ = r.render(T.make_params(100, 60, 6, 17, tile=(50, 30, 51, 31)))
    assert np.array_equal(one, full[30:31, 50:51])


def test_render_into_device_tensor(renderer_factory):
    import torch
    s = get_scene("back", 64, 64)
    r = renderer_factory(s)
    p = T.make_params(64, 64, 4, 1)
    host, _ = r.render(p)
    out = torch.empty((64, 64, 3), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        st = r.render_into(p, out, stream.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), host) and st.rays > 0


# ------------------------------------------------------------------ BASELINE.json sizes, size-independent properties
def test_config2_full_size_tile_vs_oracle_and_determinism(renderer_factory):
    """Config 2: back, 1024x1024, 256 spp.  The streams are keyed by the global (pixel, sample), so a
    tile cut out of the full-size render equals the oracle's render of just that tile; the full render
    is also reproducible bit for bit and equals the union of 8 interleaved stripes."""
    s = get_scene("back", 1024, 1024)
    r = renderer_factory(s)
    p = T.make_params(1024, 1024, 256, T.SEED_BACK)
    full, st = r.render(p)
    assert np.isfinite(full).all() and st.rays_camera == 1024 * 1024 * 256
    for (x0, y0) in ((500, 500), (40, 900), (1000, 8)):
        pt = T.make_params(1024, 1024, 256, T.SEED_BACK, tile=(x0, y0, x0 + 12, y0 + 12))
        ref, _ = O.render(s.flat, pt)
        assert_same_image(full[y0:y0 + 12, x0:x0 + 12], ref, f"tile {x0},{y0}")
    again, st2 = r.render(p)
    assert np.array_equal(full, again) and st.rays == st2.rays
    # converged image statistics vs the reference's own snapshots of this scene (+-30 %)
    lin = (T.tonemap(full).astype(np.float64) / 255) ** 2.2
    mean = lin.reshape(-1, 3).mean(0)
    ref_mean = np.array([(0.220 + 0.237) / 2, (0.218 + 0.243) / 2, (0.071 + 0.098) / 2])
    assert np.all(np.abs(mean / ref_mean - 1) < 0.30), mean


def test_config2_stripes_union(renderer_factory):
    s = get_scene("back", 1024, 1024)
    r = renderer_factory(s)
    full, _ = r.render(T.make_params(1024, 1024, 32, T.SEED_BACK))
    parts = np.empty_like(full)
    for k in range(8):
        p = T.make_params(1024, 1024, 32, T.SEED_BACK, rows=(8, 8, k))
        parts[T.rows_selected(p)] = r.render(p)[0]
    assert np.array_equal(full, parts)


def test_config3_soup_tile_vs_oracle(renderer_factory):
    """Config 3 (deep-BVH stress) at its stated size — 1 M random triangles, 1920x1080, 64 spp: tiles against the oracle."""
    s = get_scene("soup", 1920, 1080, n=1_000_000)
    assert s.info["n_triangles"] >= 1_000_000
    r = renderer_factory(s)
    for (x0, y0) in ((960, 540), (100, 1000)):
        pt = T.make_params(1920, 1080, 64, T.SEED_SOUP, tile=(x0, y0, x0 + 16, y0 + 8))
        img, st = r.render(pt)
        ref, ost = O.render(s.flat, pt)
        assert_same_image(img, ref, f"soup tile {x0},{y0}")
        assert st.rays == ost.rays


def test_config5_ten_million_triangles_at_4k_tile_vs_oracle():
    """Config 5's scene at full size — 10 M triangles, 3840 x 2160 — through the binned-SAH builder, the wide-node
    collapse, the spilling stack and the scheduler driver: tiles against the oracle (which walks the same tree),
    and full-frame properties (every camera ray accounted for; two renders bit-identical)."""
    s = T.Scene.named("blob", 3840, 2160, n=10_000_000)
    assert s.info["n_triangles"] >= 10_000_000
    r = T.Renderer(s, 0)
    try:
        for (x0, y0) in ((1900, 1000), (2300, 1500), (40, 2100)):
            pt = T.make_params(3840, 2160, 16, T.SEED_BLOB, tile=(x0, y0, x0 + 16, y0 + 8))
            img, st = r.render(pt)
            ref, ost = O.render(s.flat, pt)
            assert_same_image(img, ref, f"blob-10M tile {x0},{y0}")
            assert st.rays == ost.rays
        # the stated sample count: 4096 spp (sample indices >= 1024, the L / 4096.0f scaling) on a 4 x 4 tile
        pt = T.make_params(3840, 2160, 4096, T.SEED_BLOB, tile=(1930, 1100, 1934, 1104))
        img, st = r.render(pt)
        ref, ost = O.render(s.flat, pt)
        assert_same_image(img, ref, "blob-10M 4x4 tile at 4096 spp")
        assert st.rays == ost.rays and st.rays_camera == 16 * 4096
        full = T.make_params(3840, 2160, 2, T.SEED_BLOB)
        a, sa = r.render(full)
        b, sb = r.render(full)
        assert sa.rays_camera == 3840 * 2160 * 2 and sa.rays == sb.rays
        assert np.array_equal(a, b) and np.isfinite(a).all() and a.max() > 0
        assert np.array_equal(a[1000:1008, 1900:1916], r.render(T.make_params(3840, 2160, 2, T.SEED_BLOB, tile=(1900, 1000, 1916, 1008)))[0])
    finally:
        r.close()
        s.close()


def test_config4_staircase_tile_at_1024spp_vs_oracle(renderer_factory):
    """Config 4 (largest cg22 scene, 1920x1080, 1024 spp): tiles of the full-size frame against the oracle.
    6 lights (Q3 CDF quirk), 3 textures, glass (Fresnel / TIR), Phong lobes up to Ns = 1000."""
    s = get_scene("staircase", 1920, 1080)
    r = renderer_factory(s)
    for (x0, y0) in ((900, 500), (300, 900)):
        pt = T.make_params(1920, 1080, 1024, T.SEED_STAIRCASE, tile=(x0, y0, x0 + 8, y0 + 6))
        img, st = r.render(pt)
        ref, ost = O.render(s.flat, pt)
        assert_same_image(img, ref, f"staircase tile {x0},{y0}")
        assert st.rays == ost.rays and st.max_bounces == ost.max_bounces


# ------------------------------------------------------------------ edge cases and error behaviour
def test_tiny_tree_with_children_before_parents(scene_factory):
    """the wave-uniform walk of tiny trees evaluates nodes in index order, which is only right when every
    inner child follows its parent.  A caller's tree that does not (here: `back` with its nodes renumbered in reverse)
    must still give the oracle's hits and image — trt_create walks it per lane instead."""
    s = T.Scene.named("back", 64, 64)
    assert SU.renumber_nodes_reversed(s) > 0
    r = T.Renderer(s, 0)
    try:
        lo, hi = raygen.scene_bounds(s)
        org, dirs = raygen.random_rays(50000, lo - 5, hi + 5, seed=5)
        t0, tri0, uv0 = O.trace(s.flat, org, dirs)
        t1, tri1, uv1 = r.trace_closest(org, dirs)
        assert np.array_equal(tri0, tri1) and np.array_equal(t0, t1) and np.array_equal(uv0, uv1)
        p = T.make_params(64, 64, 16, T.SEED_BACK)
        img, st = r.render(p)
        ref, ost = O.render(s.flat, p)
        assert_same_image(img, ref, "back with reversed node order")
        assert st.rays == ost.rays
    finally:
        r.close()
        s.close()


def test_failure_mid_render_leaves_the_handle_usable(monkeypatch):
    """an error return with kernels in flight (here an injected one, right after bounce 0 of the first
    of two overlapped passes has been issued) must drain the streams; the same handle then renders the oracle's image."""
    s = get_scene("veach-mis", 96, 54)
    monkeypatch.setenv("TRT_TEST_FAIL_AT_BOUNCE", "0")
    r = T.Renderer(s, 0)
    monkeypatch.delenv("TRT_TEST_FAIL_AT_BOUNCE")
    p = T.make_params(96, 54, 8, 0x5EED0002, flags=T.TRT_FLAG_OVERLAP)
    try:
        with pytest.raises(T.TrtError, match="injected failure"):
            r.render(p)
        for _ in range(2):
            img, st = r.render(p)
            ref, ost = O.render(s.flat, p)
            assert_same_image(img, ref, "render after a failed call")
            assert st.rays == ost.rays
        tiny = T.make_params(96, 54, 8, 0x5EED0002, flags=T.TRT_FLAG_OVERLAP, mem_budget=1024)
        with pytest.raises(T.TrtError, match="mem_budget too small"):
            r.render(tiny)
        assert_same_image(r.render(p)[0], ref, "render after TRT_ENOMEM")
    finally:
        r.close()


def test_empty_and_single_triangle_scenes(tmp_path):
    SU.write_scene(tmp_path, "empty", "v 0 0 0\n", SU.MTL_BASIC, w=16, h=16)
    s = SU.load(tmp_path, "empty")
    assert s.info["n_triangles"] == 0
    r = T.Renderer(s, 0)
    img, st = r.render(T.make_params(16, 16, 2, 1))
    assert not img.any() and st.rays == 16 * 16 * 2 and st.shaded_hits == 0
    t, tri, uv = r.trace_closest(np.zeros((5, 3), np.float32), np.tile(np.array([0, 0, -1], np.float32), (5, 1)))
    assert (tri == -1).all() and (t == np.float32(T._abi.TRT_INF)).all()
    r.close()
    one = "v -1 -1 0\nv 1 -1 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\nusemtl lamp\nf 1/1/1 2/1/1 3/1/1\n"
    SU.write_scene(tmp_path, "one", one, SU.MTL_BASIC, lights=[("lamp", (3, 2, 1))], w=16, h=16)
    s1 = SU.load(tmp_path, "one")
    r1 = T.Renderer(s1, 0)
    p = T.make_params(16, 16, 4, 2)
    img, st = r1.render(p)
    ref, _ = O.render(s1.flat, p)
    assert np.array_equal(img, ref) and img.max() == 3.0     # camera rays return the raw radiance (pathTracing.cpp:9-12)
    r1.close()


def test_glass_specular_texture_scene(tmp_path, renderer_factory):
    """All lobes at once: TRANSMISSION / TIR through a glass slab, Phong lobe, two lights (Q3 CDF quirk)."""
    v = ["vt 0 0", "vt 1 0", "vt 1 1", "vt 0 1", "vn 0 1 0", "vn 0 -1 0", "vn 0 0 1"]
    obj = "\n".join(v) + "\n"
    verts = ["-4 0 -4", "4 0 -4", "4 0 4", "-4 0 4",          # floor (shiny)
             "-1 3 -1", "1 3 -1", "1 3 1", "-1 3 1",           # lamp A
             "2 2.5 -0.3", "2.6 2.5 -0.3", "2.6 2.5 0.3", "2 2.5 0.3",   # lamp B (smaller)
             "-1.5 0.8 -1.5", "1.5 0.8 -1.5", "1.5 0.8 1.5", "-1.5 0.8 1.5",   # glass slab top
             "-1.5 0.5 -1.5", "1.5 0.5 -1.5", "1.5 0.5 1.5", "-1.5 0.5 1.5"]   # glass slab bottom
    obj += "\n".join("v " + x for x in verts) + "\n"
    obj += "usemtl shiny\nf 1/1/1 3/3/1 2/2/1\nf 1/1/1 4/4/1 3/3/1\n"
    obj += "usemtl lamp\nf 5/1/2 6/2/2 7/3/2\nf 5/1/2 7/3/2 8/4/2\n"
    obj += "usemtl lamp2\nf 9/1/2 10/2/2 11/3/2\nf 9/1/2 11/3/2 12/4/2\n"
    obj += "usemtl glass\nf 13/1/1 15/3/1 14/2/1\nf 13/1/1 16/4/1 15/3/1\nf 17/1/2 18/2/2 19/3/2\nf 17/1/2 19/3/2 20/4/2\n"
    mtl = SU.MTL_BASIC + "newmtl lamp2\nKd 0 0 0\nKs 0 0 0\nNs 1\nNi 1\n"
    SU.write_scene(tmp_path, "mix", obj, mtl, lights=[("lamp", (20, 20, 20)), ("lamp2", (40, 30, 20))], w=64, h=48, fovy=45, eye=(0, 2.2, 6), lookat=(0, 0.6, 0))
    s = SU.load(tmp_path, "mix")
    r = T.Renderer(s, 0)
    p = T.make_params(64, 48, 32, 11)
    img, st = r.render(p)
    ref, ost = O.render(s.flat, p)
    assert_same_image(img, ref, "mix")
    assert st.rays == ost.rays and st.rays_indirect > 0
    r.close()


def test_error_codes(renderer_factory):
    s = get_scene("back", 64, 64)
    r = renderer_factory(s)
    lib = T._abi.load_hip()
    out = np.zeros((64, 64, 3), np.float32)
    fp = out.ctypes.data_as(C.POINTER(C.c_float))
    for bad in (T.make_params(64, 64, 0, 1), T.make_params(1, 64, 1, 1), T.make_params(64, 64, 1, 1, tile=(0, 0, 65, 64)),
                T.make_params(64, 64, 1, 1, tile=(10, 10, 10, 20)), T.make_params(64, 64, 1, 1, rows=(0, 2, 0)),
                T.make_params(64, 64, 1, 1, rows=(8, 2, 2)), T.make_params(64, 64, 1, 1, max_depth=-1)):
        rc = lib.trt_render(r._h, C.byref(bad), fp, None)
        assert rc == 1 and lib.trt_last_error()               # TRT_EINVAL + message, never exit()
    assert lib.trt_render(r._h, C.byref(T.make_params(64, 64, 1, 1)), None, None) == 1
    tiny = T.make_params(64, 64, 1, 1, mem_budget=1000)
    assert lib.trt_render(r._h, C.byref(tiny), fp, None) == 3    # TRT_ENOMEM: budget below one sample per pixel
    h = C.c_void_p()
    assert lib.trt_create(None, 0, C.byref(h)) == 1
    assert lib.trt_create(s.flat, 99, C.byref(h)) == 4           # TRT_ENODEV
    # a corrupted BVH is rejected on the host, before any kernel could chase a bad index
    from tinyraytracing_amd._abi import BvhNode, SceneFlat
    f = s.flat.contents
    g = SceneFlat()
    C.memmove(C.byref(g), C.byref(f), C.sizeof(SceneFlat))
    nodes = (BvhNode * f.n_nodes)()
    C.memmove(nodes, f.nodes, C.sizeof(BvhNode) * f.n_nodes)
    nodes[0].child0 = 12345
    g.nodes = nodes
    assert lib.trt_create(C.byref(g), 0, C.byref(h)) == 1 and b"bvh" in lib.trt_last_error()
    nodes[0].child0 = 0                                          # a cycle
    assert lib.trt_create(C.byref(g), 0, C.byref(h)) == 1
    C.memmove(nodes, f.nodes, C.sizeof(BvhNode) * f.n_nodes)     # siblings swapped: child0's triangles no longer come first
    nodes[0].child0, nodes[0].child1 = f.nodes[0].child1, f.nodes[0].child0
    assert lib.trt_create(C.byref(g), 0, C.byref(h)) == 1 and b"post-BVH order" in lib.trt_last_error()


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


# ------------------------------------------------------------------ progressive / resumable render
def test_render_samples_resumes_bit_exactly(renderer_factory):
    """trt_render_samples over [0,5), [5,6), [6,16) from zeros == one trt_render of 16 spp == the oracle; the
    accumulator may take a round trip through a file in between (checkpoint)."""
    s = get_scene("veach-mis", 80, 45)
    r = renderer_factory(s)
    p = T.make_params(80, 45, 16, 0xC0FFEE)
    full, st_full = r.render(p)
    ref, ost = O.render(s.flat, p)
    assert np.array_equal(full, ref)
    acc, rays = None, 0
    for a, b in ((0, 5), (5, 6), (6, 16)):
        img, acc, st = r.render_samples(p, a, b, acc)
        acc = np.frombuffer(acc.tobytes(), dtype=np.float64).reshape(acc.shape).copy()  # what a checkpoint file holds
        rays += st.rays
        if b < 16:
            assert not np.array_equal(img, full)
    assert np.array_equal(img, full)
    assert rays == st_full.rays == ost.rays
    # overlapping passes (two slots) leave the same sums
    img2, acc2, _ = r.render_samples(T.make_params(80, 45, 16, 0xC0FFEE, flags=T.TRT_FLAG_OVERLAP), 0, 16, None)
    assert np.array_equal(img2, full) and np.array_equal(acc2, acc)
    for bad in ((-1, 4), (4, 4), (3, 17)):
        with pytest.raises(T.TrtError):
            r.render_samples(p, bad[0], bad[1], None)


def test_cli_progressive_render_resumes_from_its_checkpoint(tmp_path):
    """tinyrt (host/main.cpp = the reference's main()): one-shot PNG == PNG of a run that was stopped after 6 of
    16 samples and resumed from its accumulator file == the library's image through tonemap (main.cpp:34)."""
    import subprocess
    exe = os.path.join(os.path.dirname(T.__file__), "lib", "tinyrt")
    d = os.path.join(os.path.dirname(T.__file__), "..", "scenes", "back")
    base = [exe, d, os.path.join(d, "back.mtl"), os.path.join(d, "back.xml"), os.path.join(d, "back.obj"), "16", "--width", "96", "--height", "54", "--seed", "77"]
    a, b, ck = str(tmp_path / "a.png"), str(tmp_path / "b.png"), str(tmp_path / "b.acc")
    subprocess.run(base + ["--out", a], check=True, capture_output=True, timeout=300)
    subprocess.run(base + ["--out", b, "--every", "3", "--checkpoint", ck, "--stop-after", "6"], check=True, capture_output=True, timeout=300)
    partial = open(b, "rb").read()
    assert partial != open(a, "rb").read()
    subprocess.run(base + ["--out", b, "--every", "5", "--checkpoint", ck], check=True, capture_output=True, timeout=300)
    assert open(b, "rb").read() == open(a, "rb").read()
    # a finished checkpoint reproduces the picture without rendering; one of another render is refused
    subprocess.run(base + ["--out", b, "--checkpoint", ck], check=True, capture_output=True, timeout=300)
    assert open(b, "rb").read() == open(a, "rb").read()
    bad = subprocess.run(base[:-1] + ["78", "--out", b, "--checkpoint", ck], capture_output=True, timeout=300)
    assert bad.returncode != 0 and b"another render" in bad.stderr
    # the sums of one estimator / one scene must not be continued with another
    bad = subprocess.run(base + ["--out", b, "--checkpoint", ck, "--fixed-nee"], capture_output=True, timeout=300)
    assert bad.returncode != 0 and b"another estimator" in bad.stderr
    dv = os.path.join(os.path.dirname(T.__file__), "..", "scenes", "veach-mis")
    other = [exe, dv, os.path.join(dv, "veach-mis.mtl"), os.path.join(dv, "veach-mis.xml"), os.path.join(dv, "veach-mis.obj"), "16", "--width", "96", "--height", "54", "--seed", "77"]
    bad = subprocess.run(other + ["--out", b, "--checkpoint", ck], capture_output=True, timeout=300)
    assert bad.returncode != 0 and b"another scene" in bad.stderr
    # a run that stops early says that its picture is not normalised yet
    ck2 = str(tmp_path / "c.acc")
    part = subprocess.run(base + ["--out", b, "--every", "4", "--checkpoint", ck2, "--stop-after", "4"], capture_output=True, timeout=300)
    assert part.returncode == 0 and b"stopped after 4 of 16 samples" in part.stderr
    s = get_scene("back", 96, 54)
    img, _ = T.Renderer(s, 0).render(T.make_params(96, 54, 16, 77))
    T.imshow(img, str(tmp_path / "c.png"))
    assert open(str(tmp_path / "c.png"), "rb").read() == open(a, "rb").read()


# ------------------------------------------------------------------ TRT_FLAG_FIXED_NEE (opt-out of Q3-Q5)
@pytest.mark.parametrize("name,w,h,spp", [("back", 96, 54, 16), ("veach-mis", 96, 54, 8), ("staircase", 64, 36, 8)])
def test_fixed_nee_image_matches_oracle(name, w, h, spp, renderer_factory):
    s = get_scene(name, w, h)
    p = T.make_params(w, h, spp, 0xF1DE, flags=T.TRT_FLAG_FIXED_NEE)
    img, st = renderer_factory(s).render(p)
    ref, ost = O.render(s.flat, p)
    assert (st.rays_camera, st.rays_shadow, st.rays_indirect) == (ost.rays_camera, ost.rays_shadow, ost.rays_indirect)
    assert np.array_equal(img, ref)


@pytest.mark.parametrize("nk", NODE_KINDS)
def test_fixed_nee_on_every_node_kind(nk, monkeypatch):
    """The occlusion test (stop at the first hit in front of the light sample) through both node kinds of the scheduler
    driver; 160 x 90 x 32 spp keeps the regular kernels (not only k_tail) busy."""
    monkeypatch.setenv("TRT_NODE_KIND", nk)
    s = get_scene("veach-mis", 160, 90)
    r = T.Renderer(s, 0)
    p = T.make_params(160, 90, 32, 31, flags=T.TRT_FLAG_FIXED_NEE)
    img, st = r.render(p)
    r.close()
    ref, ost = O.render(s.flat, p)
    assert np.array_equal(img, ref) and st.rays == ost.rays


def test_fixed_pixels_and_nee_together_match_oracle(renderer_factory):
    """TRT_FLAG_FIXED_PIXELS | TRT_FLAG_FIXED_NEE: the 'fixed' renderer a user would pick — same bar as parity mode."""
    s = get_scene("staircase", 80, 45)
    p = T.make_params(80, 45, 8, 4242, flags=T.TRT_FLAG_FIXED_PIXELS | T.TRT_FLAG_FIXED_NEE)
    img, st = renderer_factory(s).render(p)
    ref, ost = O.render(s.flat, p)
    assert np.array_equal(img, ref) and st.rays == ost.rays
    only_px = T.make_params(80, 45, 8, 4242, flags=T.TRT_FLAG_FIXED_PIXELS)
    a, _ = renderer_factory(s).render(only_px)
    b, _ = O.render(s.flat, only_px)
    assert np.array_equal(a, b) and not np.array_equal(a, img)


# ------------------------------------------------------------------ several GPUs behind the C-ABI (trt_group_*)
def test_device_group_on_one_gpu_equals_single_render():
    """the multi-GPU path of the C boundary — one host thread per device, interleaved row stripes, one
    gather to the first device, un-interleave — rehearsed on the one GPU of this box by naming device 0 several times
    (device copies stand in for ncclGather; everything else is the code an 8-GPU node runs).  Bit-identical to trt_render
    for every group size and stripe height, ray counts included."""
    s = get_scene("veach-mis", 160, 96)
    single = T.Renderer(s, 0)
    p = T.make_params(160, 96, 8, 0x5EED0002)
    ref, st = single.render(p)
    single.close()
    for n, rb in ((2, 8), (3, 4), (4, 1)):
        g = T.GroupRenderer(s, [0] * n)
        try:
            pg = T.make_params(160, 96, 8, 0x5EED0002)
            pg.row_block = rb
            img, gst, gms = g.render(pg)
            assert_same_image(img, ref, f"group of {n}, stripes of {rb} rows")
            assert (gst.rays_camera, gst.rays_shadow, gst.rays_indirect) == (st.rays_camera, st.rays_shadow, st.rays_indirect)
            assert gst.rows_rendered == 96 and gms >= 0.0
            # a tile that starts on a stripe boundary
            pt = T.make_params(160, 96, 8, 0x5EED0002, tile=(16, rb * n, 80, 96))
            pt.row_block = rb
            tile, _, _ = g.render(pt)
            assert_same_image(tile, ref[rb * n:96, 16:80], "group tile")
        finally:
            g.close()
    g = T.GroupRenderer(s, [0, 0])
    try:
        bad = T.make_params(160, 96, 8, 0x5EED0002, tile=(0, 3, 160, 96))
        with pytest.raises(T.TrtError, match="y0 must be a multiple"):
            g.render(bad)
    finally:
        g.close()


def test_group_gathers_through_rccl_with_one_rank(monkeypatch):
    """TRT_GROUP_FORCE_RCCL=1 sends a group of ONE device through the RCCL route — dlopen(librccl.so.1), the six
    dlsym-bound entry points with their hand-declared prototypes, ncclCommInitAll (rccl.h:236) for one device, ncclGroupStart / ncclGather
    (rccl.h:745, ncclFloat32 = 7, a communicator of size 1 gathering to itself) / ncclGroupEnd on the group's stream, then the un-interleave
    kernel ordered behind it — on this one-GPU box.  The image equals trt_render's bit for bit; a second render reuses the communicator."""
    monkeypatch.setenv("TRT_GROUP_FORCE_RCCL", "1")
    s = get_scene("veach-mis", 160, 96)
    single = T.Renderer(s, 0)
    p = T.make_params(160, 96, 8, 0x5EED0002)
    ref, st = single.render(p)
    single.close()
    g = T.GroupRenderer(s, [0])
    try:
        maps = open("/proc/self/maps").read()
        assert "librccl" in maps, "the RCCL route was not taken: librccl is not loaded"
        for _ in range(2):
            img, gst, gms = g.render(T.make_params(160, 96, 8, 0x5EED0002))
            assert_same_image(img, ref, "group of one through ncclGather")
            assert gst.rays == st.rays and gms >= 0.0
    finally:
        g.close()


def test_group_render_device_leaves_the_image_on_the_first_device():
    """trt_group_render_device (ABI v4): the gathered, un-interleaved image stays in device memory of devices[0]; the host threads of
    the group are created once (three renders on one group), and the result equals trt_render's."""
    import torch
    s = get_scene("staircase", 128, 72)
    single = T.Renderer(s, 0)
    ref, st = single.render(T.make_params(128, 72, 4, 17))
    single.close()
    g = T.GroupRenderer(s, [0, 0, 0])
    try:
        out = torch.zeros((72, 128, 3), dtype=torch.float32, device="cuda:0")
        for k in range(3):
            pg = T.make_params(128, 72, 4, 17)
            pg.row_block = 8
            gst, gms = g.render_into(pg, out)
            torch.cuda.synchronize()
            assert_same_image(out.cpu().numpy(), ref, f"device-resident group image, render {k}")
            assert gst.rays == st.rays
            out.zero_()
    finally:
        g.close()


def test_group_create_refuses_devices_the_node_does_not_have():
    import torch
    n = torch.cuda.device_count()
    s = get_scene("back", 32, 32)
    with pytest.raises(T.TrtError, match="device ordinal out of range"):
        T.GroupRenderer(s, [0, n])
    with pytest.raises(T.TrtError):
        T.GroupRenderer(s, [-1])


def test_cli_device_list(tmp_path):
    """tinyrt --devices 0,0: the C++ host entry (trt::render with RenderOpts::devices) through the group path."""
    import subprocess
    exe = os.path.join(T.REPO_ROOT, "tinyraytracing_amd", "lib", "tinyrt")
    d = os.path.join(T.SCENES_DIR, "back")
    outs = []
    for extra in ([], ["--devices", "0,0", "--row-block", "4"]):
        out = str(tmp_path / f"img{len(outs)}.png")
        r = subprocess.run([exe, d, os.path.join(d, "back.mtl"), os.path.join(d, "back.xml"), os.path.join(d, "back.obj"), "4", "--width", "64", "--height", "48",
                            "--out", out] + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1]


GLOO_HIP_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import tinyraytracing_amd as T
from tinyraytracing_amd import dist as D
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size={world})
scene = T.Scene.named("staircase", 96, 54)
r = T.Renderer(scene, 0)
p0 = D.shard_params(96, 54, 8, T.SEED_STAIRCASE, dist.get_rank(), {world})
out = torch.empty((len(T.rows_selected(p0)), 96, 3), dtype=torch.float32, device="cuda:0")
def render_fn(p):
    return out, r.render_into(p, out, torch.cuda.current_stream().cuda_stream)
img, st = D.render_distributed(render_fn, 96, 54, 8, T.SEED_STAIRCASE, dist=dist, device="cuda:0")
rays = torch.tensor([st.rays], dtype=torch.int64)
dist.all_reduce(rays)
if dist.get_rank() == 0:
    np.savez({out!r}, image=img.cpu().numpy(), rays=rays.numpy())
dist.destroy_process_group()
'''


def test_two_rank_gloo_render_with_the_hip_renderer(tmp_path):
    """dist.render_distributed (what bench.py --gpus N runs) driven by the HIP render_into on two
    ranks that share this box's GPU, gathered over gloo: equals the oracle's full image bit for bit."""
    import subprocess
    import sys
    world, port = 2, 31500 + (os.getpid() % 2000)
    out = str(tmp_path / "dist_hip.npz")
    script = str(tmp_path / "worker_hip.py")
    with open(script, "w") as f:
        f.write(GLOO_HIP_WORKER.format(root=T.REPO_ROOT, port=port, world=world, out=out))
    procs = [subprocess.Popen([sys.executable, script, str(k)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for k in range(world)]
    logs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = np.load(out)
    s = get_scene("staircase", 96, 54)
    ref, st = O.render(s.flat, T.make_params(96, 54, 8, T.SEED_STAIRCASE))
    assert np.array_equal(got["image"], ref)
    assert int(got["rays"][0]) == st.rays


@pytest.mark.parametrize("name", ["veach-mis", "staircase"])
def test_compressed_nodes_on_the_gpu(name, monkeypatch):
    """TRT_NODE_KIND=1: the 80-B 8-wide nodes with quantised boxes (trt_oct.h) — random, degenerate (zero direction components,
    origins on box planes: those axes drop out of the node test) and grazing rays, then an image; against the oracle, bit for bit.
    The same rays through the exact 4-wide nodes (TRT_NODE_KIND=0) for comparison of the visit counts."""
    s = get_scene(name, 96, 54)
    lo, hi = raygen.scene_bounds(s)
    org, dirs = raygen.random_rays(100000, lo - 5, hi + 5, seed=8)
    o2, d2 = raygen.adversarial_rays(s, 20000)
    o3, d3 = raygen.grazing_rays(s.flat, 50000, seed=3)
    allo, alld = np.vstack([org, o2, o3]), np.vstack([dirs, d2, d3])
    t0, tri0, uv0 = O.trace(s.flat, allo, alld)
    p = T.make_params(96, 54, 8, T.SEED_STAIRCASE)
    ref, ost = O.render(s.flat, p)
    visits = {}
    for nk in NODE_KINDS:
        monkeypatch.setenv("TRT_NODE_KIND", nk)
        r = T.Renderer(s, 0)
        try:
            t1, tri1, uv1, st = r.trace_closest(allo, alld, want_stats=True)
            assert st.inner_node_bytes == (80 if nk == "1" else 128)
            assert np.array_equal(tri0, tri1) and np.array_equal(t0, t1) and np.array_equal(uv0, uv1)
            visits[nk] = r.trace_closest(org, dirs, want_stats=True)[3].inner_visits[0]
            img, rst = r.render(p)
            assert_same_image(img, ref, f"node kind {nk}")
            assert rst.rays == ost.rays
        finally:
            r.close()
    assert visits["1"] <= visits["0"]  # eight children per visit


@pytest.mark.parametrize("name,tree", [("veach-mis", "reference-8"), ("staircase", "reference-8"), ("staircase", "sah-8"), ("veach-mis", "sah-15"), ("back", "sah-8")])
def test_the_references_own_leaf_size_takes_the_oct_nodes(name, tree, monkeypatch):
    """main.cpp:76 builds with up to 8 triangles per leaf.  That tree — from the reference's builder as the oracle restates it
    (bvh.cpp:16-144), or from this repository's with leaf_num 8 / 15 — is walked on the 80-B 8-wide nodes like any other (a leaf of more
    than 3 triangles = several slots with the leaf's own box, trt_oct_build.h): random, degenerate and grazing rays, then an image and its
    ray counts, against the oracle on the caller's tree, bit for bit."""
    if name == "back":
        monkeypatch.setenv("TRT_TRACE_IMPL", "3")  # the per-lane driver on the tiny tree (it walks the wave-uniform form otherwise)
    s = SU.load_with_reference_tree(name, 96, 54, 8) if tree == "reference-8" else T.Scene.named(name, 96, 54, leaf_num=int(tree.split("-")[1]))
    lo, hi = raygen.scene_bounds(s)
    org, dirs = raygen.random_rays(100000, lo - 5, hi + 5, seed=18)
    o2, d2 = raygen.adversarial_rays(s, 20000)
    o3, d3 = raygen.grazing_rays(s.flat, 50000, seed=13)
    allo, alld = np.vstack([org, o2, o3]), np.vstack([dirs, d2, d3])
    t0, tri0, uv0 = O.trace(s.flat, allo, alld)
    p = T.make_params(96, 54, 8, T.SEED_STAIRCASE)
    ref, ost = O.render(s.flat, p)
    r = T.Renderer(s, 0)
    try:
        t1, tri1, uv1, st = r.trace_closest(allo, alld, want_stats=True)
        assert st.inner_node_bytes == 80
        assert np.array_equal(tri0, tri1) and np.array_equal(t0, t1) and np.array_equal(uv0, uv1)
        img, rst = r.render(p)
        assert_same_image(img, ref, f"{name} {tree}")
        assert (rst.rays_camera, rst.rays_shadow, rst.rays_indirect) == (ost.rays_camera, ost.rays_shadow, ost.rays_indirect)
        pf = T.make_params(96, 54, 4, 12, flags=T.TRT_FLAG_FIXED_NEE | T.TRT_FLAG_RAY_OFFSET)
        assert_same_image(r.render(pf)[0], O.render(s.flat, pf)[0], f"{name} {tree} fixed NEE")
        # triangles per leaf step (sc.leaf_loop: 4 on such trees) is scheduling only: any value gives the same hits
        for ll in ("1", "3", "24"):
            monkeypatch.setenv("TRT_LEAF_LOOP", ll)
            r2 = T.Renderer(s, 0)
            try:
                t2, tri2, uv2 = r2.trace_closest(o3, d3)
                assert np.array_equal(tri2, tri0[-len(o3):]) and np.array_equal(t2, t0[-len(o3):]) and np.array_equal(uv2, uv0[-len(o3):]), ll
                assert_same_image(r2.render(p)[0], ref, f"{name} {tree} leaf loop {ll}")
            finally:
                r2.close()
        monkeypatch.delenv("TRT_LEAF_LOOP")
    finally:
        r.close()
        s.close()


@pytest.mark.parametrize("name,w,h,spp", [("staircase", 96, 54, 8), ("veach-mis", 96, 54, 8), ("back", 64, 64, 8)])
def test_specular_ks_flag_matches_oracle(name, w, h, spp, renderer_factory):
    """TRT_FLAG_SPECULAR_KS, alone and with the other estimator flags: bit-identical to the oracle that takes the same switch."""
    s = get_scene(name, w, h)
    r = renderer_factory(s)
    for flags in (T.TRT_FLAG_SPECULAR_KS, T.TRT_FLAG_SPECULAR_KS | T.TRT_FLAG_FIXED_NEE | T.TRT_FLAG_RAY_OFFSET, T.TRT_FLAG_SPECULAR_KS | T.TRT_FLAG_OVERLAP | T.TRT_FLAG_FIXED_PIXELS):
        p = T.make_params(w, h, spp, 99, flags=flags)
        img, st = r.render(p)
        ref, ost = O.render(s.flat, p)
        assert_same_image(img, ref, f"{name} flags {flags}")
        assert st.rays == ost.rays


def test_redo_path_is_counted_and_rare():
    """trt_stats.redo_rays: how many rays failed the check made when a result is stored and went through k_trace_fix.  On the shipped
    scenes that is a handful per ten million (with the bare `t < entry` rule of round 2 an unpadded tree sent a third of its rays there)."""
    s = get_scene("staircase", 256, 144)
    r = T.Renderer(s, 0)
    img, st = r.render(T.make_params(256, 144, 32, 99))
    r.close()
    assert st.redo_rays * 100000 <= st.rays, (st.redo_rays, st.rays)


def test_unpadded_leaf_boxes_on_the_gpu():
    """The tolerant leaf-box rule end to end: `back` with the 0.001 pad taken off its leaf boxes (triangles ON the faces of their leaves'
    boxes) traced per lane — same hits as the oracle, which loses none against the reference's own arithmetic
    (tests/test_literal_tolerance.py), and the redo path stays idle."""
    import test_literal_tolerance as TL
    s = T.Scene.named("back", 64, 64, leaf_num=2)
    assert TL._unpad_leaf_boxes(s) > 0
    lo, hi = raygen.scene_bounds(s)
    org, d = raygen.random_rays(200000, lo + 1.0, hi - 1.0, seed=12)
    t0, tri0, uv0 = O.trace(s.flat, org, d)
    tl, tril, _ = O.trace_literal(s.flat, org, d)
    assert int(((tri0 < 0) & (tril >= 0)).sum()) <= 4
    os.environ["TRT_TRACE_IMPL"] = "3"
    try:
        r = T.Renderer(s, 0)
        t1, tri1, uv1, st = r.trace_closest(org, d, want_stats=True)
        r.close()
    finally:
        del os.environ["TRT_TRACE_IMPL"]
    assert np.array_equal(tri0, tri1) and np.array_equal(t0, t1) and np.array_equal(uv0, uv1)
    assert st.redo_rays <= 20, st.redo_rays
    s.close()


@pytest.mark.parametrize("name,w,h,spp", [("back", 128, 128, 16), ("veach-mis", 96, 54, 8), ("staircase", 64, 36, 8)])
def test_ray_offset_flag_matches_oracle(name, w, h, spp, renderer_factory):
    """TRT_FLAG_RAY_OFFSET (opt-out of Q6) through the wavefront kernels and k_tail, alone and with the other opt-outs."""
    s = get_scene(name, w, h)
    r = renderer_factory(s)
    for flags in (T.TRT_FLAG_RAY_OFFSET, T.TRT_FLAG_RAY_OFFSET | T.TRT_FLAG_FIXED_NEE | T.TRT_FLAG_FIXED_PIXELS | T.TRT_FLAG_OVERLAP):
        p = T.make_params(w, h, spp, 123, flags=flags)
        img, st = r.render(p)
        ref, ost = O.render(s.flat, p)
        assert_same_image(img, ref, f"{name} ray offset flags={flags}")
        assert (st.rays_camera, st.rays_shadow, st.rays_indirect, st.shaded_hits) == (ost.rays_camera, ost.rays_shadow, ost.rays_indirect, ost.shaded_hits)


@pytest.mark.gpu
def test_exact_sqrt_sequences_equal_ieee_on_every_input():
    """include/trt_exact.h: the short v_rsq-based sequences the kernels use for sqrtf(x) and 1.0f / sqrtf(x) return the bits of
    hipcc's correctly rounded expansions for ALL 2^32 binary32 inputs (a unary function: the check is exhaustive, i.e. a proof),
    and the three-instruction binary64 division of the pixel grid returns the bits of a / b for all 1.1e12 operand pairs the grid
    can form.  The tool also checks that it really visited every input."""
    import subprocess
    exe = os.path.join(T.REPO_ROOT, "tools", "exact_unary_check")
    assert os.path.exists(exe), "tools/exact_unary_check is not built (make exactcheck)"
    r = subprocess.run([exe, "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if "inputs differ" in l]
    assert len(lines) == 4 and all(" 0 of 4294967296 inputs differ" in l for l in lines), r.stdout
    # trt_div_by (the pixel grid's binary64 divisions by W - 1, H - 1, W, H): every operand pair the grid can form
    div = [l for l in r.stdout.splitlines() if "operand pairs differ" in l]
    assert len(div) == 1 and " 0 of 1103806660608 operand pairs differ" in div[0], r.stdout


def test_hit_in_front_of_its_leaf_box_does_not_count_on_the_gpu(monkeypatch):
    """The configuration tools/fuzz_parity.py found (see tests/test_hostsim_parity.py): a grazing hit in front of the box of its own
    leaf.  Both node kinds of the traversal must agree with the oracle, in the occlusion-test mode (where it was found) and in parity mode."""
    sc = get_scene("veach-mis", 320, 180)
    for flags in (T.TRT_FLAG_FIXED_NEE, 0):
        p = T.make_params(320, 180, 33, 2073828938, tile=(132, 93, 156, 104), rows=(1, 3, 1), flags=flags)
        ref, ost = O.render(sc.flat, p)
        for nk in NODE_KINDS:
            monkeypatch.setenv("TRT_NODE_KIND", nk)
            img, st = T.Renderer(sc, 0).render(p)
            assert_same_image(img, ref, f"flags {flags} node kind {nk}")
            assert (st.rays_camera, st.rays_shadow, st.rays_indirect) == (ost.rays_camera, ost.rays_shadow, ost.rays_indirect)


@pytest.mark.parametrize("name", ["back", "veach-mis", "staircase"])
def test_trace_grazing_rays_on_every_node_kind(name, monkeypatch):
    """Rays within 1e-5 .. 1e-2 rad of a triangle's plane (raygen.grazing_rays): the closest hits through either node kind equal the unculled
    oracle's bit for bit — the case the leaf-box rule exists for; k_trace_fix handles the rays whose result fails the check."""
    s = get_scene(name, 64, 64)
    org, dirs = raygen.grazing_rays(s.flat, 150000)
    t0, tri0, uv0 = O.trace(s.flat, org, dirs)
    for nk in NODE_KINDS:
        monkeypatch.setenv("TRT_NODE_KIND", nk)
        t1, tri1, uv1 = T.Renderer(s, 0).trace_closest(org, dirs)
        assert np.array_equal(tri0, tri1) and np.array_equal(t0, t1) and np.array_equal(uv0, uv1), (name, nk)


@pytest.mark.parametrize("name,leaf,nk,boxes", [("back", 8, "0", False), ("veach-mis", 2, "1", False), ("veach-mis", 8, "1", False), ("staircase", 2, "0", False),
                                                ("staircase", 2, "1", False), ("veach-mis", 2, "1", True), ("staircase", 8, "1", True)])
def test_non_finite_geometry_is_absorbed_identically(name, leaf, nk, boxes, monkeypatch):
    """The kernels on NaN / inf / 1e38 / denormal coordinates and normals (scene_util.poison_geometry, written into the caller's arrays before trt_create): every
    launch ends (the traversal is a walk over a validated topology, whatever the numbers say), and the image is the oracle's bit for bit."""
    monkeypatch.setenv("TRT_NODE_KIND", nk)
    s = T.Scene.named(name, 96, 54, leaf_num=leaf)
    SU.poison_geometry(s, boxes=boxes)  # boxes: +-inf / +-1e38 in the tree's boxes too (the handle then keeps the exact 4-wide nodes)
    p = T.make_params(96, 54, 4, 77)
    ref, ost = O.render(s.flat, p)
    r = T.Renderer(s, 0)
    img, st = r.render(p)
    r.close()
    assert np.isfinite(ref).all()
    assert_same_image(img, ref, f"{name} leaf {leaf} node kind {nk}, poisoned")
    assert st.rays == ost.rays
    s.close()


@pytest.mark.parametrize("name,impl,nk", [("back", "0", "0"), ("back", "3", "1"), ("veach-mis", "3", "0"), ("veach-mis", "3", "1"), ("staircase", "3", "1")])
def test_non_finite_rays_find_what_the_oracle_finds(name, impl, nk, monkeypatch):
    """trt_trace_closest on NaN / inf / 1e38 / denormal origins and directions and on zero directions (raygen.non_finite_rays): every launch ends, and the
    answer is the oracle's — same triangle, same bits of t and (u, v) — on the wave-uniform walk of a tiny tree and on both node kinds of the persistent kernels."""
    monkeypatch.setenv("TRT_TRACE_IMPL", impl)
    monkeypatch.setenv("TRT_NODE_KIND", nk)
    s = get_scene(name, 64, 36)
    org, dirs = raygen.non_finite_rays(s, 200000)
    t0, tri0, uv0 = O.trace(s.flat, org, dirs)
    r = T.Renderer(s, 0)
    t1, tri1, uv1 = r.trace_closest(org, dirs)
    r.close()
    assert np.array_equal(tri0, tri1) and np.array_equal(t0.view(np.uint32), t1.view(np.uint32)) and np.array_equal(uv0.view(np.uint32), uv1.view(np.uint32))


def test_hostile_table_values_render_like_the_oracle():
    """The kernels on 36 scenes with hostile numbers in their tables (scene_util.poison_tables): every launch ends, image bits and ray counts are the oracle's."""
    for trial in range(36):
        rng = np.random.default_rng(500 + trial)
        name = ["back", "veach-mis", "staircase"][trial % 3]
        s = T.Scene.named(name, 64, 36)
        what = SU.poison_tables(s, rng)
        p = T.make_params(64, 36, 4, 2000 + trial, flags=int(rng.choice([0, T.TRT_FLAG_FIXED_NEE, T.TRT_FLAG_RAY_OFFSET, T.TRT_FLAG_SPECULAR_KS, T.TRT_FLAG_OVERLAP])))
        ref, ost = O.render(s.flat, p)
        r = T.Renderer(s, 0)
        img, st = r.render(p)
        r.close()
        assert np.array_equal(ref.view(np.uint32), img.view(np.uint32)), (name, what, p.flags)
        assert st.rays == ost.rays, (name, what, p.flags)
        s.close()


@pytest.mark.parametrize("impl,nk", [("0", "0"), ("3", "0"), ("3", "1")])
def test_zero_direction_component_on_a_box_plane_takes_the_literal_slab_test(impl, nk, tmp_path, monkeypatch):
    """The kernels on test_hostsim_parity's vertex-eye scene: the wave-uniform walk switches the rays with a zero direction component to the literal slab test, the
    per-lane drivers hand them to k_trace_fix (the caller's BVH2, literal test, no culling); hits and image are the oracle's."""
    from test_hostsim_parity import _vertex_eye_scene
    monkeypatch.setenv("TRT_TRACE_IMPL", impl)
    monkeypatch.setenv("TRT_NODE_KIND", nk)
    s = _vertex_eye_scene(tmp_path)
    f = s.flat.contents
    rng = np.random.default_rng(3)
    rays = [O.camera_ray(f.camera, 24, 5, 1, int(rng.integers(17, 22)), float(np.float32(rng.random())), float(np.float32(rng.random()))) for _ in range(20000)]
    org, dirs = np.array([r[0] for r in rays], np.float32), np.array([r[1] for r in rays], np.float32)
    t0, tri0, uv0 = O.trace(s.flat, org, dirs)
    assert (tri0[(dirs == 0).any(1)] >= 0).sum() > 50
    r = T.Renderer(s, 0)
    t1, tri1, uv1 = r.trace_closest(org, dirs)
    assert np.array_equal(tri0, tri1) and np.array_equal(t0, t1) and np.array_equal(uv0, uv1)
    for flags in (0, T.TRT_FLAG_FIXED_NEE, T.TRT_FLAG_OVERLAP):
        p = T.make_params(24, 5, 64, 4106463245, max_depth=3, flags=flags)
        img, st = r.render(p)
        ref, ost = O.render(s.flat, p)
        assert_same_image(img, ref, f"vertex eye, impl {impl}, node kind {nk}, flags {flags}")
        assert st.rays == ost.rays
    r.close()
    s.close()


def test_random_scenes_render_like_the_oracle():
    """tools/fuzz_scenes.py --gpu for 40 s: random scenes through the loaders, the builders and the kernels (default handle and both node kinds per lane)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(GOLD), "..", "tools", "fuzz_scenes.py"), "--gpu", "--seconds", "40", "--seed", "9"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "all bit-identical" in r.stdout


def test_a_queue_full_of_axis_aligned_rays(renderer_factory):
    """8 M rays of which half have a zero direction component, on the Cornell box (wave-uniform walk): every wave parks more such rays than its list holds and
    goes over its share a second time with the literal slab test; 300 k of them on staircase (per-lane kernels: all of those go through k_trace_fix).  Same hits."""
    s = get_scene("back", 64, 64)
    org, dirs = raygen.adversarial_rays(s, 8000000)
    t0, tri0, uv0 = O.trace(s.flat, org, dirs)
    t1, tri1, uv1, st = renderer_factory(s).trace_closest(org, dirs, want_stats=True)
    assert np.array_equal(tri0, tri1) and np.array_equal(t0, t1) and np.array_equal(uv0, uv1)
    assert st.redo_rays > 3000000
    s2 = get_scene("staircase", 64, 36)
    org, dirs = raygen.adversarial_rays(s2, 300000)
    t0, tri0, uv0 = O.trace(s2.flat, org, dirs)
    t1, tri1, uv1, st = renderer_factory(s2).trace_closest(org, dirs, want_stats=True)
    assert np.array_equal(tri0, tri1) and np.array_equal(t0, t1) and np.array_equal(uv0, uv1)
    assert st.redo_rays > 100000
