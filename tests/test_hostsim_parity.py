"""CPU check of the DEVICE-SIDE code: tinyraytracing_amd/csrc/trt_path.h (ordered, culled stack
traversal; shadeBegin / lightSample / shadeNext as the wavefront kernels call them) compiled with
g++ (tests/hostsim) must reproduce the oracle — which traverses recursively, unordered and unculled
like bvh.cpp:146-175 — bit for bit, and the committed golden fixtures too."""
import os

import numpy as np
import pytest

import hostsim_lib as H
import oracle_lib as O
import raygen
import tinyraytracing_amd as T
from conftest import get_scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["back", "veach-mis", "staircase"])
def test_oracle_reproduces_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    w, h, spp, seed = int(g["width"]), int(g["height"]), int(g["spp"]), int(g["seed"])
    s = get_scene(name, w, h)
    img, st = O.render(s.flat, T.make_params(w, h, spp, seed))
    assert np.array_equal(img, g["image"])
    assert [st.rays_camera, st.rays_shadow, st.rays_indirect] == g["rays"].tolist()
    t, tri, uv = O.trace(s.flat, g["org"], g["dir"])
    assert np.array_equal(t, g["t"]) and np.array_equal(tri, g["tri"]) and np.array_equal(uv, g["uv"])


@pytest.mark.parametrize("name", ["back", "veach-mis", "staircase"])
def test_device_code_on_cpu_matches_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    w, h, spp, seed = int(g["width"]), int(g["height"]), int(g["spp"]), int(g["seed"])
    s = get_scene(name, w, h)
    img, rays = H.render(s.flat, T.make_params(w, h, spp, seed))
    assert np.array_equal(img, g["image"])
    assert rays == g["rays"].tolist()
    t, tri, uv, cnt = H.trace(s.flat, g["org"], g["dir"])
    assert np.array_equal(t, g["t"]) and np.array_equal(tri, g["tri"]) and np.array_equal(uv, g["uv"])


def test_ordered_culled_traversal_visits_less_but_finds_the_same():
    s = get_scene("staircase", 64, 36)
    lo, hi = raygen.scene_bounds(s)
    org, dirs = raygen.random_rays(20000, lo, hi, seed=21)
    t0, tri0, uv0, st = O.trace(s.flat, org, dirs, want_stats=True)
    t1, tri1, uv1, cnt = H.trace(s.flat, org, dirs)
    assert np.array_equal(t0, t1) and np.array_equal(tri0, tri1) and np.array_equal(uv0, uv1)
    assert cnt[0] < st.inner_visits[0] and cnt[1] < st.tri_tests[0]


@pytest.mark.parametrize("name", ["back", "veach-mis", "staircase"])
def test_wide_tree_on_degenerate_rays(name):
    """The 4-wide collapse (trt_wide.h) drops intermediate boxes; the closest hit must not depend on that
    even where the slab test meets 0 * inf (zero direction components, origins on box planes)."""
    s = get_scene(name, 64, 36)
    org, dirs = raygen.adversarial_rays(s, 30000)
    t0, tri0, uv0 = O.trace(s.flat, org, dirs)
    t1, tri1, uv1, _ = H.trace(s.flat, org, dirs)
    assert np.array_equal(tri0, tri1) and np.array_equal(t0, t1) and np.array_equal(uv0, uv1)
    assert (tri0 >= 0).sum() > 1000


def test_foreign_bvh_with_boxes_that_do_not_nest():
    """collapseBvh (trt_wide.h) may only drop an intermediate box that contains its children's boxes; where a
    foreign tree breaks that, the node must stay a node of its own and the hits must still equal the oracle's
    (which enters a subtree iff the ray passes its stored box, bvh.cpp:162-166)."""
    import scene_util
    s = T.Scene.named("staircase", 64, 36)
    assert scene_util.shrink_some_boxes(s, 400) == 400
    # 2 M rays, among them the seven of seeds 101 / 102 that a traversal culling by distance got wrong on this tree (round 4: a box that does not
    # contain what lies below it can be ENTERED after a hit below it — nothing may be skipped for lying "beyond the best hit"; trt_wide.h boxesNested)
    lo, hi = np.array([-8, -1, -8], np.float32), np.array([8, 8, 8], np.float32)
    for seed in (101, 102):
        org, dirs = raygen.random_rays(1000000, lo, hi, seed=seed)
        t0, tri0, uv0 = O.trace(s.flat, org, dirs)
        t1, tri1, uv1, _ = H.trace(s.flat, org, dirs)
        assert np.array_equal(tri0, tri1) and np.array_equal(t0, t1) and np.array_equal(uv0, uv1)
    # and the shrunken boxes do change what is found (the test bites)
    t2, tri2, _ = O.trace(get_scene("staircase", 64, 36).flat, org, dirs)
    assert (tri0 != tri2).sum() > 50


@pytest.mark.parametrize("name", ["back", "veach-mis", "staircase"])
def test_fixed_nee_mode_device_code_equals_oracle(name):
    """TRT_FLAG_FIXED_NEE: own-area CDF draw, uniform light points, occlusion test (trt_path.h) == the oracle's."""
    s = get_scene(name, 64, 36)
    p = T.make_params(64, 36, 8, 123, flags=T.TRT_FLAG_FIXED_NEE)
    a, sa = O.render(s.flat, p)
    b, rays = H.render(s.flat, p)
    assert np.array_equal(a, b) and rays == [sa.rays_camera, sa.rays_shadow, sa.rays_indirect]
    q, _ = O.render(s.flat, T.make_params(64, 36, 8, 123))
    assert not np.array_equal(a, q)


def test_fixed_pixels_mode_device_code_equals_oracle_and_centres_the_image():
    """TRT_FLAG_FIXED_PIXELS (opt-out of Q1/Q2): same bits as the oracle; and the image is no longer shifted by a row:
    a scene that is mirror-symmetric top/bottom about the optical axis renders symmetric hit masks."""
    s = get_scene("back", 64, 64)
    fl = T.TRT_FLAG_FIXED_PIXELS | T.TRT_FLAG_FIXED_NEE
    p = T.make_params(64, 64, 8, 9, flags=fl)
    a, sa = O.render(s.flat, p)
    b, rays = H.render(s.flat, p)
    assert np.array_equal(a, b) and rays == [sa.rays_camera, sa.rays_shadow, sa.rays_indirect]
    # the quirk grid samples t = (H - i)/(H - 1) > 1 in the top row (above the image plane): with the fixed grid
    # every sample lies inside [0,1)^2, so the camera ray of pixel (i, j) and of (H-1-i, j) mirror each other
    cam = s.flat.contents.camera
    o0, d0 = O.camera_ray(cam, 64, 64, 0, 10, 0.5, 0.5, fixed=True)
    o1, d1 = O.camera_ray(cam, 64, 64, 63, 10, 0.5, 0.5, fixed=True)
    up = np.array(cam.vertical[:], np.float64); up /= np.linalg.norm(up)
    assert abs(np.dot(d0, up) + np.dot(d1, up)) < 1e-6
    q0, e0 = O.camera_ray(cam, 64, 64, 0, 10, 0.5, 0.5)
    q1, e1 = O.camera_ray(cam, 64, 64, 63, 10, 0.5, 0.5)
    assert abs(np.dot(e0, up) + np.dot(e1, up)) > 1e-3


def test_device_code_on_synthetic_soup_and_tiles():
    s = T.Scene.named("soup", 48, 27, n=20000)
    p = T.make_params(48, 27, 4, T.SEED_SOUP)
    a, sa = O.render(s.flat, p)
    b, rays = H.render(s.flat, p)
    assert np.array_equal(a, b) and rays == [sa.rays_camera, sa.rays_shadow, sa.rays_indirect]
    # tiles and row interleave address the same (pixel, sample) streams
    pt = T.make_params(48, 27, 4, T.SEED_SOUP, tile=(8, 3, 40, 20), rows=(2, 3, 1))
    c, _ = H.render(s.flat, pt)
    assert np.array_equal(c, a[T.rows_selected(pt)][:, 8:40])


def test_max_depth_and_one_spp():
    s = get_scene("back", 32, 32)
    for md in (1, 2, 5):
        p = T.make_params(32, 32, 1, 5, max_depth=md)
        a, sa = O.render(s.flat, p)
        b, rays = H.render(s.flat, p)
        assert np.array_equal(a, b)
        assert sa.max_bounces <= md - 1 and rays[2] == sa.rays_indirect
    assert O.render(s.flat, T.make_params(32, 32, 1, 5, max_depth=1))[1].rays_indirect == 0


@pytest.mark.parametrize("name", ["veach-mis", "staircase"])
def test_compressed_nodes_give_the_same_hits_and_image(name):
    """Node kind 1 (trt_oct.h): 8-wide nodes with quantised conservative boxes, tested in the node's frame with margins, reach a
    superset of the reference's leaves; with the result checked against the exact box of its leaf (and the exact form behind
    that) hits, barycentrics, image and ray counts are identical to the oracle's — also on degenerate rays (zero direction
    components, origins on box planes: those axes drop out of the node test and the traversal just visits more)."""
    s = get_scene(name, 64, 36)
    assert H.compressible(s.flat)
    old = H.set_node_kind(1)
    try:
        org, dirs = raygen.adversarial_rays(s, 20000)
        lo, hi = raygen.scene_bounds(s)
        o2, d2 = raygen.random_rays(20000, lo - 5, hi + 5, seed=4)
        o3, d3 = raygen.grazing_rays(s.flat, 20000, seed=9)
        org, dirs = np.vstack([org, o2, o3]), np.vstack([dirs, d2, d3])
        t0, tri0, uv0 = O.trace(s.flat, org, dirs)
        t1, tri1, uv1, _ = H.trace(s.flat, org, dirs)
        assert np.array_equal(t0, t1) and np.array_equal(tri0, tri1) and np.array_equal(uv0, uv1)
        _, _, _, cnt1 = H.trace(s.flat, o2, d2)
        H.set_node_kind(0)
        _, _, _, cnt0 = H.trace(s.flat, o2, d2)
        assert cnt1[0] <= cnt0[0]           # eight children per visit: fewer node visits than the 4-wide tree on ordinary rays
        assert cnt1[1] <= 1.25 * cnt0[1]    # looser boxes and no distance sort: some more triangle tests
        H.set_node_kind(1)
        p = T.make_params(64, 36, 4, 11)
        img, rays = H.render(s.flat, p)
        ref, st = O.render(s.flat, p)
        assert np.array_equal(img, ref) and rays == [st.rays_camera, st.rays_shadow, st.rays_indirect]
        pf = T.make_params(64, 36, 2, 12, flags=T.TRT_FLAG_FIXED_NEE | T.TRT_FLAG_RAY_OFFSET)  # the occlusion-test shadow rays through the oct tree
        assert np.array_equal(H.render(s.flat, pf)[0], O.render(s.flat, pf)[0])
    finally:
        H.set_node_kind(old)


@pytest.mark.parametrize("name", ["staircase", "veach-mis"])
def test_axis_aligned_rays_stay_cheap_on_the_oct_nodes(name):
    """A direction component that is exactly zero (camera rays of an axis-aligned camera: d.x is a difference of numbers near the eye's
    coordinate and comes out as exactly 0 for one pixel column in ~40 000) must not drop out of the node test: the first version of
    trt_oct.h left such an axis unconstrained — exact, but the ray then visited every node of a slab of the scene (1 500 nodes and 4 600
    triangles on the 2 M-triangle mesh: one GPU lane busy for milliseconds, 100x on the whole kernel).  With the axis tested as the
    reference tests it (origin between the planes or not) the longest traversal is of the order of the exact nodes' longest one."""
    s = get_scene(name, 160, 90)
    lo, hi = raygen.scene_bounds(s)
    rng = np.random.default_rng(3)
    n = 30000
    org = (rng.random((n, 3)) * (hi - lo) + lo).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[np.arange(n), rng.integers(0, 3, n)] = 0.0       # one component exactly zero ...
    two = rng.random(n) < 0.2
    d[two, rng.integers(0, 3, int(two.sum()))] = 0.0   # ... or two
    d[np.abs(d).sum(1) == 0] = (0.0, 0.0, 1.0)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    v0, t0 = H.trace_counts(s.flat, 0, org, d)
    v1, t1 = H.trace_counts(s.flat, 1, org, d)
    assert v1.mean() <= v0.mean()                      # still fewer node visits than the 4-wide tree
    assert int(v1.max()) <= 2 * int(v0.max()) + 16, (int(v0.max()), int(v1.max()))
    assert int(t1.max()) <= 3 * int(t0.max()) + 48, (int(t0.max()), int(t1.max()))
    assert H.oct_fallbacks(s.flat, org, d) <= 3        # and the results of such rays pass the check like any other


def test_foreign_tree_is_not_compressed():
    """A tree whose boxes are not nested (a child's box sticks out of its parent's) cannot use the leaf-box argument."""
    import scene_util as SU
    s = T.Scene.named("staircase", 64, 36)
    assert H.compressible(s.flat)
    assert SU.shrink_some_boxes(s, 50) > 0
    assert not H.compressible(s.flat)
    s.close()


@pytest.mark.parametrize("name", ["back", "veach-mis", "staircase"])
def test_ray_offset_flag_device_code_matches_oracle(name):
    """TRT_FLAG_RAY_OFFSET (opt-out of Q6): the device-side path functions and the oracle offset every shadow and continuation
    ray by the same eps * Ng, bit for bit; combined with the other opt-outs as well."""
    s = get_scene(name, 48, 27)
    for flags in (T.TRT_FLAG_RAY_OFFSET, T.TRT_FLAG_RAY_OFFSET | T.TRT_FLAG_FIXED_NEE | T.TRT_FLAG_FIXED_PIXELS):
        p = T.make_params(48, 27, 4, 21, flags=flags)
        img, rays = H.render(s.flat, p)
        ref, st = O.render(s.flat, p)
        assert np.array_equal(img, ref) and rays == [st.rays_camera, st.rays_shadow, st.rays_indirect]
        rec, _ = O.render(s.flat, p, mode=O.MODE_RECURSIVE)
        assert np.allclose(rec, ref, rtol=2e-4, atol=1e-6)


def test_ray_offset_removes_the_self_hits_of_q6():
    """What the flag is for: on `back` at 1024 x 1024 the far cube face is ~1000 units from the eye, the hit point's rounding
    error beats t < 0.0005 for grazing directions and a few per cent of the rays leaving it hit it again at once (Q6) — visible as
    extra path vertices.  With the offset the image keeps its mean (the self-hits mostly swallowed NEE light and added a bounce) to
    a few per cent and the vertex count per camera ray drops."""
    s = get_scene("back", 256, 256)
    p0 = T.make_params(256, 256, 16, T.SEED_BACK)
    p1 = T.make_params(256, 256, 16, T.SEED_BACK, flags=T.TRT_FLAG_RAY_OFFSET)
    a, sa = O.render(s.flat, p0)
    b, sb = O.render(s.flat, p1)
    assert not np.array_equal(a, b)
    assert abs(float(b.mean()) / float(a.mean()) - 1.0) < 0.05
    assert sb.shaded_hits < sa.shaded_hits  # fewer vertices: the immediate re-hits are gone


def test_magic_number_division_is_exact():
    """divMagic() (trt_path.h: path id -> sample, row, column with host-formed magic numbers) equals the integer division
    for divisors of every size, on the edges of the 32-bit range and on a few million random numerators each."""
    import ctypes as C
    lib = H.lib()
    lib.hostsim_div_magic_mismatches.restype = C.c_uint64
    lib.hostsim_div_magic_mismatches.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64]
    rng = np.random.default_rng(7)
    divisors = [1, 2, 3, 5, 7, 64, 1000, 1080, 1920, 65535, 65536, 65537, 1920 * 1080, 3840 * 2160, 2**31 - 1, 2**31, 2**32 - 1]
    divisors += [int(x) for x in rng.integers(1, 2**32, 40, dtype=np.uint64)]
    for d in divisors:
        n = rng.integers(0, 2**32, 2_000_000, dtype=np.uint64).astype(np.uint32)
        edges = np.array([0, 1, d - 1, d, (d + 1) & 0xFFFFFFFF, 2**32 - 1, 2**32 - 2, 2**31, (2**32 // d) * d - 1 & 0xFFFFFFFF, ((2**32 // d) * d) & 0xFFFFFFFF], dtype=np.uint64).astype(np.uint32)
        mult = (np.arange(1, 200001, dtype=np.uint64) * d)  # multiples of d and their predecessors: where a quotient steps
        mult = mult[mult < 2**32]
        n = np.ascontiguousarray(np.concatenate([n, edges, mult.astype(np.uint32), (mult - 1).astype(np.uint32)]))
        assert lib.hostsim_div_magic_mismatches(d, n.ctypes.data, n.size) == 0, d


def test_hit_in_front_of_its_leaf_box_does_not_count():
    """A ray within ~1e-4 rad of a triangle's plane (veach-mis, a facet of the small sphere light, found by tools/fuzz_parity.py): the
    Moller-Trumbore distance comes out 0.006 in front of the triangle, outside the box of its own leaf, while the barycentrics say
    'inside'.  Unculled (oracle) and culled (kernels) traversal used to disagree on that shadow ray; with the rule 'a hit in
    front of its leaf's box does not count' (leafEntry(), trt_path.h; leafScan(), oracle.cpp) they agree, bit for bit."""
    sc = get_scene("veach-mis", 320, 180)
    p = T.make_params(320, 180, 33, 2073828938, tile=(132, 93, 156, 104), rows=(1, 3, 1), flags=T.TRT_FLAG_FIXED_NEE)
    ref, ost = O.render(sc.flat, p)
    img, rays = H.render(sc.flat, p)
    assert np.array_equal(img, ref)
    assert tuple(int(x) for x in rays) == (ost.rays_camera, ost.rays_shadow, ost.rays_indirect)
    # the ray itself: in closest-hit mode both sides see the facet (2138 in BVH order) 0.006 in front of its leaf's box ...
    o = np.array([[3.65241623, 2.63776708, 1.35494876]], np.float32)
    d = np.array([[-0.660412669, 0.707522273, 0.251529932]], np.float32)
    t_h, tri_h, _, _ = H.trace(sc.flat, o, d)
    t_o, tri_o, _ = O.trace(sc.flat, o, d)
    assert tri_h[0] == tri_o[0] and t_h[0] == t_o[0]
    assert tri_h[0] != 2138  # ... and neither accepts it any more


@pytest.mark.parametrize("name", ["veach-mis", "staircase"])
def test_grazing_rays_culled_equals_unculled(name):
    """150 000 rays within 1e-5 .. 1e-2 rad of a triangle's plane: the kernels' ordered, culled traversal (compiled for the CPU) returns
    the unculled oracle's hit bit for bit.  (Without the leaf-box rule of DESIGN.md §2 this set gives a few dozen mismatches on staircase.)"""
    sc = get_scene(name, 64, 64)
    o, d = raygen.grazing_rays(sc.flat, 150000)
    th, trih, uvh, _ = H.trace(sc.flat, o, d)
    to, trio, uvo = O.trace(sc.flat, o, d)
    assert np.array_equal(trih, trio) and np.array_equal(th, to) and np.array_equal(uvh, uvo)


@pytest.mark.parametrize("name,kw", [("veach-mis", {}), ("staircase", {}), ("soup", {"n": 60000}), ("blob", {"n": 150000})])
def test_the_collapses_do_not_depend_on_the_number_of_host_threads(name, kw):
    """trt_create collapses the caller's tree with several host threads (trt_wide.h `par`: subtrees of a cut of the binary tree for the
    dynamic programmes, blocks placed where the sequential walk puts them for the 4-wide nodes, level by level for the 8-wide ones): the
    4-wide trees of both collapses, the 8-wide nodes, their triangle records and the leaf boxes have the same bytes for 1, 2, 3 and 7 threads."""
    s = T.Scene.named(name, 64, 64, **kw)
    ref = H.tree_hashes(s.flat, 1)
    assert ref[2] != 0 and ref[6] > 1, "the scene qualifies for the 8-wide nodes"
    for threads in (2, 3, 7):
        assert H.tree_hashes(s.flat, threads) == ref, threads
    s.close()


def test_step_tape_agrees_with_the_visit_counters():
    """tools/pool_sim.py feeds on the per-ray tape of steps of the oct traversal (hostsim_oct_step_tape): its node steps are the traversal's node
    visits, its leaf steps add up to the traversal's triangle tests, two at most per step (the wave driver's leaf loop)."""
    import ctypes as C
    s = T.Scene.named("staircase", 64, 36)
    lo, hi = raygen.scene_bounds(s)
    org, dirs = raygen.random_rays(20000, lo, hi, seed=8)
    v, t = H.trace_counts(s.flat, 1, org, dirs)
    lib = H.lib()
    lib.hostsim_oct_step_tape.argtypes = [C.c_void_p, C.c_uint64, H.fp, H.fp, H.fp, C.c_int, C.c_uint32, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)]
    cap = 255
    tape = np.zeros((len(org), cap), np.uint8)
    ln = np.zeros(len(org), np.uint32)
    o = np.ascontiguousarray(org, np.float32)
    d = np.ascontiguousarray(dirs, np.float32)
    assert lib.hostsim_oct_step_tape(C.cast(s.flat, C.c_void_p), len(o), o.ctypes.data_as(H.fp), d.ctypes.data_as(H.fp), None, -1, cap, tape.ctypes.data_as(C.POINTER(C.c_uint8)),
                                     ln.ctypes.data_as(C.POINTER(C.c_uint32))) == 0
    assert ln.max() <= cap
    valid = np.arange(cap)[None, :] < ln[:, None]
    assert np.array_equal(((tape == 0) & valid).sum(1), v)
    assert np.array_equal((tape * valid).sum(1), t) and tape.max() <= 2
    s.close()


def _large_leaf_scene(name, tree, w, h):
    import scene_util as SU
    if tree == "reference-8":
        return SU.load_with_reference_tree(name, w, h, 8)
    return T.Scene.named(name, w, h, leaf_num=int(tree.split("-")[1]))


@pytest.mark.parametrize("name,tree", [("veach-mis", "reference-8"), ("staircase", "reference-8"), ("staircase", "sah-8"), ("veach-mis", "sah-15"), ("staircase", "sah-5")])
def test_trees_with_large_leaves_walk_the_oct_nodes(name, tree):
    """The reference builds its tree with up to 8 triangles per leaf (main.cpp:76).  Such a tree — the reference builder's own, restated
    in the oracle, or this repository's with leaf_num 5 / 8 / 15 — gets the 8-wide nodes too: a leaf of more than 3 triangles is laid out
    as several slots that all carry the leaf's own box (trt_oct_build.h), the tie rule and the check of the result keep speaking of the
    caller's leaf.  Hits, barycentrics, image and ray counts equal the oracle's on the caller's tree, grazing and degenerate rays included."""
    s = _large_leaf_scene(name, tree, 64, 36)
    old = H.set_node_kind(1)
    try:
        assert H.compressible(s.flat)
        info = H.oct_info(s.flat)
        assert info is not None and info[2] == s.info["n_triangles"]
        org, dirs = raygen.adversarial_rays(s, 15000)
        lo, hi = raygen.scene_bounds(s)
        o2, d2 = raygen.random_rays(15000, lo - 5, hi + 5, seed=4)
        o3, d3 = raygen.grazing_rays(s.flat, 15000, seed=9)
        org, dirs = np.vstack([org, o2, o3]), np.vstack([dirs, d2, d3])
        t0, tri0, uv0 = O.trace(s.flat, org, dirs)
        t1, tri1, uv1, _ = H.trace(s.flat, org, dirs)
        assert np.array_equal(t0, t1) and np.array_equal(tri0, tri1) and np.array_equal(uv0, uv1)
        p = T.make_params(64, 36, 3, 11)
        img, rays = H.render(s.flat, p)
        ref, st = O.render(s.flat, p)
        assert np.array_equal(img, ref) and rays == [st.rays_camera, st.rays_shadow, st.rays_indirect]
        # the same triangles are tested as on the exact 4-wide nodes of the same tree, give or take the culling order
        v1, n1 = H.trace_counts(s.flat, 1, o2, d2)
        v0, n0 = H.trace_counts(s.flat, 0, o2, d2)
        assert v1.sum() < v0.sum() and n1.sum() <= 1.3 * n0.sum()
    finally:
        H.set_node_kind(old)
        s.close()


def test_equal_distance_hits_inside_one_large_leaf(tmp_path):
    """bvh.cpp:219 inside a leaf (the first of several equally distant triangles, unless a later one is emissive: then the last
    emissive one) and bvh.cpp:168-172 between leaves, on a leaf of 8 coincident triangles that the oct nodes hold as three slots
    entered in octant order: every stacking order of lamp and white quads gives the oracle's triangle, from both sides."""
    import scene_util as SU
    for k, mats in enumerate([("white", "lamp", "white", "lamp"), ("lamp", "white", "white", "white"), ("white", "white", "white", "white"),
                              ("lamp", "lamp", "white", "lamp"), ("white", "white", "lamp", "white")]):
        lines, body, vb = ["vt 0 0", "vn 0 0 1"], "", 1
        for m in mats:
            v, f, vb = SU.quad(-1, 1, -1, 1, 0, vb)
            lines += v
            body += f"usemtl {m}\n" + "\n".join(x.format(n=1) for x in f) + "\n"
        # some more geometry around it so that the tree has inner nodes
        for z in (-2.0, 2.5):
            v, f, vb = SU.quad(-3, 3, -3, 3, z, vb)
            lines += v
            body += "usemtl shiny\n" + "\n".join(x.format(n=1) for x in f) + "\n"
        SU.write_scene(tmp_path, f"stack{k}", "\n".join(lines) + "\n" + body, SU.MTL_BASIC, lights=[("lamp", (5, 5, 5))])
        for leaf in (8, 4, 3):
            s = SU.load(tmp_path, f"stack{k}", leaf_num=leaf)
            rng = np.random.default_rng(k)
            n = 400
            org = np.column_stack([rng.uniform(-0.95, 0.95, n), rng.uniform(-0.95, 0.95, n), np.where(rng.random(n) < 0.5, 1.5, -1.5)]).astype(np.float32)
            dirs = np.column_stack([np.zeros(n), np.zeros(n), -np.sign(org[:, 2])]).astype(np.float32)
            t0, tri0, uv0 = O.trace(s.flat, org, dirs)
            assert np.all(t0 == 1.5)
            old = H.set_node_kind(1)
            try:
                assert H.compressible(s.flat)
                t1, tri1, uv1, _ = H.trace(s.flat, org, dirs)
            finally:
                H.set_node_kind(old)
            assert np.array_equal(tri0, tri1) and np.array_equal(t0, t1) and np.array_equal(uv0, uv1), (mats, leaf)
            s.close()


@pytest.mark.parametrize("name", ["staircase", "veach-mis"])
def test_specular_ks_flag_device_code_equals_oracle(name):
    """TRT_FLAG_SPECULAR_KS (the look of the reference's own saved renders: a SPECULAR bounce weighted by Ks instead of the texel Kd): the device-side path functions
    and both oracles apply the same switch.  staircase has the materials that tell the two weights apart (FloorTiles, Metal), veach-mis has Kd = Ks on every glossy
    plate: there the flag must not change a bit."""
    s = get_scene(name, 64, 36)
    p = T.make_params(64, 36, 4, 21, flags=T.TRT_FLAG_SPECULAR_KS)
    ref, st = O.render(s.flat, p)
    img, rays = H.render(s.flat, p)
    assert np.array_equal(img, ref) and rays == [st.rays_camera, st.rays_shadow, st.rays_indirect]
    plain = O.render(s.flat, T.make_params(64, 36, 4, 21))[0]
    assert np.array_equal(O.render(s.flat, p, mode=O.MODE_RECURSIVE)[0].shape, ref.shape)
    if name == "veach-mis":
        assert np.array_equal(ref, plain)
    else:
        assert not np.array_equal(ref, plain) and ref.mean() < plain.mean()
        rec = O.render(s.flat, p, mode=O.MODE_RECURSIVE)[0]   # the literal recursion of shade() takes the same switch
        assert np.allclose(rec, ref, rtol=2e-4, atol=1e-6)
        lit = O.render_literal(s.flat, p)[0]                   # ... and so does the reference's own arithmetic
        assert abs(lit.mean() / ref.mean() - 1.0) < 0.02


@pytest.mark.parametrize("name,leaf,boxes", [("back", 8, False), ("veach-mis", 2, False), ("veach-mis", 8, False), ("veach-mis", 2, True), ("staircase", 8, True)])
def test_non_finite_geometry_is_absorbed_identically(name, leaf, boxes):
    """NaN / inf / 1e38 / denormal coordinates and normals in the caller's arrays (scene_util.poison_geometry): the device code renders the oracle's image bit for
    bit on both node kinds, and the image itself stays finite (a NaN fails every comparison of bvh.cpp:185-207 and pathTracing.cpp:60)."""
    import scene_util as SU
    s = T.Scene.named(name, 48, 48, leaf_num=leaf)
    assert SU.poison_geometry(s, boxes=boxes) >= 14  # boxes: +-inf / +-1e38 in the tree's boxes too (it then keeps the exact 4-wide nodes)
    p = T.make_params(48, 48, 4, 77)
    ref, ost = O.render(s.flat, p)
    assert np.isfinite(ref).all()
    for nk in (0, 1):
        old = H.set_node_kind(nk)
        try:
            img, rays = H.render(s.flat, p)
        finally:
            H.set_node_kind(old)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (name, leaf, nk)
        assert rays == [ost.rays_camera, ost.rays_shadow, ost.rays_indirect]
    s.close()


@pytest.mark.parametrize("name", ["back", "veach-mis", "staircase"])
def test_non_finite_rays_find_what_the_oracle_finds(name):
    """NaN / inf / 1e38 / denormal origins and directions, zero directions (raygen.non_finite_rays): same triangle, same bits of t and (u, v), both node kinds."""
    s = get_scene(name, 64, 36)
    org, dirs = raygen.non_finite_rays(s, 30000)
    t0, tri0, uv0 = O.trace(s.flat, org, dirs)
    assert (tri0 >= 0).sum() > 3000
    for nk in (0, 1):
        old = H.set_node_kind(nk)
        try:
            t1, tri1, uv1, _ = H.trace(s.flat, org, dirs)
        finally:
            H.set_node_kind(old)
        assert np.array_equal(tri0, tri1) and np.array_equal(t0.view(np.uint32), t1.view(np.uint32)) and np.array_equal(uv0.view(np.uint32), uv1.view(np.uint32)), (name, nk)


def test_hostile_table_values_device_code_equals_oracle():
    """45 scenes with hostile numbers in their material / light / light-triangle / texture-coordinate / camera tables (scene_util.poison_tables: NaN, +-inf, +-1e38,
    denormal, negative, > 1): same image bits, same ray counts, every flag.  (Round 4 found one field this way: the direct term's radiance is the light MATERIAL's,
    pathTracing.cpp:65 — the device read trt_light::radiance, the same number in every loaded scene and a different one in a caller's hands.)"""
    import scene_util as SU
    for trial in range(45):
        rng = np.random.default_rng(trial)
        name = ["back", "veach-mis", "staircase"][trial % 3]
        s = T.Scene.named(name, 40, 30)
        what = SU.poison_tables(s, rng)
        p = T.make_params(40, 30, 4, 1000 + trial, flags=int(rng.choice([0, T.TRT_FLAG_FIXED_NEE, T.TRT_FLAG_RAY_OFFSET, T.TRT_FLAG_SPECULAR_KS])))
        ref, ost = O.render(s.flat, p)
        img, rays = H.render(s.flat, p)
        assert np.array_equal(ref.view(np.uint32), img.view(np.uint32)), (name, what, p.flags)
        assert rays == [ost.rays_camera, ost.rays_shadow, ost.rays_indirect], (name, what, p.flags)
        s.close()


def _vertex_eye_scene(tmp_path):
    """One emissive triangle at coordinates of 7e4 (the 0.001 pad of bvh.cpp:31-40 is below one ulp there: the vertex IS a corner of the leaf's box) seen by a
    camera that sits on one of its vertices: rays whose direction has an exactly zero component start ON a box plane, the slab test multiplies 0 by inf, and
    glm::min / glm::max (bvh.cpp:238-242) lose the constraint of that axis and of the ones nested inside it — the reference ENTERS the box and finds the triangle."""
    import scene_util as SU
    obj = ("v 71246.609375 79342.484375 -881.6929931640625\nv 72923.890625 77794.8359375 476.5187683105469\nv 74036.484375 79794.8046875 -4266.68115234375\n"
           "vn 0 0 1\nvt 0 0\nusemtl lamp\nf 1/1/1 2/1/1 3/1/1\n")
    SU.write_scene(tmp_path, "eye", obj, SU.MTL_BASIC, lights=[("lamp", (4, 8, 9.5))], w=24, h=5, fovy=20.0, eye=(74036.484375, 79794.8046875, -4266.68115234375),
                   lookat=(72698.703125, 79148.5546875, -1440.929931640625))
    return SU.load(tmp_path, "eye", leaf_num=15)


def test_zero_direction_component_on_a_box_plane_takes_the_literal_slab_test(tmp_path):
    """tools/fuzz_scenes.py, round 4: 155 of 20 000 camera rays of this scene have d.y == 0 exactly and the oracle (the literal slab test) finds the triangle for them;
    the fast form (fminf / fmaxf: a NaN dropped from either side) missed it.  Rays with such a direction now walk the caller's BVH2 with the literal test
    (trt_path.h raySpecial / traceClosestBvh2Glm): same hits, same image, both node kinds."""
    s = _vertex_eye_scene(tmp_path)
    f = s.flat.contents
    rng = np.random.default_rng(3)
    rays = [O.camera_ray(f.camera, 24, 5, 1, int(rng.integers(17, 22)), float(np.float32(rng.random())), float(np.float32(rng.random()))) for _ in range(20000)]
    org, dirs = np.array([r[0] for r in rays], np.float32), np.array([r[1] for r in rays], np.float32)
    assert ((dirs == 0).any(1)).sum() > 50
    t0, tri0, uv0 = O.trace(s.flat, org, dirs)
    assert (tri0[(dirs == 0).any(1)] >= 0).sum() > 50   # the reference does find the triangle through the poisoned test
    p = T.make_params(24, 5, 16, 4106463245, max_depth=3)
    ref, ost = O.render(s.flat, p)
    for nk in (0, 1):
        old = H.set_node_kind(nk)
        try:
            t1, tri1, uv1, _ = H.trace(s.flat, org, dirs)
            img, rays_n = H.render(s.flat, p)
        finally:
            H.set_node_kind(old)
        assert np.array_equal(tri0, tri1) and np.array_equal(t0, t1) and np.array_equal(uv0, uv1), nk
        assert np.array_equal(ref.view(np.uint32), img.view(np.uint32)), nk
    s.close()


def test_a_light_beyond_the_references_infinity_is_never_seen(tmp_path):
    """Q7 (bvh.h:5: INF = 114514): a light 128 000 units away is never the closest hit of a shadow ray, so it lights nothing — also when the ray carries a search
    hint of 1.001 |x' - x| > INF (the 8-wide path took the hint as its bound and found the light; tools/fuzz_scenes.py, round 4), and in TRT_FLAG_FIXED_NEE mode."""
    import scene_util as SU
    lines, body, vb = ["vt 0 0", "vn 0 0 1"], "", 1
    for (z, m, half) in ((0.0, "white", 3.0), (-2.0, "shiny", 4.0)):
        v, f, vb = SU.quad(-half, half, -half, half, z, vb)
        lines += v
        body += f"usemtl {m}\n" + "\n".join(x.format(n=1) for x in f) + "\n"
    v, f, vb = SU.quad(-20000.0, 20000.0, -20000.0, 20000.0, 128000.0, vb)
    lines += v
    body += "usemtl lamp\n" + "\n".join(x.format(n=1) for x in f) + "\n"
    SU.write_scene(tmp_path, "far", "\n".join(lines) + "\n" + body, SU.MTL_BASIC, lights=[("lamp", (50, 50, 50))], w=24, h=16, eye=(0, 0, 9), lookat=(0, 0, 0))
    s = SU.load(tmp_path, "far", leaf_num=2)
    for flags in (0, T.TRT_FLAG_FIXED_NEE):
        p = T.make_params(24, 16, 8, 11, flags=flags)
        ref, ost = O.render(s.flat, p)
        assert ost.rays_shadow > 500 and (flags != 0 or float(ref.max()) == 0.0)   # parity mode: shadow rays are traced, none finds the light (fixed: a miss is visible)
        for nk in (0, 1):
            old = H.set_node_kind(nk)
            try:
                assert H.compressible(s.flat)
                img, rays_n = H.render(s.flat, p)
            finally:
                H.set_node_kind(old)
            assert np.array_equal(ref.view(np.uint32), img.view(np.uint32)), (flags, nk)
    s.close()


def test_random_scenes_device_code_equals_oracle():
    """tools/fuzz_scenes.py for a few seconds: random geometry, materials from every branch of nextRay(), 0-8 lights, random cameras, both builders, leaves of 1-15."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_scenes.py"), "--seconds", "12", "--seed", "5"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "all bit-identical" in r.stdout
