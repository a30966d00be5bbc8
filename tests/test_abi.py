"""The C-ABI libraries load and export every symbol include/*.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from tinyraytracing_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    lib = C.CDLL(os.path.join(_abi.LIB_DIR, "libtrt_hip.so"))
    names = _declared("trt.h", "trt_")
    assert set(names) == set(_abi.HIP_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n


def test_host_library_exports_every_declared_symbol():
    lib = C.CDLL(os.path.join(_abi.LIB_DIR, "libtrt_host.so"))
    names = _declared("trt_host.h", "trth_")
    assert set(names) == set(_abi.HOST_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n


def test_builder_library_exports_every_declared_symbol_and_fails_loudly_without_a_device():
    lib = C.CDLL(os.path.join(_abi.LIB_DIR, "libtrt_lbvh.so"))
    names = _declared("trt_build.h", "trt_build_")
    assert set(names) == set(_abi.BUILD_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n
    # argument checks come before any device is touched
    b = _abi.load_build()
    nodes = (_abi.BvhNode * 4)()
    order = (C.c_uint32 * 4)()
    nn = C.c_uint32(0)
    v = (C.c_float * 36)()
    assert b.trt_build_lbvh(v, 4, 0, 0, nodes, 4, C.byref(nn), order, None, None) == 1 and b"leaf_num" in b.trt_build_last_error()
    assert b.trt_build_lbvh(v, 4, 2, 0, None, 4, C.byref(nn), order, None, None) == 1 and b"null" in b.trt_build_last_error()
    import torch
    if not torch.cuda.is_available():
        assert b.trt_build_lbvh(v, 4, 2, 0, nodes, 4, C.byref(nn), order, None, None) == 4 and b"HIP device" in b.trt_build_last_error()


def test_oracle_exports_every_declared_symbol():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    text = open(os.path.join(ROOT, "oracle", "oracle.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for n in set(re.findall(r"\b(oracle_[a-z0-9_]+)\s*\(", text)):
        assert hasattr(lib, n), n


def test_ctypes_mirror_matches_c_struct_sizes():
    lib = _abi.load_host()
    sizes = (C.c_int64 * 12)()
    assert lib.trth_abi_sizes(sizes) == 0
    mirror = [_abi.BvhNode, _abi.Material, _abi.Light, _abi.LightTri, _abi.Texture, _abi.Camera, _abi.SceneFlat, _abi.Params, _abi.Stats]
    assert [int(s) for s in sizes[:9]] == [C.sizeof(m) for m in mirror]
    assert C.sizeof(_abi.BvhNode) == 64
    assert sizes[9] == _abi.TRT_ABI_VERSION


def test_abi_version_and_rows_selected_without_gpu():
    lib = _abi.load_hip()
    assert lib.trt_abi_version() == _abi.TRT_ABI_VERSION
    import tinyraytracing_amd as T
    p = T.make_params(64, 37, 1, 0, rows=(8, 3, 1))
    assert lib.trt_rows_selected(C.byref(p)) == len(T.rows_selected(p))
    p = T.make_params(64, 37, 1, 0, tile=(3, 5, 20, 30))
    assert lib.trt_rows_selected(C.byref(p)) == 25


def test_product_has_no_oracle_dependency():
    """Nothing under tinyraytracing_amd/ or include/ may reference oracle/ or the hostsim."""
    bad = []
    for base in ("tinyraytracing_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if not f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                    continue
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"#include\s+[\"<][^\">]*oracle", txt) or re.search(r"^\s*(import|from)\s+oracle", txt, flags=re.M) or "liboracle" in txt or "libhostsim" in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_missing_hip_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_abi, "_hip", None)
    monkeypatch.setattr(_abi, "LIB_DIR", str(tmp_path))
    with pytest.raises(RuntimeError, match="only compute path"):
        _abi.load_hip()


def test_trt_create_validates_the_bvh_before_touching_a_device():
    """validateBvh runs on the host first: a foreign tree with a bad index, a cycle, or siblings out of post-BVH order is
    refused with TRT_EINVAL and a message — also on a machine without a GPU."""
    import tinyraytracing_amd as T
    from tinyraytracing_amd._abi import BvhNode, SceneFlat
    lib = _abi.load_hip()
    s = T.Scene.named("veach-mis", 32, 18)
    f = s.flat.contents
    g = SceneFlat()
    C.memmove(C.byref(g), C.byref(f), C.sizeof(SceneFlat))
    nodes = (BvhNode * f.n_nodes)()
    g.nodes = nodes
    h = C.c_void_p()
    for what, needle in (("index", b"out of range"), ("cycle", b"twice"), ("order", b"post-BVH order"), ("nan", b"NaN")):
        C.memmove(nodes, f.nodes, C.sizeof(BvhNode) * f.n_nodes)
        if what == "index":
            nodes[3].child1 = 0x7FFFFFF0
        elif what == "nan":
            nodes[7].hi1[1] = float("nan")
        elif what == "cycle":
            nodes[5].child0 = 0
        else:
            nodes[0].child0, nodes[0].child1 = f.nodes[0].child1, f.nodes[0].child0
        assert lib.trt_create(C.byref(g), 0, C.byref(h)) == 1
        assert needle in lib.trt_last_error(), (what, lib.trt_last_error())


def test_random_damage_to_a_tree_is_refused_or_harmless():
    """400 random mutations of the caller's node array (child words replaced by random bits, by other nodes' words, leaf counts and first indices nudged,
    boxes made NaN / inverted): trt_create answers each with TRT_EINVAL and a message, or — where the damage leaves a tree the rules accept (a box is not
    a rule: foreign boxes need not nest) — goes on to the device; it never crashes, loops or reads outside the arrays."""
    import numpy as np
    import tinyraytracing_amd as T
    from tinyraytracing_amd._abi import BvhNode, SceneFlat
    lib = _abi.load_hip()
    s = T.Scene.named("veach-mis", 32, 18)
    f = s.flat.contents
    g = SceneFlat()
    C.memmove(C.byref(g), C.byref(f), C.sizeof(SceneFlat))
    nodes = (BvhNode * f.n_nodes)()
    g.nodes = nodes
    words = np.ctypeslib.as_array(C.cast(nodes, C.POINTER(C.c_uint32)), (f.n_nodes, C.sizeof(BvhNode) // 4))
    floats = words.view(np.float32)
    child_cols = [BvhNode.child0.offset // 4, BvhNode.child1.offset // 4]
    rng = np.random.default_rng(11)
    h = C.c_void_p()
    refused = 0
    for it in range(400):
        C.memmove(nodes, f.nodes, C.sizeof(BvhNode) * f.n_nodes)
        for _ in range(int(rng.integers(1, 4))):
            n, c = int(rng.integers(0, f.n_nodes)), child_cols[int(rng.integers(0, 2))]
            k = int(rng.integers(0, 6))
            if k == 0:
                words[n, c] = rng.integers(0, 2**32, dtype=np.uint64).astype(np.uint32)
            elif k == 1:
                words[n, c] = words[int(rng.integers(0, f.n_nodes)), child_cols[int(rng.integers(0, 2))]]
            elif k == 2:
                words[n, c] ^= np.uint32(1 << int(rng.integers(0, 32)))
            elif k == 3:
                words[n, c] = np.uint32((int(words[n, c]) + int(rng.integers(-3, 4))) & 0xFFFFFFFF)
            elif k == 4:
                col = int(rng.integers(0, words.shape[1]))
                if col not in child_cols:
                    floats[n, col] = [np.nan, np.inf, -np.inf, 1e38, -1e38][int(rng.integers(0, 5))]
            else:
                words[n, child_cols[0]], words[n, child_cols[1]] = words[n, child_cols[1]], words[n, child_cols[0]]
        rc = lib.trt_create(C.byref(g), 0, C.byref(h))
        assert rc != 0 or h.value
        if rc == 0:
            lib.trt_destroy(h)
        else:
            assert lib.trt_last_error()
            refused += b"HIP device" not in lib.trt_last_error() and b"gfx950" not in lib.trt_last_error()
    assert refused > 200
    s.close()


def test_validation_on_several_host_threads_finds_faults_deep_in_a_big_tree(monkeypatch):
    """trt_create validates subtrees of a cut of the tree side by side (atomic marks): the same faults are found when they sit deep
    below the cut of a 90 k-node tree, with 1 and with 8 host threads; the untouched tree passes validation (and then fails for
    want of a device on a machine without one, or succeeds)."""
    import numpy as np
    import tinyraytracing_amd as T
    from tinyraytracing_amd._abi import BvhNode, SceneFlat
    lib = _abi.load_hip()
    s = T.Scene.named("blob", 32, 18, n=150000)
    f = s.flat.contents
    g = SceneFlat()
    C.memmove(C.byref(g), C.byref(f), C.sizeof(SceneFlat))
    nodes = (BvhNode * f.n_nodes)()
    g.nodes = nodes
    LEAF = 0x80000000
    inner = [n for n in range(f.n_nodes - 2000, f.n_nodes) if not (f.nodes[n].child0 & LEAF) and not (f.nodes[n].child1 & LEAF)]
    leafy = [n for n in range(f.n_nodes - 2000, f.n_nodes) if (f.nodes[n].child0 & LEAF) and (f.nodes[n].child1 & LEAF)]
    assert inner and leafy
    h = C.c_void_p()
    for threads in ("1", "8"):
        monkeypatch.setenv("TRT_HOST_THREADS", threads)
        for what, needle in (("index", b"out of range"), ("cycle", b"twice"), ("order", b"post-BVH order"), ("shared", (b"two leaves", b"post-BVH order")), ("none", None)):
            C.memmove(nodes, f.nodes, C.sizeof(BvhNode) * f.n_nodes)
            n = inner[len(inner) // 2]
            if what == "index":
                nodes[n].child1 = 0x7FFFFFF0
            elif what == "cycle":
                nodes[n].child0 = n - 1000
            elif what == "order":
                nodes[n].child0, nodes[n].child1 = f.nodes[n].child1, f.nodes[n].child0
            elif what == "shared":
                nodes[leafy[0]].child1 = f.nodes[leafy[-1]].child0
            rc = lib.trt_create(C.byref(g), 0, C.byref(h))
            if needle is None:
                assert rc == 0 or b"HIP device" in lib.trt_last_error() or b"gfx950" in lib.trt_last_error(), lib.trt_last_error()
                if rc == 0:
                    lib.trt_destroy(h)
            else:
                # (a triangle range shared by two leaves also breaks the index order of some ancestor's children: whichever walk gets there first)
                needles = needle if isinstance(needle, tuple) else (needle,)
                assert rc == 1 and any(x in lib.trt_last_error() for x in needles), (threads, what, rc, lib.trt_last_error())
    s.close()


def test_adopting_a_foreign_tree_reorders_the_triangles_like_the_in_place_sort():
    """trth_scene_vertices / trth_scene_adopt_bvh (the host side of the GPU builder's hand-over): adopting the host builder's own nodes with the
    identity order leaves the flat scene as it was; a reversed two-leaf tree with the reversed order reverses the triangle arrays; an order that is
    not a permutation is refused."""
    import numpy as np
    import tinyraytracing_amd as T
    from tinyraytracing_amd._abi import BvhNode
    host = _abi.load_host()
    s = T.Scene.named("veach-mis", 32, 18)
    f = s.flat.contents
    n, nn = f.n_tris, f.n_nodes
    v0 = np.ctypeslib.as_array(f.tri_v, shape=(n * 9,)).copy()
    mat0 = np.ctypeslib.as_array(f.tri_mat, shape=(n,)).copy()
    nodes = (BvhNode * nn)()
    C.memmove(nodes, f.nodes, C.sizeof(BvhNode) * nn)
    v = np.empty(n * 9, np.float32)
    assert host.trth_scene_vertices(s._h, v.ctypes.data_as(C.POINTER(C.c_float)), v.size) == 0 and np.array_equal(v, v0)
    ident = np.arange(n, dtype=np.uint32)
    assert host.trth_scene_adopt_bvh(s._h, nodes, nn, ident.ctypes.data_as(C.POINTER(C.c_uint32)), f.bvh_depth) == 0
    f = s.flat.contents
    assert np.array_equal(np.ctypeslib.as_array(f.tri_v, shape=(n * 9,)), v0) and f.n_nodes == nn
    rev = ident[::-1].copy()
    assert host.trth_scene_adopt_bvh(s._h, nodes, nn, rev.ctypes.data_as(C.POINTER(C.c_uint32)), f.bvh_depth) == 0
    f = s.flat.contents
    assert np.array_equal(np.ctypeslib.as_array(f.tri_v, shape=(n, 9)), v0.reshape(n, 9)[::-1])
    assert np.array_equal(np.ctypeslib.as_array(f.tri_mat, shape=(n,)), mat0[::-1])
    bad = ident.copy()
    bad[3] = bad[4]
    assert host.trth_scene_adopt_bvh(s._h, nodes, nn, bad.ctypes.data_as(C.POINTER(C.c_uint32)), 1) == 1 and b"permutation" in host.trth_last_error()
    s.close()
