"""ctypes binding of tests/hostsim/libhostsim.so (device-side path functions compiled for the CPU)."""
import ctypes as C
import os

import numpy as np

import tinyraytracing_amd as T
from tinyraytracing_amd._abi import Params, SceneFlat

SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "libhostsim.so")
fp = C.POINTER(C.c_float)
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(SO)
        L.hostsim_render.argtypes = [C.POINTER(SceneFlat), C.POINTER(Params), fp, C.POINTER(C.c_uint64)]
        L.hostsim_trace.argtypes = [C.POINTER(SceneFlat), C.c_uint64, fp, fp, fp, C.POINTER(C.c_int32), fp, C.POINTER(C.c_uint64)]
        L.hostsim_set_node_kind.argtypes = [C.c_int]
        L.hostsim_compressible.argtypes = [C.POINTER(SceneFlat)]
        L.hostsim_trace_counts.argtypes = [C.POINTER(SceneFlat), C.c_int, C.c_uint64, fp, fp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.hostsim_oct_fallbacks.restype = C.c_uint64
        L.hostsim_oct_fallbacks.argtypes = [C.POINTER(SceneFlat), C.c_uint64, fp, fp]
        L.hostsim_oct_info.argtypes = [C.POINTER(SceneFlat), C.POINTER(C.c_uint64)]
        L.hostsim_tree_hashes.argtypes = [C.POINTER(SceneFlat), C.c_uint, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def set_node_kind(nk):
    """0 = exact 4-wide nodes, 1 = compressed 80-B 8-wide nodes (trt_oct.h) wherever the tree qualifies; returns the old setting."""
    return lib().hostsim_set_node_kind(int(nk))


def compressible(flat):
    return bool(lib().hostsim_compressible(flat))


def render(flat, p):
    rows = len(T.rows_selected(p))
    out = np.empty((rows, p.x1 - p.x0, 3), np.float32)
    rays = (C.c_uint64 * 3)()
    rc = lib().hostsim_render(flat, C.byref(p), out.ctypes.data_as(fp), rays)
    assert rc == 0
    return out, [int(x) for x in rays]


def trace(flat, org, direction):
    org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
    direction = np.ascontiguousarray(direction, np.float32).reshape(-1, 3)
    n = org.shape[0]
    t = np.empty(n, np.float32)
    tri = np.empty(n, np.int32)
    uv = np.empty((n, 2), np.float32)
    cnt = (C.c_uint64 * 2)()
    rc = lib().hostsim_trace(flat, n, org.ctypes.data_as(fp), direction.ctypes.data_as(fp), t.ctypes.data_as(fp),
                             tri.ctypes.data_as(C.POINTER(C.c_int32)), uv.ctypes.data_as(fp), cnt)
    assert rc == 0
    return t, tri, uv, [int(x) for x in cnt]


def trace_counts(flat, node_kind, org, direction):
    """Per-ray work of the closest-hit search on node kind 0 / 1: (inner-node visits, triangle tests)."""
    org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
    direction = np.ascontiguousarray(direction, np.float32).reshape(-1, 3)
    n = org.shape[0]
    v = np.zeros(n, np.uint32)
    t = np.zeros(n, np.uint32)
    u32 = C.POINTER(C.c_uint32)
    rc = lib().hostsim_trace_counts(flat, int(node_kind), n, org.ctypes.data_as(fp), direction.ctypes.data_as(fp), v.ctypes.data_as(u32), t.ctypes.data_as(u32))
    assert rc == 0
    return v, t


def oct_fallbacks(flat, org, direction):
    """How many of the rays end the oct traversal on a result that fails the check (and take the exact form)."""
    org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
    direction = np.ascontiguousarray(direction, np.float32).reshape(-1, 3)
    return int(lib().hostsim_oct_fallbacks(flat, org.shape[0], org.ctypes.data_as(fp), direction.ctypes.data_as(fp)))


def oct_info(flat):
    """(nodes, levels, triangle records) of the oct tree, or None when the tree does not qualify."""
    out = (C.c_uint64 * 3)()
    return None if lib().hostsim_oct_info(flat, out) else tuple(int(x) for x in out)


def tree_hashes(flat, threads):
    """Hashes of the 4-wide trees (both collapses), the 8-wide tree, its triangle records and the leaf boxes, built with `threads` host threads."""
    out = (C.c_uint64 * 8)()
    assert lib().hostsim_tree_hashes(flat, int(threads), out) == 0
    return [int(x) for x in out]
