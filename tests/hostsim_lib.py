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
        _lib = L
    return _lib


def set_node_kind(nk):
    """0 = exact 4-wide nodes (the default, as in trt_create), 1 = compressed 64-B nodes wherever the tree is nested; returns the old setting."""
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
