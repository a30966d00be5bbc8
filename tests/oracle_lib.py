"""ctypes binding of oracle/liboracle.so — the CPU checker.  Test infrastructure:
imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os

import numpy as np

from tinyraytracing_amd import _abi
from tinyraytracing_amd._abi import BvhNode, Camera, Material, Params, SceneFlat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")

MODE_ITERATIVE, MODE_RECURSIVE = 0, 1
MODE_EXPERIMENT_NO_RR_DIV = 0x100  # or-ed into `mode`: oracle.h
MODE_EXPERIMENT_SPECULAR_KS = 0x200
MODE_EXPERIMENT_GLASS_MIRROR, MODE_EXPERIMENT_NO_TR_ON_EMITTER, MODE_EXPERIMENT_NO_NEE_ON_GLASS = 0x400, 0x800, 0x1000
TRACE_REFERENCE, TRACE_BRUTE = 0, 1


class OracleStats(C.Structure):
    _fields_ = [("rays_camera", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_indirect", C.c_uint64),
                ("shaded_hits", C.c_uint64), ("inner_visits", C.c_uint64 * 2), ("tri_tests", C.c_uint64 * 2),
                ("max_bounces", C.c_uint32), ("threads", C.c_int32), ("seconds", C.c_double)]

    @property
    def rays(self):
        return self.rays_camera + self.rays_shadow + self.rays_indirect


_lib = None
fp = C.POINTER(C.c_float)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_SO):
        raise RuntimeError(f"{ORACLE_SO} missing: run `make oracle`")
    L = C.CDLL(ORACLE_SO)
    L.oracle_render.argtypes = [C.POINTER(SceneFlat), C.POINTER(Params), fp, C.POINTER(OracleStats), C.c_int, C.c_int]
    L.oracle_trace.argtypes = [C.POINTER(SceneFlat), C.c_uint64, fp, fp, fp, C.POINTER(C.c_int32), fp, C.c_int, C.POINTER(OracleStats)]
    L.oracle_build_bvh.argtypes = [C.c_uint32, fp, C.c_int, C.POINTER(C.c_uint32), C.POINTER(BvhNode), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.oracle_tri_test.argtypes = [fp, fp, fp, fp]
    L.oracle_aabb.restype = C.c_float
    L.oracle_aabb.argtypes = [fp, fp, fp, fp]
    L.oracle_sample.argtypes = [fp, C.c_int, C.c_float, C.c_float, C.c_float, fp]
    L.oracle_reflect.argtypes = [fp, fp, fp]
    L.oracle_refract.argtypes = [fp, fp, C.c_float, fp]
    L.oracle_camera_ray.argtypes = [C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, fp, fp]
    L.oracle_camera_ray_mode.argtypes = [C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, fp, fp]
    L.oracle_next_ray.argtypes = [C.POINTER(Material), fp, fp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), fp]
    L.oracle_prims_sincos2pi.argtypes = [C.c_float, fp, fp]
    L.oracle_prims_pow01.restype = C.c_float
    L.oracle_prims_pow01.argtypes = [C.c_float, C.c_float]
    L.oracle_prims_uniform.restype = C.c_float
    L.oracle_prims_uniform.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.oracle_render_literal.argtypes = [C.POINTER(SceneFlat), C.POINTER(Params), fp, C.POINTER(OracleStats), C.c_int]
    L.oracle_tri_test_literal.argtypes = [fp, fp, fp, fp]
    L.oracle_trace_literal.argtypes = [C.POINTER(SceneFlat), C.c_uint64, fp, fp, fp, C.POINTER(C.c_int32), fp]
    L.oracle_debug_path.argtypes = [C.POINTER(SceneFlat), C.POINTER(Params), C.c_int, C.c_int, C.c_int, fp, C.c_int]
    _lib = L
    return L


def _f3(a):
    arr = np.ascontiguousarray(a, dtype=np.float32)
    return arr, arr.ctypes.data_as(fp)


def render(flat, params, threads=0, mode=MODE_ITERATIVE):
    nrows = len(_rows(params))
    tw = params.x1 - params.x0
    out = np.empty((nrows, tw, 3), np.float32)
    st = OracleStats()
    rc = lib().oracle_render(flat, C.byref(params), out.ctypes.data_as(fp), C.byref(st), threads, mode)
    if rc != 0:
        raise RuntimeError(f"oracle_render failed: {rc}")
    return out, st


def render_literal(flat, params, threads=0):
    """ORACLE_MODE_LITERAL: the reference's own arithmetic (oracle/oracle_literal.cpp)."""
    nrows = len(_rows(params))
    tw = params.x1 - params.x0
    out = np.empty((nrows, tw, 3), np.float32)
    st = OracleStats()
    rc = lib().oracle_render_literal(flat, C.byref(params), out.ctypes.data_as(fp), C.byref(st), threads)
    if rc != 0:
        raise RuntimeError(f"oracle_render_literal failed: {rc}")
    return out, st


EXP_RACY_ACCUM, EXP_SHARED_ENGINES, EXP_INDEPENDENT_ENGINES = 1, 2, 3
ORACLE_EXP_SO = os.path.join(ROOT, "oracle", "liboracle_exp.so")
_exp_lib = None


def exp_lib():
    """oracle/liboracle_exp.so: the oracle's sources built with -DORACLE_EXPERIMENTS.  Only the experiment tests load it; the
    library behind every parity test (liboracle.so) contains no data race."""
    global _exp_lib
    if _exp_lib is None:
        _exp_lib = C.CDLL(ORACLE_EXP_SO)
    return _exp_lib



def render_literal_experiment(flat, params, experiment, threads=0):
    """oracle_render_literal_experiment (oracle.h): the reference's racy accumulation / its shared random engines.  Whole image."""
    out = np.empty((params.height, params.width, 3), np.float32)
    L = exp_lib()
    L.oracle_render_literal_experiment.argtypes = [C.POINTER(SceneFlat), C.POINTER(Params), fp, C.c_int, C.c_int]
    rc = L.oracle_render_literal_experiment(flat, C.byref(params), out.ctypes.data_as(fp), int(threads), int(experiment))
    if rc != 0:
        raise RuntimeError(f"oracle_render_literal_experiment failed: {rc}")
    return out


def trace_literal(flat, org, direction):
    org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
    direction = np.ascontiguousarray(direction, dtype=np.float32).reshape(-1, 3)
    n = org.shape[0]
    t = np.empty(n, np.float32)
    tri = np.empty(n, np.int32)
    uv = np.empty((n, 2), np.float32)
    rc = lib().oracle_trace_literal(flat, n, org.ctypes.data_as(fp), direction.ctypes.data_as(fp), t.ctypes.data_as(fp),
                                    tri.ctypes.data_as(C.POINTER(C.c_int32)), uv.ctypes.data_as(fp))
    if rc != 0:
        raise RuntimeError(f"oracle_trace_literal failed: {rc}")
    return t, tri, uv


def tri_test_literal(v, o, d):
    va, vp = _f3(v)
    oa, op = _f3(o)
    da, dp = _f3(d)
    out = np.zeros(4, np.float32)
    hit = lib().oracle_tri_test_literal(vp, op, dp, out.ctypes.data_as(fp))
    return bool(hit), out


def _rows(p):
    ys = range(p.y0, p.y1)
    if p.row_mod <= 1:
        return list(ys)
    return [y for y in ys if (y // p.row_block) % p.row_mod == p.row_rem]


def trace(flat, org, direction, mode=TRACE_REFERENCE, want_stats=False):
    org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
    direction = np.ascontiguousarray(direction, dtype=np.float32).reshape(-1, 3)
    n = org.shape[0]
    t = np.empty(n, np.float32)
    tri = np.empty(n, np.int32)
    uv = np.empty((n, 2), np.float32)
    st = OracleStats()
    rc = lib().oracle_trace(flat, n, org.ctypes.data_as(fp), direction.ctypes.data_as(fp), t.ctypes.data_as(fp),
                            tri.ctypes.data_as(C.POINTER(C.c_int32)), uv.ctypes.data_as(fp), mode, C.byref(st))
    if rc != 0:
        raise RuntimeError(f"oracle_trace failed: {rc}")
    return (t, tri, uv, st) if want_stats else (t, tri, uv)


def tri_test(v, o, d):
    va, vp = _f3(v)
    oa, op = _f3(o)
    da, dp = _f3(d)
    out = np.zeros(3, np.float32)
    hit = lib().oracle_tri_test(vp, op, dp, out.ctypes.data_as(fp))
    return bool(hit), out


def aabb(lo, hi, o, d):
    la, lp = _f3(lo)
    ha, hp = _f3(hi)
    oa, op = _f3(o)
    da, dp = _f3(d)
    return float(lib().oracle_aabb(lp, hp, op, dp))


def sample(axis, ray_type, Ns, u_phi, u_theta):
    aa, ap = _f3(axis)
    out = np.zeros(3, np.float32)
    lib().oracle_sample(ap, ray_type, Ns, u_phi, u_theta, out.ctypes.data_as(fp))
    return out


def reflect(I, N):
    ia, ip = _f3(I)
    na, np_ = _f3(N)
    out = np.zeros(3, np.float32)
    lib().oracle_reflect(ip, np_, out.ctypes.data_as(fp))
    return out


def refract(I, N, eta):
    ia, ip = _f3(I)
    na, np_ = _f3(N)
    out = np.zeros(3, np.float32)
    lib().oracle_refract(ip, np_, eta, out.ctypes.data_as(fp))
    return out


def camera_ray(cam, width, height, i, j, u1, u2, fixed=False):
    o = np.zeros(3, np.float32)
    d = np.zeros(3, np.float32)
    if fixed:
        lib().oracle_camera_ray_mode(C.byref(cam), width, height, i, j, u1, u2, 1, o.ctypes.data_as(fp), d.ctypes.data_as(fp))
    else:
        lib().oracle_camera_ray(C.byref(cam), width, height, i, j, u1, u2, o.ctypes.data_as(fp), d.ctypes.data_as(fp))
    return o, d


def build_bvh(tri_v, leaf_num=8):
    tv = np.ascontiguousarray(tri_v, dtype=np.float32).reshape(-1, 9)
    n = tv.shape[0]
    perm = np.zeros(n, np.uint32)
    nodes = (BvhNode * max(2 * n, 2))()
    nn = C.c_uint32()
    depth = C.c_uint32()
    rc = lib().oracle_build_bvh(n, tv.ctypes.data_as(fp), leaf_num, perm.ctypes.data_as(C.POINTER(C.c_uint32)), nodes, C.byref(nn), C.byref(depth))
    if rc != 0:
        raise RuntimeError(f"oracle_build_bvh failed: {rc}")
    return perm, nodes, nn.value, depth.value


def debug_path(flat, params, x, y, sample_idx, max_vertices=256):
    out = np.zeros((max_vertices, 8), np.float32)
    n = lib().oracle_debug_path(flat, C.byref(params), x, y, sample_idx, out.ctypes.data_as(fp), max_vertices)
    return out[:max(n, 0)]
