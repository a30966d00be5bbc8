"""tinyraytracing_amd — MI355X-native path-tracing hot path of TinyRayTracing.

Python is the thin host layer used by tests and bench.py; the compute path is
libtrt_hip.so (hand-written HIP for gfx950, include/trt.h) and the loaders are
libtrt_host.so (C++, include/trt_host.h).  The classes mirror the reference's
driver (main.cpp:44-119): load the scene (readxml -> readobj -> readmtl), build
the BVH, render(), imshow().
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from ._abi import Params, Stats, SceneFlat, TRT_FLAG_COUNT, TRT_FLAG_TIMING, TRT_FLAG_OVERLAP, TRT_FLAG_FIXED_NEE, TRT_FLAG_FIXED_PIXELS, TRT_FLAG_RAY_OFFSET, TRT_FLAG_SPECULAR_KS, KERNEL_NAMES  # noqa: F401

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES_DIR = os.path.join(REPO_ROOT, "scenes")

# seeds fixed by SURVEY.md §8d / BASELINE.md §2
SEED_BACK = 0x5EED0001
SEED_SOUP = 0x5EED0003
SEED_STAIRCASE = 0x5EED0004
SEED_BLOB = 0x5EED0005

_BUILDERS = {"sweep": 0, "binned": 1, "auto": 2}
# Triangles per BVH leaf.  The reference calls buildBVH(..., 8) (main.cpp:76); on the GPU 2 measured fastest on every
# scene larger than the Cornell box, and the topology is free: only nearest-hit + tie rules matter.
DEFAULT_LEAF = 2
TINY_LEAF = 8
SOUP_LEAF = 1   # unstructured triangle soup: one triangle per leaf (soup-1M 16 spp: 112 -> 106 ms/step)


def default_leaf(name, n_triangles):
    """Leaf size Scene.named() builds with when none is given (the topology is free: any valid tree gives the hits
    of the oracle on that same tree)."""
    if n_triangles <= 64:
        return TINY_LEAF
    return SOUP_LEAF if name == "soup" else DEFAULT_LEAF


class TrtError(RuntimeError):
    pass


class Scene:
    """Scene + BVH + flat arrays (reference: Scene in scene.h:19-36 and buildBVH, bvh.cpp:16)."""

    def __init__(self, handle):
        self._lib = _abi.load_host()
        self._h = handle
        self._built = False

    @classmethod
    def load(cls, xml_path, obj_path, mtl_path, basedir, width=0, height=0, triangulate_polygons=False):
        """scene.readxml(xml); scene.readobj(obj); scene.readmtl(mtl, basedir) (main.cpp:66-69).
        width/height override the XML resolution; triangulate_polygons fans faces of more than three vertices instead of
        keeping their first three only (the reference's behaviour, scene.cpp:162, and the default)."""
        lib = _abi.load_host()
        h = lib.trth_scene_load_opts(os.fsencode(xml_path), os.fsencode(obj_path), os.fsencode(mtl_path),
                                     os.fsencode(basedir), int(width), int(height), 1 if triangulate_polygons else 0)
        if not h:
            raise TrtError(lib.trth_last_error().decode())
        return cls(h)

    @classmethod
    def named(cls, name, width=0, height=0, leaf_num=None, builder="auto", n=None, seed=None, device=0):
        """Shipped and synthetic scenes: back, veach-mis, staircase, soup (n random triangles in
        the back box, BASELINE config 3), blob (displaced geodesic sphere, config 5).  `device`: where builder="lbvh" runs
        (a rank of a multi-GPU job passes its own GPU; the host builders ignore it)."""
        if name in ("back", "veach-mis", "staircase"):
            d = os.path.join(SCENES_DIR, name)
            s = cls.load(os.path.join(d, name + ".xml"), os.path.join(d, name + ".obj"), os.path.join(d, name + ".mtl"), d, width, height)
        elif name in ("soup", "blob"):
            d = os.path.join(SCENES_DIR, "back")
            s = cls.load(os.path.join(d, "back.xml"), os.path.join(d, "back.obj"), os.path.join(d, "back.mtl"), d, width, height)
            s._check(s._lib.trth_scene_drop_tris(s._h, 6, 12))  # the cube of back.obj (faces 7..18)
            if name == "soup":
                s._check(s._lib.trth_scene_add_soup(s._h, SEED_SOUP if seed is None else seed, 1_000_000 if n is None else int(n)))
            else:
                s._check(s._lib.trth_scene_add_blob(s._h, SEED_BLOB if seed is None else seed, 10_000_000 if n is None else int(n)))
        else:
            raise TrtError(f"unknown scene {name!r}")
        if leaf_num is None:
            # tiny scenes are walked wave-uniformly (every node, every triangle: trt_kernels.h IMPL 0), where fewer,
            # fuller leaves are cheaper; everything else is traversed per ray, where 2 measured best
            leaf_num = default_leaf(name, s.info["n_triangles"])
        s.build_bvh(leaf_num, builder, device)
        return s

    def _check(self, rc):
        if rc != 0:
            raise TrtError(self._lib.trth_last_error().decode())

    def build_bvh(self, leaf_num=DEFAULT_LEAF, builder="auto", device=0):
        """BVHNode* root = buildBVH(scene.triangles, 0, n-1, leaf_num) (main.cpp:76 passes 8) + flattening.
        builder: "sweep" / "binned" / "auto" = the host builders (exact SAH up to 64 k triangles, 32-bin SAH above); "lbvh" = the GPU
        builder of include/trt_build.h on `device` (Morton order + radix tree; `self.build_ms` = (device ms, whole call ms))."""
        if builder == "lbvh":
            lib = _abi.load_build()
            n = self.info["n_triangles"]
            v = np.empty(max(n, 1) * 9, np.float32)
            self._check(self._lib.trth_scene_vertices(self._h, v.ctypes.data_as(C.POINTER(C.c_float)), v.size))
            cap = max(n, 2) - 1
            node_bytes = np.empty(cap * C.sizeof(_abi.BvhNode), np.uint8)  # (a ctypes array of 10 M nodes would be zeroed first)
            nodes = C.cast(node_bytes.ctypes.data, C.POINTER(_abi.BvhNode))
            order = np.empty(max(n, 1), np.uint32)
            n_nodes, depth = C.c_uint32(0), C.c_uint32(0)
            ms = (C.c_double * 2)()
            rc = lib.trt_build_lbvh(v.ctypes.data_as(C.POINTER(C.c_float)), n, int(leaf_num), int(device), nodes, cap, C.byref(n_nodes),
                                    order.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(depth), ms)
            if rc != 0:
                raise TrtError(f"trt_build_lbvh failed ({rc}): {lib.trt_build_last_error().decode()}")
            self.build_ms = (ms[0], ms[1])
            self._check(self._lib.trth_scene_adopt_bvh(self._h, nodes, n_nodes.value, order.ctypes.data_as(C.POINTER(C.c_uint32)), depth.value))
        else:
            self._check(self._lib.trth_scene_build(self._h, int(leaf_num), _BUILDERS[builder]))
        self._built = True
        return self

    @property
    def flat(self):
        p = self._lib.trth_scene_flat(self._h)
        if not p:
            raise TrtError(self._lib.trth_last_error().decode())
        return p

    @property
    def info(self):
        arr = (C.c_int64 * 8)()
        self._check(self._lib.trth_scene_info(self._h, arr))
        keys = ["width", "height", "n_vertices", "n_vn", "n_vt", "n_triangles", "n_materials", "n_lights"]
        return dict(zip(keys, [int(x) for x in arr]))

    def light_area(self, i):
        return float(self._lib.trth_scene_light_area(self._h, i))

    def material_name(self, i):
        s = self._lib.trth_scene_material_name(self._h, i)
        if s is None:
            raise TrtError(self._lib.trth_last_error().decode())
        return s.decode()

    # numpy views of the flat arrays (copies), mostly for tests
    def arrays(self):
        f = self.flat.contents
        n = f.n_tris
        out = {
            "tri_v": np.ctypeslib.as_array(f.tri_v, shape=(n, 3, 3)).copy() if n else np.zeros((0, 3, 3), np.float32),
            "tri_vn": np.ctypeslib.as_array(f.tri_vn, shape=(n, 3, 3)).copy() if n else np.zeros((0, 3, 3), np.float32),
            "tri_vt": np.ctypeslib.as_array(f.tri_vt, shape=(n, 3, 2)).copy() if n else np.zeros((0, 3, 2), np.float32),
            "tri_mat": np.ctypeslib.as_array(f.tri_mat, shape=(n,)).copy() if n else np.zeros((0,), np.int32),
            "n_nodes": int(f.n_nodes),
            "bvh_depth": int(f.bvh_depth),
        }
        return out

    def close(self):
        if self._h:
            self._lib.trth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_params(width, height, spp, seed, tile=None, rows=None, max_depth=0, flags=0, mem_budget=0):
    p = Params()
    p.width, p.height, p.spp, p.seed = int(width), int(height), int(spp), int(seed) & 0xFFFFFFFF
    x0, y0, x1, y1 = tile if tile is not None else (0, 0, width, height)
    p.x0, p.y0, p.x1, p.y1 = int(x0), int(y0), int(x1), int(y1)
    rb, rm, rr = rows if rows is not None else (1, 1, 0)
    p.row_block, p.row_mod, p.row_rem = int(rb), int(rm), int(rr)
    p.max_depth, p.flags, p.mem_budget = int(max_depth), int(flags), int(mem_budget)
    return p


def rows_selected(p):
    """Image rows a Params selects, in output order (mirrors trt_rows_selected)."""
    ys = range(p.y0, p.y1)
    if p.row_mod <= 1:
        return list(ys)
    return [y for y in ys if (y // p.row_block) % p.row_mod == p.row_rem]


class Renderer:
    """Owns a trt_handle: the scene resident in HBM of one MI355X."""

    def __init__(self, scene, device=0):
        self._lib = _abi.load_hip()  # raises if the HIP extension is missing
        self._scene = scene          # keep the flat arrays alive
        h = C.c_void_p()
        rc = self._lib.trt_create(scene.flat, int(device), C.byref(h))
        if rc != 0:
            raise TrtError(f"trt_create failed ({rc}): {self._lib.trt_last_error().decode()}")
        self._h = h
        self.device = int(device)

    def render(self, params):
        """render() -> (float32 image [rows, tile_w, 3] linear radiance, Stats).  Host output."""
        nrows = self._lib.trt_rows_selected(C.byref(params))
        tw = params.x1 - params.x0
        if nrows <= 0 or tw <= 0:
            raise TrtError("render: empty tile")
        out = np.empty((nrows, tw, 3), dtype=np.float32)
        st = Stats()
        rc = self._lib.trt_render(self._h, C.byref(params), out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st))
        if rc != 0:
            raise TrtError(f"trt_render failed ({rc}): {self._lib.trt_last_error().decode()}")
        return out, st

    def render_samples(self, params, sample_begin, sample_end, accum=None):
        """Progressive render (trt_render_samples): adds samples [sample_begin, sample_end) of params.spp onto
        `accum` (float64 [rows, tile_w, 3], the running per-pixel sums; None = start from zeros).
        Returns (image so far, accum, Stats); save accum + sample_end to checkpoint, pass them back to resume."""
        nrows = self._lib.trt_rows_selected(C.byref(params))
        tw = params.x1 - params.x0
        if nrows <= 0 or tw <= 0:
            raise TrtError("render_samples: empty tile")
        if accum is None:
            accum = np.zeros((nrows, tw, 3), dtype=np.float64)
        if accum.dtype != np.float64 or accum.shape != (nrows, tw, 3) or not accum.flags["C_CONTIGUOUS"]:
            raise TrtError("render_samples: accum must be a contiguous float64 array of shape (rows, tile_w, 3)")
        out = np.empty((nrows, tw, 3), dtype=np.float32)
        st = Stats()
        rc = self._lib.trt_render_samples(self._h, C.byref(params), int(sample_begin), int(sample_end), accum.ctypes.data_as(C.POINTER(C.c_double)),
                                          out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st))
        if rc != 0:
            raise TrtError(f"trt_render_samples failed ({rc}): {self._lib.trt_last_error().decode()}")
        return out, accum, st

    def render_into(self, params, out_tensor, stream_ptr=0):
        """Renders into a CUDA/HIP torch tensor (float32, >= rows*tile_w*3 elements) on this device."""
        nrows = self._lib.trt_rows_selected(C.byref(params))
        tw = params.x1 - params.x0
        need = nrows * tw * 3
        if out_tensor.numel() < need or str(out_tensor.dtype) != "torch.float32" or not out_tensor.is_cuda or not out_tensor.is_contiguous():
            raise TrtError("render_into: need a contiguous float32 device tensor with rows*tile_w*3 elements")
        st = Stats()
        rc = self._lib.trt_render_device(self._h, C.byref(params), C.c_void_p(out_tensor.data_ptr()), C.c_void_p(stream_ptr), C.byref(st))
        if rc != 0:
            raise TrtError(f"trt_render_device failed ({rc}): {self._lib.trt_last_error().decode()}")
        return st

    def trace_closest(self, org, direction, want_stats=False):
        """traverseBVH on a ray batch: returns (t, tri, uv[, Stats])."""
        org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
        direction = np.ascontiguousarray(direction, dtype=np.float32).reshape(-1, 3)
        n = org.shape[0]
        if direction.shape[0] != n:
            raise TrtError("trace_closest: org/dir length mismatch")
        t = np.empty(n, np.float32)
        tri = np.empty(n, np.int32)
        uv = np.empty((n, 2), np.float32)
        st = Stats()
        fp = C.POINTER(C.c_float)
        rc = self._lib.trt_trace_closest(self._h, n, org.ctypes.data_as(fp), direction.ctypes.data_as(fp), t.ctypes.data_as(fp),
                                         tri.ctypes.data_as(C.POINTER(C.c_int32)), uv.ctypes.data_as(fp), C.byref(st))
        if rc != 0:
            raise TrtError(f"trt_trace_closest failed ({rc}): {self._lib.trt_last_error().decode()}")
        return (t, tri, uv, st) if want_stats else (t, tri, uv)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.trt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GroupRenderer:
    """Owns a trt_group: the scene resident on several GPUs of one node; render() tiles the image over them in
    interleaved row stripes and gathers with ONE ncclGather on the first device (include/trt.h).  `devices` may name one
    device several times (a rehearsal on a one-GPU box: same code path, device copies instead of RCCL)."""

    def __init__(self, scene, devices):
        self._lib = _abi.load_hip()
        self._scene = scene
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        g = C.c_void_p()
        rc = self._lib.trt_group_create(scene.flat, len(devices), devs, C.byref(g))
        if rc != 0:
            raise TrtError(f"trt_group_create failed ({rc}): {self._lib.trt_last_error().decode()}")
        self._g = g
        self.devices = list(devices)

    def render(self, params):
        """-> (float32 image [tile rows, tile_w, 3], Stats summed over the devices, gather + un-interleave ms)."""
        th, tw = params.y1 - params.y0, params.x1 - params.x0
        out = np.empty((th, tw, 3), dtype=np.float32)
        st = Stats()
        gms = C.c_double(0.0)
        rc = self._lib.trt_group_render(self._g, C.byref(params), out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st), C.byref(gms))
        if rc != 0:
            raise TrtError(f"trt_group_render failed ({rc}): {self._lib.trt_last_error().decode()}")
        return out, st, gms.value

    def render_into(self, params, out_tensor):
        """trt_group_render_device: the image stays on devices[0] (a contiguous float32 torch tensor there with tile rows * tile_w * 3
        elements).  -> (Stats summed over the devices, gather + un-interleave ms)."""
        th, tw = params.y1 - params.y0, params.x1 - params.x0
        if out_tensor.numel() < th * tw * 3 or str(out_tensor.dtype) != "torch.float32" or not out_tensor.is_cuda or not out_tensor.is_contiguous():
            raise TrtError("render_into: need a contiguous float32 device tensor with tile rows * tile_w * 3 elements")
        st = Stats()
        gms = C.c_double(0.0)
        rc = self._lib.trt_group_render_device(self._g, C.byref(params), C.c_void_p(out_tensor.data_ptr()), C.byref(st), C.byref(gms))
        if rc != 0:
            raise TrtError(f"trt_group_render_device failed ({rc}): {self._lib.trt_last_error().decode()}")
        return st, gms.value

    def close(self):
        if getattr(self, "_g", None):
            self._lib.trt_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tonemap(image):
    """imshow()'s transfer: (uchar) clamp(pow(x, 1/2.2f) * 255, 0, 255) (main.cpp:34-36)."""
    lib = _abi.load_host()
    img = np.ascontiguousarray(image, dtype=np.float32)
    h, w = img.shape[0], img.shape[1]
    out = np.empty((h, w, 3), np.uint8)
    if lib.trth_tonemap(img.ctypes.data_as(C.POINTER(C.c_float)), w, h, out.ctypes.data_as(C.POINTER(C.c_uint8))) != 0:
        raise TrtError(lib.trth_last_error().decode())
    return out


def imshow(image, path):
    """Writes the linear image as <path> (PNG, stored deflate) after the reference's gamma."""
    lib = _abi.load_host()
    img = np.ascontiguousarray(image, dtype=np.float32)
    h, w = img.shape[0], img.shape[1]
    if lib.trth_write_png(os.fsencode(path), w, h, img.ctypes.data_as(C.POINTER(C.c_float))) != 0:
        raise TrtError(lib.trth_last_error().decode())
