"""ctypes mirror of include/trt.h and include/trt_host.h, and the library loaders.

The product path is the HIP library: `load_hip()` raises if
tinyraytracing_amd/lib/libtrt_hip.so is missing or cannot be loaded — there is
no CPU fallback anywhere in this package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

TRT_ABI_VERSION = 4
TRT_INF = 114514.0
TRT_FLAG_TIMING = 1
TRT_FLAG_COUNT = 2
TRT_FLAG_OVERLAP = 4
TRT_FLAG_FIXED_NEE = 8
TRT_FLAG_FIXED_PIXELS = 16
TRT_FLAG_RAY_OFFSET = 32
TRT_FLAG_SPECULAR_KS = 64
TRT_MAX_KERNELS = 8
KERNEL_NAMES = ["gen_primary", "trace_closest", "shade", "trace_shadow", "resolve", "tail"]

c_float3 = C.c_float * 3


class BvhNode(C.Structure):
    _fields_ = [("lo0", c_float3), ("hi0", c_float3), ("lo1", c_float3), ("hi1", c_float3),
                ("child0", C.c_uint32), ("child1", C.c_uint32), ("reserved", C.c_uint32 * 2)]


class Material(C.Structure):
    _fields_ = [("Kd", c_float3), ("Ks", c_float3), ("Tr", c_float3), ("Ns", C.c_float), ("Ni", C.c_float),
                ("radiance", c_float3), ("is_emissive", C.c_int32), ("tex", C.c_int32)]


class Light(C.Structure):
    _fields_ = [("mat", C.c_int32), ("radiance", c_float3), ("area", C.c_float),
                ("tri_first", C.c_uint32), ("tri_count", C.c_uint32)]


class LightTri(C.Structure):
    _fields_ = [("v", c_float3 * 3), ("vn", c_float3 * 3), ("cum_area", C.c_float)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.POINTER(C.c_uint8))]


class Camera(C.Structure):
    _fields_ = [("eye", c_float3), ("lower_left_corner", c_float3), ("horizontal", c_float3), ("vertical", c_float3)]


class SceneFlat(C.Structure):
    _fields_ = [("n_tris", C.c_uint32), ("tri_v", C.POINTER(C.c_float)), ("tri_vn", C.POINTER(C.c_float)),
                ("tri_vt", C.POINTER(C.c_float)), ("tri_mat", C.POINTER(C.c_int32)),
                ("n_nodes", C.c_uint32), ("nodes", C.POINTER(BvhNode)), ("bvh_depth", C.c_uint32),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(Material)),
                ("n_lights", C.c_uint32), ("lights", C.POINTER(Light)),
                ("n_light_tris", C.c_uint32), ("light_tris", C.POINTER(LightTri)),
                ("n_textures", C.c_uint32), ("textures", C.POINTER(Texture)),
                ("camera", Camera)]


class Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("seed", C.c_uint32),
                ("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32),
                ("row_block", C.c_int32), ("row_mod", C.c_int32), ("row_rem", C.c_int32),
                ("max_depth", C.c_int32), ("flags", C.c_uint32), ("mem_budget", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("rays_camera", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_indirect", C.c_uint64),
                ("shaded_hits", C.c_uint64), ("inner_visits", C.c_uint64 * 2), ("tri_tests", C.c_uint64 * 2), ("wave_steps", C.c_uint64 * 2),
                ("launches", C.c_uint64 * TRT_MAX_KERNELS), ("kernel_ms", C.c_double * TRT_MAX_KERNELS),
                ("render_ms", C.c_double), ("passes", C.c_uint32), ("max_bounces", C.c_uint32),
                ("rows_rendered", C.c_uint64), ("inner_node_bytes", C.c_uint32), ("redo_rays", C.c_uint32), ("lane_census", C.c_uint64 * 4)]

    @property
    def rays(self):
        return self.rays_camera + self.rays_shadow + self.rays_indirect


# the symbols include/trt.h declares (checked by tests/test_abi.py)
HIP_SYMBOLS = ["trt_rows_selected", "trt_create", "trt_render", "trt_render_device", "trt_render_samples", "trt_trace_closest",
               "trt_destroy", "trt_last_error", "trt_abi_version", "trt_group_create", "trt_group_render", "trt_group_render_device", "trt_group_size", "trt_group_destroy"]
BUILD_SYMBOLS = ["trt_build_lbvh", "trt_build_last_error"]
HOST_SYMBOLS = ["trth_scene_load", "trth_scene_load_opts", "trth_scene_drop_tris", "trth_scene_add_soup", "trth_scene_add_blob",
                "trth_scene_build", "trth_scene_vertices", "trth_scene_adopt_bvh", "trth_scene_flat", "trth_scene_info", "trth_scene_light_area",
                "trth_scene_material_name", "trth_scene_free", "trth_tonemap", "trth_write_png",
                "trth_write_png_bytes", "trth_decode_jpeg", "trth_decode_png", "trth_abi_sizes", "trth_last_error"]

_hip = None
_host = None
_build = None


def load_build():
    """Loads the GPU BVH builder (include/trt_build.h).  A library of its own: the render path does not need it."""
    global _build
    if _build is not None:
        return _build
    _bind_to_torch_hip_runtime()
    path = os.path.join(LIB_DIR, "libtrt_lbvh.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make lbvh`")
    lib = C.CDLL(path)
    lib.trt_build_last_error.restype = C.c_char_p
    lib.trt_build_lbvh.argtypes = [C.POINTER(C.c_float), C.c_uint32, C.c_int, C.c_int, C.POINTER(BvhNode), C.c_uint32, C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    _build = lib
    return lib


def load_host():
    global _host
    if _host is not None:
        return _host
    path = os.path.join(LIB_DIR, "libtrt_host.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` or `make`")
    lib = C.CDLL(path)
    lib.trth_last_error.restype = C.c_char_p
    lib.trth_scene_load.restype = C.c_void_p
    lib.trth_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    lib.trth_scene_load_opts.restype = C.c_void_p
    lib.trth_scene_load_opts.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
    lib.trth_scene_drop_tris.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    lib.trth_scene_add_soup.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
    lib.trth_scene_add_blob.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
    lib.trth_scene_build.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.trth_scene_vertices.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint64]
    lib.trth_scene_adopt_bvh.argtypes = [C.c_void_p, C.POINTER(BvhNode), C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32]
    lib.trth_scene_flat.restype = C.POINTER(SceneFlat)
    lib.trth_scene_flat.argtypes = [C.c_void_p]
    lib.trth_scene_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.trth_scene_light_area.restype = C.c_double
    lib.trth_scene_light_area.argtypes = [C.c_void_p, C.c_uint32]
    lib.trth_scene_material_name.restype = C.c_char_p
    lib.trth_scene_material_name.argtypes = [C.c_void_p, C.c_uint32]
    lib.trth_scene_free.argtypes = [C.c_void_p]
    lib.trth_scene_free.restype = None
    lib.trth_tonemap.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_uint8)]
    lib.trth_write_png.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.trth_abi_sizes.argtypes = [C.POINTER(C.c_int64)]
    lib.trth_decode_jpeg.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint8), C.c_uint64]
    lib.trth_decode_png.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint8), C.c_uint64]
    lib.trth_write_png_bytes.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
    _host = lib
    return lib


def _bind_to_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7) and load it through the unversioned name, so a process that had already loaded
    /opt/rocm's copy for libtrt_hip.so would end up with two runtimes and torch would then see no GPU.
    When a torch installation is present its copy is loaded first (by path, RTLD_GLOBAL, without
    importing torch); libtrt_hip.so's NEEDED libamdhip64.so.7 then resolves to it by SONAME, and torch
    later finds the same file again.  Without torch the system ROCm runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    C.CDLL(path, mode=C.RTLD_GLOBAL)
    return path


def load_hip():
    """Loads the HIP C-ABI library.  Raises loudly if it is absent: the hot path has no fallback."""
    global _hip
    if _hip is not None:
        return _hip
    _bind_to_torch_hip_runtime()
    path = os.environ.get("TRT_HIP_LIB") or os.path.join(LIB_DIR, "libtrt_hip.so")  # TRT_HIP_LIB: A/B tuning builds
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: the HIP extension is the only compute path; "
                           "build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make hip`")
    lib = C.CDLL(path)
    lib.trt_last_error.restype = C.c_char_p
    lib.trt_abi_version.restype = C.c_int
    lib.trt_rows_selected.argtypes = [C.POINTER(Params)]
    lib.trt_create.argtypes = [C.POINTER(SceneFlat), C.c_int, C.POINTER(C.c_void_p)]
    lib.trt_render.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(C.c_float), C.POINTER(Stats)]
    lib.trt_render_device.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    lib.trt_render_samples.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(Stats)]
    lib.trt_trace_closest.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                      C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(Stats)]
    lib.trt_destroy.argtypes = [C.c_void_p]
    lib.trt_destroy.restype = None
    lib.trt_group_create.argtypes = [C.POINTER(SceneFlat), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
    lib.trt_group_render.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(C.c_float), C.POINTER(Stats), C.POINTER(C.c_double)]
    lib.trt_group_render_device.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.POINTER(Stats), C.POINTER(C.c_double)]
    lib.trt_group_size.argtypes = [C.c_void_p]
    lib.trt_group_destroy.argtypes = [C.c_void_p]
    lib.trt_group_destroy.restype = None
    if lib.trt_abi_version() != TRT_ABI_VERSION:
        raise RuntimeError("libtrt_hip.so ABI version mismatch")
    _hip = lib
    return lib
