#!/usr/bin/env python3
"""Dumps the tree the GPU builder gives a shipped scene (nodes + triangle order) so that it can be studied on a machine without a GPU
(tools/lbvh_study.py: the same scene adopts it through trth_scene_adopt_bvh and the CPU build of the device code walks it).  GPU box.
usage: tools/lbvh_dump.py OUT.npz scene [leaf] [TRT_LBVH_CLUSTER]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinyraytracing_amd as T  # noqa: E402
from tinyraytracing_amd import _abi  # noqa: E402


def main():
    out, name = sys.argv[1], sys.argv[2]
    leaf = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    if len(sys.argv) > 4:
        os.environ["TRT_LBVH_CLUSTER"] = sys.argv[4]
    d = os.path.join(T.SCENES_DIR, name)
    s = T.Scene.load(os.path.join(d, name + ".xml"), os.path.join(d, name + ".obj"), os.path.join(d, name + ".mtl"), d, 64, 36)
    n = s.info["n_triangles"]
    v = np.empty(n * 9, np.float32)
    s._check(s._lib.trth_scene_vertices(s._h, v.ctypes.data_as(C.POINTER(C.c_float)), v.size))
    lib = _abi.load_build()
    cap = max(n, 2) - 1
    node_bytes = np.zeros(cap * C.sizeof(_abi.BvhNode), np.uint8)
    nodes = C.cast(node_bytes.ctypes.data, C.POINTER(_abi.BvhNode))
    order = np.empty(n, np.uint32)
    n_nodes, depth = C.c_uint32(0), C.c_uint32(0)
    ms = (C.c_double * 2)()
    rc = lib.trt_build_lbvh(v.ctypes.data_as(C.POINTER(C.c_float)), n, leaf, 0, nodes, cap, C.byref(n_nodes), order.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(depth), ms)
    assert rc == 0, lib.trt_build_last_error()
    np.savez_compressed(out, nodes=node_bytes[: n_nodes.value * C.sizeof(_abi.BvhNode)], order=order, n_nodes=n_nodes.value, depth=depth.value, leaf=leaf)
    print(f"{name}: {n} triangles, {n_nodes.value} nodes, depth {depth.value} -> {out}")


if __name__ == "__main__":
    main()
