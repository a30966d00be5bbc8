#!/usr/bin/env python3
"""CPU study of the GPU builder's tree quality on the small shipped scenes (round 4, VERDICT r03 task 8; profiles/r04_lbvh_quality.txt (3)): where do the
extra node visits of staircase / veach-mis come from?  Needs no GPU: a tree dumped on the GPU box (tools/lbvh_dump.py) is adopted by the same scene here
(trth_scene_adopt_bvh), the CPU build of the device code (tests/hostsim) walks its 8-wide collapse and counts node visits / triangle tests per ray; a Python
emulation of the builder (Morton codes of the box centres, radix splits, leaves of <= 2, clusters, sweep SAH above) reproduces the dumped tree's counts, and
variants of it are counted the same way.

usage: tools/lbvh_study.py scene [DUMP.npz]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hostsim_lib as H  # noqa: E402
import raygen  # noqa: E402
import tinyraytracing_amd as T  # noqa: E402
from tinyraytracing_amd._abi import BvhNode  # noqa: E402

LEAF = 0x80000000
PAD = 0.001
sys.setrecursionlimit(100000)


def load_unbuilt(name):
    d = os.path.join(T.SCENES_DIR, name)
    return T.Scene.load(os.path.join(d, name + ".xml"), os.path.join(d, name + ".obj"), os.path.join(d, name + ".mtl"), d, 160, 90)


def vertices(s):
    n = s.info["n_triangles"]
    v = np.empty(n * 9, np.float32)
    s._check(s._lib.trth_scene_vertices(s._h, v.ctypes.data_as(C.POINTER(C.c_float)), v.size))
    return v.reshape(n, 3, 3)


def adopt(s, nodes, order, depth):
    if isinstance(nodes, np.ndarray):  # raw bytes of a dump
        buf = np.ascontiguousarray(nodes)
        s._check(s._lib.trth_scene_adopt_bvh(s._h, C.cast(buf.ctypes.data, C.POINTER(BvhNode)), len(buf) // C.sizeof(BvhNode), order.ctypes.data_as(C.POINTER(C.c_uint32)), depth))
    else:
        arr = (BvhNode * len(nodes))(*nodes)
        s._check(s._lib.trth_scene_adopt_bvh(s._h, arr, len(nodes), order.ctypes.data_as(C.POINTER(C.c_uint32)), depth))
    s._built = True


def half_area(lo, hi):
    e = np.maximum(hi - lo, 0.0)
    return e[..., 0] * e[..., 1] + e[..., 1] * e[..., 2] + e[..., 2] * e[..., 0]


def measure(s, rays):
    old = H.set_node_kind(1)
    try:
        v, t = H.trace_counts(s.flat, 1, *rays)
    finally:
        H.set_node_kind(old)
    return float(v.mean()), float(t.mean())


# a subtree is (lo, hi, triangles, kind, payload, centre): kind 'leaf' -> payload = triangle ids; kind 'node' -> payload = (left, right)
def spread21(x):
    x = x.astype(np.uint64)
    for sh, m in ((32, 0x1F00000000FFFF), (16, 0x1F0000FF0000FF), (8, 0x100F00F00F00F00F), (4, 0x10C30C30C30C30C3), (2, 0x1249249249249249)):
        x = (x | (x << np.uint64(sh))) & np.uint64(m)
    return x


def morton(c):
    lo = c.min(0)
    ext = c.max(0) - lo
    sc = np.where(ext > 0, 2097152.0 / np.where(ext > 0, ext, 1), 0)
    q = np.clip((c - lo) * sc, 0, 2097151).astype(np.uint64)
    return (spread21(q[:, 0]) << np.uint64(2)) | (spread21(q[:, 1]) << np.uint64(1)) | spread21(q[:, 2])


class Study:
    def __init__(self, name):
        self.name = name
        host = T.Scene.named(name, 160, 90)
        lo_s, hi_s = raygen.scene_bounds(host)
        o1, d1 = raygen.primary_rays(host, 160, 90, step=2)
        o2, d2 = raygen.random_rays(30000, lo_s, hi_s, seed=5)
        self.rays = (np.vstack([o1, o2]), np.vstack([d1, d2]))
        self.v0, self.t0 = measure(host, self.rays)
        print(f"{name}: host SAH builder: {self.v0:.2f} node visits, {self.t0:.2f} triangle tests per ray")
        s = load_unbuilt(name)
        self.V = vertices(s).astype(np.float64)
        self.tlo, self.thi = self.V.min(1), self.V.max(1)
        self.vc = self.V.mean(1)  # vertex centroids (what the reference's and the host builder's SAH sort by)

    def report(self, tag, s):
        v, t = measure(s, self.rays)
        print(f"   {tag:72s} visits {v:6.2f} ({(v / self.v0 - 1) * 100:+5.1f} %)  tests {t:6.2f} ({(t / self.t0 - 1) * 100:+5.1f} %)", flush=True)

    def leaf_item(self, ids, by_centroid=True):
        ids = list(ids)
        lo, hi = self.tlo[ids].min(0), self.thi[ids].max(0)
        return (lo, hi, len(ids), "leaf", ids, self.vc[ids].mean(0) if by_centroid else 0.5 * (lo + hi))

    def join(self, L, R, by_centroid=True):
        lo, hi = np.minimum(L[0], R[0]), np.maximum(L[1], R[1])
        return (lo, hi, L[2] + R[2], "node", (L, R), (L[5] * L[2] + R[5] * R[2]) / (L[2] + R[2]) if by_centroid else 0.5 * (lo + hi))

    def sweep(self, items, leaf_stop, by_centroid=True):
        """exact sweep SAH over subtrees; single triangles are folded into leaves of <= leaf_stop"""
        if len(items) == 1:
            return items[0]
        if leaf_stop and len(items) <= leaf_stop and all(it[3] == "leaf" and it[2] == 1 for it in items):
            return self.leaf_item([i for it in items for i in it[4]], by_centroid)
        lo = np.array([it[0] for it in items]); hi = np.array([it[1] for it in items])
        cnt = np.array([it[2] for it in items]); cen = np.array([it[5] for it in items])
        best = None
        for ax in range(3):
            o = np.argsort(cen[:, ax], kind="stable")
            pl = np.minimum.accumulate(lo[o], 0); ph = np.maximum.accumulate(hi[o], 0)
            sl = np.minimum.accumulate(lo[o][::-1], 0)[::-1]; sh = np.maximum.accumulate(hi[o][::-1], 0)[::-1]
            cn = np.cumsum(cnt[o])
            cost = half_area(pl[:-1], ph[:-1]) * cn[:-1] + half_area(sl[1:], sh[1:]) * (cn[-1] - cn[:-1])
            k = int(np.argmin(cost))
            if best is None or cost[k] < best[0]:
                best = (cost[k], o, k + 1)
        _, o, k = best
        return self.join(self.sweep([items[i] for i in o[:k]], leaf_stop, by_centroid), self.sweep([items[i] for i in o[k:]], leaf_stop, by_centroid), by_centroid)

    def radix_clusters(self, codes, order, leaf, cluster):
        """the radix tree over the sorted codes, cut into its maximal subtrees of <= cluster triangles (each returned as a subtree with radix splits inside)"""
        def split(a, b):
            ca, cb = int(codes[a]), int(codes[b - 1])
            if ca == cb:
                return (a + b) // 2
            mask = 1 << ((ca ^ cb).bit_length() - 1)
            lo_, hi_ = a, b - 1
            while lo_ < hi_:
                m = (lo_ + hi_) // 2
                if int(codes[m]) & mask:
                    hi_ = m
                else:
                    lo_ = m + 1
            return lo_

        def build(a, b):
            if b - a <= leaf:
                return self.leaf_item(order[a:b])
            k = split(a, b)
            return self.join(build(a, k), build(k, b))
        out = []

        def cut(a, b):
            if b - a <= cluster:
                out.append(build(a, b))
                return
            k = split(a, b)
            cut(a, k)
            cut(k, b)
        cut(0, len(codes))
        return out

    def run(self, tag, root):
        nodes, order = [], []

        def ref(t):
            if t[3] == "leaf":
                f = len(order)
                order.extend(t[4])
                return LEAF | (len(t[4]) << 27) | f, 0
            me = len(nodes)
            nodes.append(None)
            (r0, d0), (r1, d1) = ref(t[4][0]), ref(t[4][1])
            nd = BvhNode()
            for x in range(3):
                nd.lo0[x] = t[4][0][0][x] - PAD; nd.hi0[x] = t[4][0][1][x] + PAD
                nd.lo1[x] = t[4][1][0][x] - PAD; nd.hi1[x] = t[4][1][1][x] + PAD
            nd.child0, nd.child1 = r0, r1
            nodes[me] = nd
            return me, 1 + max(d0, d1)
        _, depth = ref(root)
        sc = load_unbuilt(self.name)
        adopt(sc, nodes, np.array(order, np.uint32), depth)
        self.report(tag, sc)


def leaves(it):
    return [it] if it[3] == "leaf" else leaves(it[4][0]) + leaves(it[4][1])


def main():
    name = sys.argv[1]
    st = Study(name)
    n = len(st.V)
    if len(sys.argv) > 2:
        dump = np.load(sys.argv[2])
        s = load_unbuilt(name)
        adopt(s, dump["nodes"], dump["order"].astype(np.uint32), int(dump["depth"]))
        st.report("the GPU builder's tree as dumped", s)
    singles = [st.leaf_item([i]) for i in range(n)]
    st.run("full sweep SAH in Python, ordered by vertex centroid (the host builder's and the reference's)", st.sweep(singles, 2))
    st.run("full sweep SAH in Python, ordered by box centre", st.sweep([st.leaf_item([i], False) for i in range(n)], 2, False))
    codes = morton(0.5 * (st.tlo + st.thi))
    o = np.argsort(codes, kind="stable")
    codes = codes[o]
    c_vc = morton(st.vc)
    o_vc = np.argsort(c_vc, kind="stable")
    big = max(256, n // 64)
    st.run(f"emulation of the GPU builder: clusters <= {big}, SAH over them", st.sweep(st.radix_clusters(codes, o, 2, big), 0, False))
    st.run(f"  ... Morton codes from the vertex centroid instead of the box centre", st.sweep(st.radix_clusters(c_vc[o_vc], o_vc, 2, big), 0, False))
    for cl in (32, 16, 8, 4):
        st.run(f"radix clusters <= {cl} under a sweep SAH ordered by centroid", st.sweep(st.radix_clusters(codes, o, 2, cl), 0))
    st.run("the Morton PAIRS kept as leaves, everything above by that SAH", st.sweep([l for c in st.radix_clusters(codes, o, 2, big) for l in leaves(c)], 0))
    for cl in (32, 16, 8):
        subs = [st.sweep([st.leaf_item([i]) for l in leaves(c) for i in l[4]], 2) for c in st.radix_clusters(codes, o, 2, cl)]
        st.run(f"clusters <= {cl} REBUILT inside by SAH, under that SAH", st.sweep(subs, 0))


if __name__ == "__main__":
    main()
