#!/usr/bin/env python3
"""Would grouping rays by phase across the waves of a block pay?  A step-count simulation on the CPU, no GPU needed.

tests/hostsim (the device code compiled for the CPU) writes, for camera rays and for parity-mode shadow rays of a scene, the tape of steps the
oct driver takes per ray (node step / leaf step of 1-2 triangles).  Two schedulers consume the same tapes in queue order:
  (a) the shipped driver: 64 lanes per wave, a static slice per wave, refill in batches (refill_min lanes free), every iteration the step kind
      that more lanes wait for;
  (b) a pool of 256 ray slots per block (state in LDS): every round the slots are listed by phase and dealt to the block's four waves, leaf
      entries first, so at most one wave per round holds both kinds; finished slots cost nothing until 64 of them are refilled together.
Instruction volume = steps x the static VALU counts of the shipped kernel (node step 275, triangle pass 76, refill + store 250 per executing
wave, loop overhead 30 per iteration) with 60 more per round and wave for (b): classification, block-wide prefix, state to and from LDS.
The simulation is generous to (b): no loss of occupancy (its LDS footprint costs one to two waves per SIMD), no barrier stalls, no bank conflicts.
usage: tools/pool_sim.py [scene ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hostsim_lib as H  # noqa: E402
import tinyraytracing_amd as T  # noqa: E402

NODE, TRI, REFILL, LOOP, POOL_EXTRA = 275, 76, 250, 30, 60
CAP = 96


def tapes(s, o, d, t_init=None, light=-1):
    n = len(o)
    lib = H.lib()
    lib.hostsim_oct_step_tape.argtypes = [C.c_void_p, C.c_uint64, H.fp, H.fp, H.fp, C.c_int, C.c_uint32, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)]
    tape = np.zeros((n, CAP), np.uint8)
    ln = np.zeros(n, np.uint32)
    o = np.ascontiguousarray(o, np.float32)
    d = np.ascontiguousarray(d, np.float32)
    ti = None if t_init is None else np.ascontiguousarray(t_init, np.float32)
    rc = lib.hostsim_oct_step_tape(C.cast(s.flat, C.c_void_p), n, o.ctypes.data_as(H.fp), d.ctypes.data_as(H.fp), ti.ctypes.data_as(H.fp) if ti is not None else None,
                                   light, CAP, tape.ctypes.data_as(C.POINTER(C.c_uint8)), ln.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert rc == 0
    return tape, np.minimum(ln, CAP)


def camera_rays(s, w, h):
    c = s.flat.contents.camera
    eye, llc, hor, ver = (np.array(list(x), np.float32) for x in (c.eye, c.lower_left_corner, c.horizontal, c.vertical))
    rng = np.random.default_rng(1)
    u = (np.arange(w, dtype=np.float32)[None, :] + rng.random((h, w), dtype=np.float32)) / w
    v = 1.0 - (np.arange(h, dtype=np.float32)[:, None] + rng.random((h, w), dtype=np.float32)) / h
    d = llc + u[..., None] * hor + v[..., None] * ver - eye
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32).reshape(-1, 3)
    return np.broadcast_to(eye, d.shape).copy(), d


def shadow_rays(s, o, d, light):
    """from the camera rays' hit points towards uniform points on the triangles of `light`, with the search hint of the parity mode"""
    t, tri, _, _ = H.trace(s.flat, o, d)
    hit = tri >= 0
    P = o[hit] + t[hit, None] * d[hit]
    f = s.flat.contents
    L = f.lights[light]
    tris = np.array([[list(f.light_tris[L.tri_first + k].v[j]) for j in range(3)] for k in range(L.tri_count)], np.float32)
    rng = np.random.default_rng(2)
    k = rng.integers(0, len(tris), len(P))
    a, b = rng.random(len(P), dtype=np.float32), rng.random(len(P), dtype=np.float32)
    flip = a + b > 1
    a[flip], b[flip] = 1 - a[flip], 1 - b[flip]
    Q = tris[k, 0] + a[:, None] * (tris[k, 1] - tris[k, 0]) + b[:, None] * (tris[k, 2] - tris[k, 0])
    w = Q - P
    dist = np.linalg.norm(w, axis=1)
    ok = dist > 1e-4
    return P[ok], (w[ok] / dist[ok, None]).astype(np.float32), (1.001 * dist[ok]).astype(np.float32)


def sim_waves(tape, ln, refill_min=48, rays_per_wave=1024):
    """(a): returns wave-level counts"""
    n = len(ln)
    st = dict(node_steps=0, node_lanes=0, leaf_steps=0, tri_passes=0, leaf_lanes=0, refills=0, iters=0)
    for w0 in range(0, n, rays_per_wave):
        nxt, end = w0, min(n, w0 + rays_per_wave)
        ray = np.full(64, -1, np.int64)
        ptr = np.zeros(64, np.int64)
        while True:
            has = ray >= 0
            working = has & (ptr < ln[np.maximum(ray, 0)])
            free = ~working
            if not working.any() or (has & free).any() and free.sum() >= refill_min or (nxt < end and not has.any()):
                if nxt < end or (has & free).any():
                    st["refills"] += 1
                take = np.flatnonzero(free)[: max(0, end - nxt)]
                ray[free] = -1
                ray[take] = np.arange(nxt, nxt + len(take))
                ptr[take] = 0
                nxt += len(take)
                has = ray >= 0
                working = has & (ptr < ln[np.maximum(ray, 0)])
                if not working.any():
                    if nxt >= end:
                        break
                    continue
            tok = tape[np.maximum(ray, 0), np.minimum(ptr, CAP - 1)]
            is_node = working & (tok == 0)
            is_leaf = working & (tok != 0)
            st["iters"] += 1
            if is_node.sum() >= is_leaf.sum():
                st["node_steps"] += 1
                st["node_lanes"] += int(is_node.sum())
                ptr[is_node] += 1
            else:
                st["leaf_steps"] += 1
                st["tri_passes"] += int(tok[is_leaf].max())
                st["leaf_lanes"] += int(tok[is_leaf].sum())
                ptr[is_leaf] += 1
    return st


def sim_pool(tape, ln, slots=256, refill_at=64, rays_per_block=4096):
    """(b)"""
    n = len(ln)
    st = dict(node_steps=0, node_lanes=0, leaf_steps=0, tri_passes=0, leaf_lanes=0, refills=0, iters=0)
    for b0 in range(0, n, rays_per_block):
        nxt, end = b0, min(n, b0 + rays_per_block)
        ray = np.full(slots, -1, np.int64)
        ptr = np.zeros(slots, np.int64)
        while True:
            has = ray >= 0
            working = has & (ptr < ln[np.maximum(ray, 0)])
            free = ~working
            if (not working.any()) or (free.sum() >= refill_at and (nxt < end or (has & free).any())):
                take = np.flatnonzero(free)[: max(0, end - nxt)]
                touched = np.flatnonzero(has & free)
                waves = set((np.concatenate([take, touched]) // 64).tolist())
                st["refills"] += len(waves)  # the owner threads' waves run the store / fetch code
                ray[free] = -1
                ray[take] = np.arange(nxt, nxt + len(take))
                ptr[take] = 0
                nxt += len(take)
                has = ray >= 0
                working = has & (ptr < ln[np.maximum(ray, 0)])
                if not working.any():
                    if nxt >= end:
                        break
                    continue
            tok = tape[np.maximum(ray, 0), np.minimum(ptr, CAP - 1)]
            leaf = np.flatnonzero(working & (tok != 0))
            node = np.flatnonzero(working & (tok == 0))
            # entries dealt to the virtual lanes: leaf entries first
            nL, nN = len(leaf), len(node)
            for wv in range((nL + nN + 63) // 64):
                lo, hi = wv * 64, min(nL + nN, wv * 64 + 64)
                l_here = leaf[lo:min(hi, nL)] if lo < nL else leaf[:0]
                n_here = hi - max(lo, nL) if hi > nL else 0
                st["iters"] += 1
                if len(l_here):
                    st["leaf_steps"] += 1
                    st["tri_passes"] += int(tok[l_here].max())
                    st["leaf_lanes"] += int(tok[l_here].sum())
                if n_here > 0:
                    st["node_steps"] += 1
                    st["node_lanes"] += n_here
            ptr[leaf] += 1
            ptr[node] += 1
    return st


def volume(st, per_iter):
    return st["node_steps"] * NODE + st["tri_passes"] * TRI + st["refills"] * REFILL + st["iters"] * per_iter


def main():
    for name in sys.argv[1:] or ["veach-mis", "staircase"]:
        s = T.Scene.named(name, 640, 360)
        o, d = camera_rays(s, 640, 360)
        sets = [("camera rays", tapes(s, o, d))]
        for light in range(min(s.flat.contents.n_lights, 2)):
            so, sd, st_ = shadow_rays(s, o, d, light)
            sets.append((f"shadow rays to light {light}", tapes(s, so, sd, st_, light)))
        for label, (tape, ln) in sets:
            n = len(ln)
            a, b = sim_waves(tape, ln), sim_pool(tape, ln)
            va, vb = volume(a, LOOP), volume(b, LOOP + POOL_EXTRA)
            print(f"{name:10s} {label:26s} {n:7d} rays, {ln.mean():5.1f} steps per ray | shipped: lanes/node step {a['node_lanes'] / max(64 * a['node_steps'], 1):.2f} tris/leaf step {a['leaf_lanes'] / max(64 * a['leaf_steps'], 1):.2f} "
                  f"{va / n * 64:6.0f} instr per 64 rays | pool: {b['node_lanes'] / max(64 * b['node_steps'], 1):.2f} / {b['leaf_lanes'] / max(64 * b['leaf_steps'], 1):.2f} {vb / n * 64:6.0f} "
                  f"({100 * (vb / va - 1):+.0f} %; with a free re-deal {100 * (volume(b, LOOP) / va - 1):+.0f} %)", flush=True)
        s.close()


if __name__ == "__main__":
    main()
