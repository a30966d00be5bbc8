"""Re-renders one configuration reported by tools/fuzz_parity.py on the default and the optional-path handles and compares with the oracle.
usage: python tools/repro_mismatch.py scene x0 y0 x1 y1 spp seed flags max_depth rb rm rr"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import tinyraytracing_amd as T
name = sys.argv[1]
x0, y0, x1, y1, spp, seed, flags, md, rb, rm, rr = [int(v, 0) for v in sys.argv[2:13]]
sizes = {"back": (257, 131), "veach-mis": (320, 180), "staircase": (192, 108), "soup": (160, 90)}
w, h = sizes[name]
sc = T.Scene.named(name, w, h, **({"n": 50000} if name == "soup" else {}))
p = T.make_params(w, h, spp, seed, tile=(x0, y0, x1, y1), rows=(rb, rm, rr) if rm > 1 else None, max_depth=md, flags=flags)
ref, ost = O.render(sc.flat, p)
for label, env in (("default", {}), ("exact 4-wide nodes, per lane", {"TRT_NODE_KIND": "0", "TRT_TRACE_IMPL": "3"}), ("8-wide compressed nodes, per lane", {"TRT_NODE_KIND": "1", "TRT_TRACE_IMPL": "3"})):
    os.environ.update(env)
    r = T.Renderer(sc, 0)
    for k in env: del os.environ[k]
    for fl in (flags, flags & ~T.TRT_FLAG_OVERLAP, flags & ~T.TRT_FLAG_COUNT):
        pp = T.make_params(w, h, spp, seed, tile=(x0, y0, x1, y1), rows=(rb, rm, rr) if rm > 1 else None, max_depth=md, flags=fl)
        img, st = r.render(pp)
        d = np.abs(img.astype(np.float64) - ref.astype(np.float64))
        bad = np.argwhere(d.max(-1) > 0)
        print(label, "flags", fl, "pixels differing", len(bad), "max abs", d.max(), "rays", (st.rays_camera, st.rays_shadow, st.rays_indirect), "oracle", (ost.rays_camera, ost.rays_shadow, ost.rays_indirect), "first", bad[:3].tolist(), flush=True)
