"""Per-kernel times of renders cut off after 1, 2 and 3 path vertices (max_depth): the cost of the first launches of each
kernel, one at a time.  usage: python tools/depth_times.py [scene] [spp]      (TRT_HIP_LIB selects an A/B build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tinyraytracing_amd as T
name = sys.argv[1] if len(sys.argv) > 1 else "back"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
s = T.Scene.named(name, 1920, 1080)
r = T.Renderer(s, 0)
out = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
tag = os.path.basename(os.environ.get("TRT_HIP_LIB", "default"))
for depth in (1, 2, 3):
    p = T.make_params(1920, 1080, spp, T.SEED_BACK, flags=T.TRT_FLAG_TIMING, max_depth=depth)
    r.render_into(p, out)
    best = None
    for _ in range(3):
        st = r.render_into(p, out)
        k = [st.kernel_ms[i] for i in range(len(T.KERNEL_NAMES))]
        best = k if best is None else [min(a, b) for a, b in zip(best, k)]
    print(f"{tag:28s} {name} depth<={depth}  " + " ".join(f"{T.KERNEL_NAMES[i]}={best[i]:.2f}" for i in range(len(best)) if best[i] > 0) + f"  rays {st.rays}", flush=True)
