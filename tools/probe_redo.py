import sys, os
sys.path.insert(0, os.getcwd())
import torch, tinyraytracing_amd as T
for name, spp in (("back", 64), ("soup", 16), ("staircase", 16)):
    s = T.Scene.named(name, 1920, 1080)
    r = T.Renderer(s, 0)
    out = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
    p = T.make_params(1920, 1080, spp, 7, flags=T.TRT_FLAG_TIMING)
    r.render_into(p, out); st = r.render_into(p, out)
    print(name, "rays", st.rays, "redo_rays", st.redo_rays, "ratio", st.redo_rays / st.rays, {T.KERNEL_NAMES[i]: round(st.kernel_ms[i], 2) for i in range(len(T.KERNEL_NAMES))}, flush=True)
    r.close()
