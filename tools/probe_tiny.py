import sys, os
sys.path.insert(0, os.getcwd())
import torch, tinyraytracing_amd as T
s = T.Scene.named("back", 1920, 1080)
r = T.Renderer(s, 0)
out = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
p = T.make_params(1920, 1080, 64, 7, flags=T.TRT_FLAG_TIMING)
r.render_into(p, out); st = r.render_into(p, out); st2 = r.render_into(p, out)
print(os.environ.get("TRT_HIP_LIB", "default")[-22:], "redo", st.redo_rays, {T.KERNEL_NAMES[i]: round(min(st.kernel_ms[i], st2.kernel_ms[i]), 2) for i in (1, 2, 3)}, flush=True)
