"""Wall time of a render with and without TRT_FLAG_TIMING (hipEvents around every kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tinyraytracing_amd as T
name = sys.argv[1] if len(sys.argv) > 1 else "back"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
world = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # > 1: time the stripes of rank 0 of `world` GPUs
s = T.Scene.named(name, 1920, 1080)
r = T.Renderer(s, 0)
out = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
for label, flags in (("no flags", 0), ("TIMING", T.TRT_FLAG_TIMING), ("OVERLAP", T.TRT_FLAG_OVERLAP), ("no flags", 0), ("TIMING", T.TRT_FLAG_TIMING)):
    p = T.make_params(1920, 1080, spp, T.SEED_BACK, flags=flags, rows=(8, world, 0) if world > 1 else None)
    r.render_into(p, out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        st = r.render_into(p, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print(f"{name} 1/{world} {label:9s} {dt*1e3:8.2f} ms/step  {st.rays/dt/1e6:9.1f} Mrays/s  device render_ms {st.render_ms:.2f}" + (f"  kernels {sum(st.kernel_ms):.2f} ms in {sum(st.launches)} launches, {st.passes} passes, deepest vertex {st.max_bounces}  " + " ".join(f"{T.KERNEL_NAMES[k]}={st.kernel_ms[k]:.2f}" for k in range(len(T.KERNEL_NAMES)) if st.kernel_ms[k] > 0) if flags & T.TRT_FLAG_TIMING else ""), flush=True)
