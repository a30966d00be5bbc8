"""How long trt_create takes on the 10 M-triangle mesh (validation, collapses, leaf boxes, nesting check, plane filter, uploads)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import tinyraytracing_amd as T
t = time.time(); s = T.Scene.named("blob", 640, 360, n=10000000); print("load + build s", round(time.time() - t, 2), s.info["n_triangles"], flush=True)
os.environ["TRT_DEBUG"] = "1"
t = time.time(); r = T.Renderer(s, 0); print("trt_create s", round(time.time() - t, 2), flush=True)
r.close()
