#!/bin/bash
# node-kind A/B on config 5's scene (10 M triangles at 3840x2160, the one beyond the Infinity Cache)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/ab
for nk in 0 1; do
  TRT_NODE_KIND=$nk python bench.py --scene blob --tris 10000000 --width 3840 --height 2160 --spp 16 --steps 2 --no-cpu-baseline > gpurun_out/ab/blob10m_nk$nk.json 2> gpurun_out/ab/blob10m_nk$nk.err || echo "nk$nk failed"
  python - gpurun_out/ab/blob10m_nk$nk.json nk$nk <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("blob10m", sys.argv[2], d["value"], "Mrays/s", d["ms_per_step"], "ms", {k: v["ms_per_step"] for k, v in d["kernels_rank0"].items() if v["ms_per_step"]}, d["simd_utilisation_traversal"], flush=True)
PY
done
