#!/bin/bash
# A/B of the node kind (TRT_NODE_KIND=0 exact 128-B 4-wide nodes, 1 compressed 80-B 8-wide nodes: trt_oct.h) on the per-lane-traversal scenes.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/ab
run() { # tag envs args...
  tag=$1; envs=$2; shift 2
  env $envs python bench.py "$@" --no-cpu-baseline --no-extra --no-overlap-extra > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err || echo "$tag failed"
  python - gpurun_out/ab/$tag.json "$tag" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2].ljust(28), d["value"], "Mrays/s", d["ms_per_step"], "ms", {k: v["ms_per_step"] for k, v in d["kernels_rank0"].items() if v["ms_per_step"]}, d["simd_utilisation_traversal"], flush=True)
except Exception as e:
    print(sys.argv[2], "no result", e, flush=True)
PY
}
for nk in 0 1; do
  run veach_nk$nk TRT_NODE_KIND=$nk --scene veach-mis --steps 2
  run stair_nk$nk TRT_NODE_KIND=$nk --scene staircase --spp 64 --steps 2
  run soup_nk$nk TRT_NODE_KIND=$nk --scene soup --spp 16 --steps 2
  run blob2m_nk$nk TRT_NODE_KIND=$nk --scene blob --tris 2000000 --spp 64 --steps 2
  run blob10m_nk$nk TRT_NODE_KIND=$nk --scene blob --tris 10000000 --width 3840 --height 2160 --spp 16 --steps 2
done
