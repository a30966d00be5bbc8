#!/bin/bash
# A/B of the wave driver (TRT_TRACE_IMPL=3 scheduler, 4 scheduler + postponed leaf) on the per-lane-traversal scenes
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/ab
run() { # tag envs args...
  tag=$1; envs=$2; shift 2
  env $envs python bench.py "$@" --no-cpu-baseline --no-extra > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err || echo "$tag failed"
  python - gpurun_out/ab/$tag.json "$tag" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2].ljust(28), d["value"], "Mrays/s", d["ms_per_step"], "ms", {k: v["ms_per_step"] for k, v in d["kernels_rank0"].items() if v["ms_per_step"]}, d["simd_utilisation_traversal"], flush=True)
except Exception as e:
    print(sys.argv[2], "no result", e, flush=True)
PY
}
for impl in 3 4; do
  run veach_i$impl TRT_TRACE_IMPL=$impl --scene veach-mis --steps 2
  run stair_i$impl TRT_TRACE_IMPL=$impl --scene staircase --spp 64 --steps 2
  run soup_i$impl TRT_TRACE_IMPL=$impl --scene soup --spp 16 --steps 2
  run blob2m_i$impl TRT_TRACE_IMPL=$impl --scene blob --tris 2000000 --spp 64 --steps 2
done
