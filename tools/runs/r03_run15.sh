#!/bin/bash
# round 3, run 15: rays per second on the host SAH tree, the LBVH with SAH top (cluster sizes) and the plain radix tree, per kernel (bench.py, 64 spp, 2 steps)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/r03
run() { # tag envs args...
  tag=$1; envs=$2; shift 2
  env $envs timeout -k 10 400 python bench.py "$@" --no-cpu-baseline --no-extra --no-overlap-extra > gpurun_out/r03/$tag.json 2> gpurun_out/r03/$tag.err || echo "$tag failed"
  python - gpurun_out/r03/$tag.json "$tag" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = {a: b["ms_per_step"] for a, b in d["kernels_rank0"].items() if b["ms_per_step"]}
    u = d["simd_utilisation_traversal"]
    print(f'{sys.argv[2]:30s} {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:9.2f} ms  closest {k.get("trace_closest", 0):8.2f} shade {k.get("shade", 0):7.2f} shadow {k.get("trace_shadow", 0):8.2f} tail {k.get("tail", 0):6.2f} | visits {u["visits_per_ray"]} tests {u["tri_tests_per_ray"]}', flush=True)
except Exception as e:
    print(sys.argv[2], "no result", e, flush=True)
PY
}
for sc in "blob10m_4k --scene blob --tris 10000000 --width 3840 --height 2160" "blob2m --scene blob --tris 2000000" "soup --scene soup" "stair --scene staircase"; do
  set -- $sc; sc_tag=$1; shift
  run s15_${sc_tag}_sah X=1 "$@" --spp 64 --steps 2 --builder auto
  for c in 2048 512 8192 0; do run s15_${sc_tag}_lbvh_c$c TRT_LBVH_CLUSTER=$c "$@" --spp 64 --steps 2 --builder lbvh; done
done
