#!/bin/bash
# Round 3, GPU call 7: oct nodes at a stride of 128 B (one line per visit) against 80 B (packed).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/r03/ab7
export PYTHONUNBUFFERED=1
V=$root/tinyraytracing_amd/lib/variants
run() { # tag envs args...
  tag=$1; envs=$2; shift 2
  env $envs timeout -k 10 400 python bench.py "$@" --no-cpu-baseline --no-extra --no-overlap-extra > gpurun_out/r03/ab7/$tag.json 2> gpurun_out/r03/ab7/$tag.err || echo "$tag failed"
  python - gpurun_out/r03/ab7/$tag.json "$tag" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = {a: b["ms_per_step"] for a, b in d["kernels_rank0"].items() if b["ms_per_step"]}
    u = d["simd_utilisation_traversal"]
    print(f'{sys.argv[2]:30s} {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:9.2f} ms  closest {k.get("trace_closest", 0):8.2f} shade {k.get("shade", 0):7.2f} shadow {k.get("trace_shadow", 0):8.2f} tail {k.get("tail", 0):6.2f} | lanes {u["inner_steps"]}/{u["leaf_steps"]} visits {u["visits_per_ray"]} tests {u["tri_tests_per_ray"]}', flush=True)
except Exception as e:
    print(sys.argv[2], "no result", e, flush=True)
PY
}
scene_args() {
  case $1 in
    veach) echo "--scene veach-mis --steps 2";;
    stair) echo "--scene staircase --spp 64 --steps 2";;
    soup) echo "--scene soup --spp 64 --steps 2";;
    blob2m) echo "--scene blob --tris 2000000 --spp 64 --steps 2";;
    blob10m) echo "--scene blob --tris 10000000 --width 3840 --height 2160 --spp 64 --steps 1";;
  esac
}
for sc in soup blob10m blob2m stair veach; do
  a=$(scene_args $sc)
  run ${sc}_n80 "X=1" $a
  run ${sc}_n128 "TRT_HIP_LIB=$V/libtrt_hip_n128.so" $a
done
