#!/bin/bash
# Round 3, GPU call 3: the oct driver with queue counters + parked results: parity, then refill batch / grid sweeps.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/r03/ab3
export PYTHONUNBUFFERED=1
echo "== parity subset"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "node_kind or compressed or degenerate or grazing or redo or unpadded or golden or incoherent or soup or tiny or image_matches or chunking or overlapped or interleave or fixed_nee" 2>&1 | tail -5
test ${PIPESTATUS[0]} -eq 0 || exit 1
run() { # tag envs args...
  tag=$1; envs=$2; shift 2
  env $envs timeout -k 10 400 python bench.py "$@" --no-cpu-baseline --no-extra --no-overlap-extra > gpurun_out/r03/ab3/$tag.json 2> gpurun_out/r03/ab3/$tag.err || echo "$tag failed"
  python - gpurun_out/r03/ab3/$tag.json "$tag" <<'PY'
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
    soup) echo "--scene soup --spp 16 --steps 2";;
    blob2m) echo "--scene blob --tris 2000000 --spp 64 --steps 2";;
    blob10m) echo "--scene blob --tris 10000000 --width 3840 --height 2160 --spp 16 --steps 2";;
  esac
}
for sc in veach stair soup blob2m; do
  a=$(scene_args $sc)
  run ${sc}_default "TRT_NODE_KIND=1" $a
  for rf in 16 24 32; do run ${sc}_refill$rf "TRT_NODE_KIND=1 TRT_REFILL_MIN=$rf" $a; done
done
for sc in veach stair; do
  a=$(scene_args $sc)
  for ob in 1024 1536; do run ${sc}_blocks$ob "TRT_NODE_KIND=1 TRT_OCT_BLOCKS=$ob" $a; done
done
run blob10m_default "TRT_NODE_KIND=1" $(scene_args blob10m)
CENSUS_REFILLS=default,16 timeout -k 10 300 python tools/lane_census.py veach-mis:64 staircase:32 soup:16 blob:32 2>&1 | grep -v amdgpu.ids
