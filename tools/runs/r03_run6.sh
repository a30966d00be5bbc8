#!/bin/bash
# Round 3, GPU call 6: eager refill (no batches, parked results) against the default; weights / batch / occupancy sweeps with two triangles per leaf step.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/r03/ab6
export PYTHONUNBUFFERED=1
V=$root/tinyraytracing_amd/lib/variants
echo "== parity of the eager build"
TRT_HIP_LIB=$V/libtrt_hip_eg7.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "node_kind or compressed or degenerate or grazing or redo or unpadded or golden or incoherent or soup or image_matches or fixed_nee or chunking or overlapped" 2>&1 | tail -4
test ${PIPESTATUS[0]} -eq 0 || exit 1
run() { # tag envs args...
  tag=$1; envs=$2; shift 2
  env $envs timeout -k 10 400 python bench.py "$@" --no-cpu-baseline --no-extra --no-overlap-extra > gpurun_out/r03/ab6/$tag.json 2> gpurun_out/r03/ab6/$tag.err || echo "$tag failed"
  python - gpurun_out/r03/ab6/$tag.json "$tag" <<'PY'
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
for sc in veach stair soup blob2m blob10m; do
  a=$(scene_args $sc)
  run ${sc}_default "X=1" $a
  for v in eg7 eg l8 w8; do run ${sc}_$v "TRT_HIP_LIB=$V/libtrt_hip_$v.so" $a; done
done
for sc in veach stair; do
  a=$(scene_args $sc)
  for w in 2:3 1:2 3:4; do run ${sc}_sched$w "TRT_SCHED_W=$w" $a; done
  for rf in 32 40 56; do run ${sc}_refill$rf "TRT_REFILL_MIN=$rf" $a; done
  for w in 1:1 2:3 1:2; do run ${sc}_eg7_sched$w "TRT_HIP_LIB=$V/libtrt_hip_eg7.so TRT_SCHED_W=$w" $a; done
done
echo "== back per lane on the oct nodes (leaves of 2) against the uniform walk"
run back_uniform "X=1" --steps 5
run back_oct "TRT_TRACE_IMPL=3 TRT_NODE_KIND=1" --steps 5 --leaf 2
CENSUS_REFILLS=default timeout -k 10 300 python tools/lane_census.py veach-mis:64 staircase:32 soup:16 blob:32 2>&1 | grep -v amdgpu.ids
TRT_HIP_LIB=$V/libtrt_hip_eg7.so CENSUS_REFILLS=default timeout -k 10 300 python tools/lane_census.py veach-mis:64 staircase:32 soup:16 blob:32 2>&1 | grep -v amdgpu.ids
