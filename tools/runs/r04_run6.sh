#!/bin/bash
# Round 4, GPU call 6: (a) k_tail_group (a group of lanes per path: the traversals of a bounce side by side) — the whole -m gpu suite runs through it,
# then A/B against k_tail (TRT_TAIL_GROUP=0) and a sweep of the hand-over point TRT_TAIL_N; (b) k_shade in 256-thread blocks (4 / 5 blocks per CU).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run6
mkdir -p $out
export PYTHONUNBUFFERED=1
V=$root/tinyraytracing_amd/lib/variants
echo "== pytest -m gpu (grouped tail)"
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1 || { tail -30 $out/pytest_gpu.log; exit 1; }
tail -2 $out/pytest_gpu.log
run() { # tag envs args...
  tag=$1; envs=$2; shift 2
  env $envs timeout -k 10 500 python bench.py "$@" --no-cpu-baseline --no-extra --no-overlap-extra > $out/$tag.json 2> $out/$tag.err || echo "$tag failed"
  python - $out/$tag.json "$tag" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = {a: b["ms_per_step"] for a, b in d["kernels_rank0"].items() if b["ms_per_step"]}
    print(f'{sys.argv[2]:24s} {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:9.2f} ms  closest {k.get("trace_closest", 0):8.2f} shade {k.get("shade", 0):7.2f} shadow {k.get("trace_shadow", 0):8.2f} tail {k.get("tail", 0):6.2f}  launches {sum(b["launches_per_step"] for b in d["kernels_rank0"].values())}', flush=True)
except Exception as e:
    print(sys.argv[2], "no result", e, flush=True)
PY
}
args() {
  case $1 in
    back) echo "--scene back --steps 5";; veach) echo "--scene veach-mis --steps 3";; stair) echo "--scene staircase --spp 64 --steps 3";;
    soup) echo "--scene soup --spp 64 --steps 3";; blob10m) echo "--scene blob --tris 10000000 --width 3840 --height 2160 --spp 64 --steps 3";;
  esac
}
echo "== tail: one lane per path (TRT_TAIL_GROUP=0) against a group of lanes; hand-over point"
for sc in back veach stair soup blob10m; do
  a=$(args $sc)
  run ${sc}_tail_lane "TRT_TAIL_GROUP=0" $a
  run ${sc}_tail_group "" $a
  for n in 262144 524288 1048576; do run ${sc}_tail_group_n$n "TRT_TAIL_N=$n" $a; done
done
echo "== k_shade in 256-thread blocks"
for sc in back veach stair; do
  a=$(args $sc)
  for v in s512r s256 s256w5; do run ${sc}_$v "TRT_HIP_LIB=$V/libtrt_hip_$v.so" $a; done
done
