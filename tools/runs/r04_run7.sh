#!/bin/bash
# Round 4, GPU call 7: (a) k_shade compiled for ONE light (no per-light stage pipeline: 92 VGPRs; with 6 waves per SIMD asked for: 80, no scratch) on the
# single-light scenes; (b) the hand-over point to k_tail BELOW the default 131072 on the scenes whose tail is long (round 1 swept it on `back` only).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run7
mkdir -p $out
export PYTHONUNBUFFERED=1
V=$root/tinyraytracing_amd/lib/variants
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
echo "== parity of the one-light build on a one-light scene"
TRT_HIP_LIB=$V/libtrt_hip_nl1w6.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "(image_matches or golden or chunking or overlapped or config2 or soup) and not veach and not staircase" 2>&1 | tail -2
test ${PIPESTATUS[0]} -eq 0 || exit 1
echo "== k_shade for one light"
for sc in back soup blob10m; do
  a=$(args $sc)
  run ${sc}_default "" $a
  run ${sc}_nl1 "TRT_HIP_LIB=$V/libtrt_hip_nl1.so" $a
  run ${sc}_nl1w6 "TRT_HIP_LIB=$V/libtrt_hip_nl1w6.so" $a
done
echo "== hand-over point to k_tail"
for sc in soup blob10m stair veach back; do
  a=$(args $sc)
  for n in 4096 16384 32768 65536 131072; do run ${sc}_tail_n$n "TRT_TAIL_N=$n TRT_TAIL_GROUP=0" $a; done
done
