#!/bin/bash
# Round 4, GPU call 8: k_shade in its two flavours (one light: 512-thread blocks at 6 waves per SIMD; several lights: 256-thread blocks at 5) — the whole
# -m gpu suite, then every workload of the default command against round 3's configuration (variant shade_old: 512-thread blocks, 4 waves).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run8
mkdir -p $out
export PYTHONUNBUFFERED=1
V=$root/tinyraytracing_amd/lib/variants
echo "== pytest -m gpu"
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
    print(f'{sys.argv[2]:24s} {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:9.2f} ms  closest {k.get("trace_closest", 0):8.2f} shade {k.get("shade", 0):7.2f} shadow {k.get("trace_shadow", 0):8.2f} tail {k.get("tail", 0):6.2f}', flush=True)
except Exception as e:
    print(sys.argv[2], "no result", e, flush=True)
PY
}
args() {
  case $1 in
    back) echo "--scene back --steps 5";; veach) echo "--scene veach-mis --steps 3";; stair) echo "--scene staircase --steps 2";;
    soup) echo "--scene soup --spp 64 --steps 3";; blob10m) echo "--scene blob --tris 10000000 --width 3840 --height 2160 --spp 64 --steps 3";;
  esac
}
for sc in back veach stair soup blob10m; do
  a=$(args $sc)
  run ${sc}_old "TRT_HIP_LIB=$V/libtrt_hip_shade_old.so" $a
  run ${sc}_new "" $a
  run ${sc}_old2 "TRT_HIP_LIB=$V/libtrt_hip_shade_old.so" $a
  run ${sc}_new2 "" $a
done
echo "== counters of k_shade, new build (waves per SIMD, memory wait)"
tools/roofs.sh r04_back "--scene back" > $out/roofs_back.log 2>&1; tail -30 gpurun_out/roofs_r04_back/summary.txt
