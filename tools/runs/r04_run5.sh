#!/bin/bash
# Round 4, GPU call 5: k_shade's occupancy (VERDICT r03 task 4).  (a) how its time hangs on the waves per SIMD: dynamic LDS padding takes blocks off the CU
# (TRT_SHADE_PAD_LDS); (b) the same kernel compiled for 5 / 6 waves per SIMD with a block size whose multiples fill that occupancy (make variants: s640, s768, s512w5).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run5
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
    print(f'{sys.argv[2]:24s} {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:9.2f} ms  closest {k.get("trace_closest", 0):8.2f} shade {k.get("shade", 0):7.2f} shadow {k.get("trace_shadow", 0):8.2f} tail {k.get("tail", 0):6.2f}', flush=True)
except Exception as e:
    print(sys.argv[2], "no result", e, flush=True)
PY
}
echo "== parity of the variants (image + golden subset)"
for v in s640 s768; do
  TRT_HIP_LIB=$V/libtrt_hip_$v.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "image_matches or golden or chunking or overlapped or glass or fixed_nee_image" 2>&1 | tail -2
  test ${PIPESTATUS[0]} -eq 0 || exit 1
done
for sc in back veach stair; do
  case $sc in back) a="--scene back --steps 5";; veach) a="--scene veach-mis --steps 3";; stair) a="--scene staircase --spp 64 --steps 3";; esac
  echo "== $sc"
  run ${sc}_default "" $a
  for pad in 16000 45000 90000; do run ${sc}_pad$pad "TRT_SHADE_PAD_LDS=$pad" $a; done
  for v in s640 s768 s512w5; do run ${sc}_$v "TRT_HIP_LIB=$V/libtrt_hip_$v.so" $a; done
  run ${sc}_default_again "" $a
done
