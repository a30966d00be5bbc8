#!/bin/bash
# Round 4, GPU call 31: the final form of the special-ray handling: tests, then prev against now (three times on the headline).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run31
mkdir -p $out
export PYTHONUNBUFFERED=1
echo "== new tests"
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "zero_direction or hostile or non_finite or redo or foreign or golden or axis_aligned or degenerate" 2>&1 | tee $out/new.log | tail -4 || exit 1
V=$root/tinyraytracing_amd/lib/variants
X="--steps 3 --warmup 1 --no-extra --no-traffic --no-overlap-extra"
for rep in 1 2 3; do
  bash tools/ab.sh "back_prev$rep|TRT_HIP_LIB=$V/libtrt_hip_prev.so|$X" "back_now$rep|TRT_X=1|$X"
done 2>&1 | tee $out/ab.txt
for sc in "soup|--scene soup --spp 64" "stair|--scene staircase --spp 64" "blob|--scene blob --tris 10000000 --width 3840 --height 2160 --spp 32" "veach|--scene veach-mis --spp 128"; do
  IFS='|' read -r name args <<< "$sc"
  bash tools/ab.sh "${name}_prev|TRT_HIP_LIB=$V/libtrt_hip_prev.so|$args $X" "${name}_now|TRT_X=1|$args $X"
done 2>&1 | tee -a $out/ab.txt
