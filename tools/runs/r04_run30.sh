#!/bin/bash
# Round 4, GPU call 30: special rays, third form — the wave-uniform walk lists them and its last block walks them again (nothing inside the walk), the per-lane
# kernels' redo pass clears them with a Bloom filter over the box planes before paying for the unculled walk.  New tests first, then prev against now.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run30
mkdir -p $out
export PYTHONUNBUFFERED=1
echo "== new tests"
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "zero_direction or hostile or non_finite or redo or foreign or golden" 2>&1 | tee $out/new.log | tail -4 || exit 1
V=$root/tinyraytracing_amd/lib/variants
X="--steps 3 --warmup 1 --no-extra --no-traffic --no-overlap-extra"
for rep in 1 2; do
for sc in "back|" "soup|--scene soup --spp 64" "stair|--scene staircase --spp 64" "blob|--scene blob --tris 10000000 --width 3840 --height 2160 --spp 32" "veach|--scene veach-mis --spp 128"; do
  IFS='|' read -r name args <<< "$sc"
  bash tools/ab.sh "${name}_prev$rep|TRT_HIP_LIB=$V/libtrt_hip_prev.so|$args $X" "${name}_now$rep|TRT_X=1|$args $X"
done
done 2>&1 | tee $out/ab.txt
