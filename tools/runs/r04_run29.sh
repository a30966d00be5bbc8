#!/bin/bash
# Round 4, GPU call 29: the three instructions of the direction check move the traversal kernels by 2-3 % — code placement?  The same sources with every block that has
# no fall-through predecessor (loop headers entered by a jump, no executed padding) aligned to 64 B / 32 B, against the unaligned builds.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run29
mkdir -p $out
export PYTHONUNBUFFERED=1
V=$root/tinyraytracing_amd/lib/variants
X="--steps 3 --warmup 1 --no-extra --no-traffic --no-overlap-extra"
for rep in 1 2; do
for sc in "back|" "soup|--scene soup --spp 64" "stair|--scene staircase --spp 64" "veach|--scene veach-mis --spp 128" "blob|--scene blob --tris 10000000 --width 3840 --height 2160 --spp 32"; do
  IFS='|' read -r name args <<< "$sc"
  bash tools/ab.sh "${name}_prev$rep|TRT_HIP_LIB=$V/libtrt_hip_prev.so|$args $X" "${name}_now$rep|TRT_X=1|$args $X" "${name}_al6_$rep|TRT_HIP_LIB=$V/libtrt_hip_al6.so|$args $X" \
     "${name}_al5_$rep|TRT_HIP_LIB=$V/libtrt_hip_al5.so|$args $X" "${name}_prev_al6_$rep|TRT_HIP_LIB=$V/libtrt_hip_prev_al6.so|$args $X"
done
done 2>&1 | tee $out/ab.txt
