#!/bin/bash
# Round 4, GPU call 38: non-temporal accesses, second round: bit 16 = the shadow kernels' weight loads, 32 = their read-modify-write of the radiance sums,
# on top of the kept 3 (k_shade's loads and stores).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run38
mkdir -p $out
export PYTHONUNBUFFERED=1
V=$root/tinyraytracing_amd/lib/variants
X="--steps 3 --warmup 1 --no-extra --no-traffic --no-overlap-extra"
for rep in 1 2; do
for sc in "back|" "veach|--scene veach-mis --spp 128" "stair|--scene staircase --spp 64" "soup|--scene soup --spp 64"; do
  IFS='|' read -r name args <<< "$sc"
  bash tools/ab.sh "${name}_nt3_$rep|TRT_X=1|$args $X" "${name}_nt19_$rep|TRT_HIP_LIB=$V/libtrt_hip_nt19.so|$args $X" "${name}_nt35_$rep|TRT_HIP_LIB=$V/libtrt_hip_nt35.so|$args $X" \
     "${name}_nt51_$rep|TRT_HIP_LIB=$V/libtrt_hip_nt51.so|$args $X"
done
done 2>&1 | tee $out/ab.txt
