#!/bin/bash
# Round 4, GPU call 36: non-temporal accesses to the ray queues (written once by one kernel, read once by the next, far more than the caches hold):
# TRT_NT bit 1 = k_shade's loads, 2 = k_shade's stores, 4 = the traversal kernels' ray loads, 8 = their hit stores.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run36
mkdir -p $out
export PYTHONUNBUFFERED=1
V=$root/tinyraytracing_amd/lib/variants
X="--steps 3 --warmup 1 --no-extra --no-traffic --no-overlap-extra"
for rep in 1 2; do
for sc in "back|" "veach|--scene veach-mis --spp 128" "soup|--scene soup --spp 64" "blob|--scene blob --tris 10000000 --width 3840 --height 2160 --spp 32"; do
  IFS='|' read -r name args <<< "$sc"
  bash tools/ab.sh "${name}_nt0_$rep|TRT_X=1|$args $X" "${name}_nt1_$rep|TRT_HIP_LIB=$V/libtrt_hip_nt1.so|$args $X" "${name}_nt2_$rep|TRT_HIP_LIB=$V/libtrt_hip_nt2.so|$args $X" \
     "${name}_nt3_$rep|TRT_HIP_LIB=$V/libtrt_hip_nt3.so|$args $X" "${name}_nt12_$rep|TRT_HIP_LIB=$V/libtrt_hip_nt12.so|$args $X" "${name}_nt15_$rep|TRT_HIP_LIB=$V/libtrt_hip_nt15.so|$args $X"
done
done 2>&1 | tee $out/ab.txt
