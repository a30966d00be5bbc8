#!/bin/bash
# Round 4, GPU call 27: the library before the special-ray routing (commit 2808982, built as variants/libtrt_hip_prev.so) against the current one, alternating on one box:
# does the check of a ray's direction at write-back (three compares per ray) or the vote per ray of the wave-uniform walk cost anything?
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run27
mkdir -p $out
export PYTHONUNBUFFERED=1
P=$root/tinyraytracing_amd/lib/variants/libtrt_hip_prev.so
X="--steps 3 --warmup 1 --no-extra --no-traffic --no-overlap-extra"
bash tools/ab.sh \
  "back_prev|TRT_HIP_LIB=$P|$X" "back_now|TRT_X=1|$X" "back_prev2|TRT_HIP_LIB=$P|$X" "back_now2|TRT_X=1|$X" "back_prev3|TRT_HIP_LIB=$P|$X" "back_now3|TRT_X=1|$X" \
  "veach_prev|TRT_HIP_LIB=$P|--scene veach-mis $X" "veach_now|TRT_X=1|--scene veach-mis $X" \
  "stair_prev|TRT_HIP_LIB=$P|--scene staircase --spp 64 $X" "stair_now|TRT_X=1|--scene staircase --spp 64 $X" \
  "soup_prev|TRT_HIP_LIB=$P|--scene soup --spp 64 $X" "soup_now|TRT_X=1|--scene soup --spp 64 $X" "soup_prev2|TRT_HIP_LIB=$P|--scene soup --spp 64 $X" "soup_now2|TRT_X=1|--scene soup --spp 64 $X" \
  "blob_prev|TRT_HIP_LIB=$P|--scene blob --tris 10000000 --width 3840 --height 2160 --spp 64 $X" "blob_now|TRT_X=1|--scene blob --tris 10000000 --width 3840 --height 2160 --spp 64 $X" \
  2>&1 | tee $out/ab.txt
