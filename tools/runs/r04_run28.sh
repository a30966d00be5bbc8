#!/bin/bash
# Round 4, GPU call 28: where do the 2-3 % of run 27 come from?  prev = before the special-ray routing; nostore = without the direction check at write-back;
# nowalk = without the vote in the wave-uniform walk.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run28
mkdir -p $out
export PYTHONUNBUFFERED=1
V=$root/tinyraytracing_amd/lib/variants
X="--steps 3 --warmup 1 --no-extra --no-traffic --no-overlap-extra"
S="--scene soup --spp 64 $X"
bash tools/ab.sh \
  "back_prev|TRT_HIP_LIB=$V/libtrt_hip_prev.so|$X" "back_now|TRT_X=1|$X" "back_nowalk|TRT_HIP_LIB=$V/libtrt_hip_nowalk.so|$X" "back_nostore|TRT_HIP_LIB=$V/libtrt_hip_nostore.so|$X" \
  "back_prev2|TRT_HIP_LIB=$V/libtrt_hip_prev.so|$X" "back_now2|TRT_X=1|$X" "back_nowalk2|TRT_HIP_LIB=$V/libtrt_hip_nowalk.so|$X" "back_nostore2|TRT_HIP_LIB=$V/libtrt_hip_nostore.so|$X" \
  "soup_prev|TRT_HIP_LIB=$V/libtrt_hip_prev.so|$S" "soup_now|TRT_X=1|$S" "soup_nostore|TRT_HIP_LIB=$V/libtrt_hip_nostore.so|$S" \
  "soup_prev2|TRT_HIP_LIB=$V/libtrt_hip_prev.so|$S" "soup_now2|TRT_X=1|$S" "soup_nostore2|TRT_HIP_LIB=$V/libtrt_hip_nostore.so|$S" \
  2>&1 | tee $out/ab.txt
