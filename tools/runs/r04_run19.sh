#!/bin/bash
# Round 4, GPU call 19: the stripes of an 8-GPU render with two passes in flight (TRT_FLAG_OVERLAP): does the overlap hide the end of a pass on a rank's short steps?
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run19
mkdir -p $out
export PYTHONUNBUFFERED=1
for sc in back veach-mis staircase; do
  timeout -k 10 300 python tools/stripe_balance.py $sc --blocks 8 --ranks 8 --overlap 2>$out/ov_$sc.err | tee $out/ov_$sc.md
  timeout -k 10 300 python tools/stripe_balance.py $sc --blocks 8 --ranks 8 2>$out/no_$sc.err | tee $out/no_$sc.md
done
timeout -k 10 300 python tools/stripe_balance.py soup --spp 64 --blocks 8 --ranks 8 --overlap 2>$out/ov_soup.err | tee $out/ov_soup.md
timeout -k 10 400 python tools/stripe_balance.py blob --tris 10000000 --width 3840 --height 2160 --spp 64 --blocks 8 --ranks 8 --overlap 2>$out/ov_blob.err | tee $out/ov_blob.md
