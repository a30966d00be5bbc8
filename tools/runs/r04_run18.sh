#!/bin/bash
# Round 4, GPU call 18: every rank's stripes on one GPU once more, on the FINAL kernels (run 4 predates the two k_shade flavours).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run18
mkdir -p $out
export PYTHONUNBUFFERED=1
for sc in back veach-mis staircase; do
  timeout -k 10 300 python tools/stripe_balance.py $sc --blocks 4,8,16 --json $out/stripes_$sc.json 2>$out/stripes_$sc.err | tee $out/stripes_$sc.md
done
timeout -k 10 400 python tools/stripe_balance.py staircase --spp 1024 --blocks 8 --reps 1 --json $out/stripes_config4.json 2>$out/stripes_config4.err | tee $out/stripes_config4.md
timeout -k 10 400 python tools/stripe_balance.py blob --tris 10000000 --width 3840 --height 2160 --spp 64 --blocks 8 --json $out/stripes_config5_64spp.json 2>$out/stripes_config5.err | tee $out/stripes_config5_64spp.md
timeout -k 10 300 python tools/stripe_balance.py soup --spp 64 --blocks 8 --json $out/stripes_config3.json 2>$out/stripes_config3.err | tee $out/stripes_config3.md
