#!/bin/bash
# rocprofv3 kernel statistics of the GPU BVH builder on 10 M triangles (three builds in one process).
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_lbvh
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/tools/lbvh_only.py 10000000 > $out/run.log 2> $out/stats.err
grep build $out/run.log
st=$(find $out/stats -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 $st | head -30
