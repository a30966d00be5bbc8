#!/bin/bash
# Round 4, GPU call 9: tree quality of the GPU builder against the size of its clusters (VERDICT r03 task 8).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run9
mkdir -p $out
export PYTHONUNBUFFERED=1
timeout -k 10 1000 python tools/lbvh_cluster_sweep.py staircase veach-mis blob:150000 blob:2000000 soup:1000000 blob:10000000 2>&1 | grep -v amdgpu.ids | tee $out/cluster_sweep.txt
