#!/bin/bash
# round 3, run 14: LBVH with the SAH top over clusters — tests, then quality / cost against the host SAH tree and the plain radix tree
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_lbvh.py -m gpu -x -q -s > gpurun_out/r03/pytest_lbvh2.log 2>&1 || { grep -v amdgpu.ids gpurun_out/r03/pytest_lbvh2.log | tail -40; exit 1; }
grep -v amdgpu.ids gpurun_out/r03/pytest_lbvh2.log | tail -6
timeout -k 10 900 python tools/lbvh_cost.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03/lbvh_cost3.log; cat gpurun_out/r03/lbvh_cost3.log
