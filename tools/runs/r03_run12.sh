#!/bin/bash
# round 3, run 12: the GPU LBVH builder — its tests, then cost and quality against the host SAH builder
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_lbvh.py -m gpu -x -q -s > gpurun_out/r03/pytest_lbvh.log 2>&1 || { tail -40 gpurun_out/r03/pytest_lbvh.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r03/pytest_lbvh.log | tail -6
timeout -k 10 600 python tools/lbvh_cost.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03/lbvh_cost.log; cat gpurun_out/r03/lbvh_cost.log
