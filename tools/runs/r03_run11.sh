#!/bin/bash
# round 3, run 11: the host half of trt_create on several threads, built once per group: GPU tests, start-up cost of 10 M triangles
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu11.log 2>&1 || { tail -30 gpurun_out/r03/pytest_gpu11.log; exit 1; }
tail -2 gpurun_out/r03/pytest_gpu11.log
timeout -k 10 400 python tools/create_cost.py 10000000 2>&1 | grep -v amdgpu.ids > gpurun_out/r03/create_cost_threads.log
cat gpurun_out/r03/create_cost_threads.log
TRT_HOST_THREADS=1 timeout -k 10 400 python tools/create_cost.py 10000000 2>&1 | grep -v amdgpu.ids > gpurun_out/r03/create_cost_1thread.log
grep "trt_create \|collapse\|group" gpurun_out/r03/create_cost_1thread.log
