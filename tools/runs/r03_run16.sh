#!/bin/bash
# round 3, run 16: the whole GPU suite on the final libraries (packed-fma arm removed, tinyrt --gpu-bvh), then the fuzz soak with trees of the GPU builder in the mix
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r03/pytest_gpu16.log 2>&1 || { grep -v amdgpu.ids gpurun_out/r03/pytest_gpu16.log | tail -40; exit 1; }
tail -2 gpurun_out/r03/pytest_gpu16.log
timeout -k 10 700 python tools/fuzz_parity.py 420 5 2>&1 | grep -v amdgpu.ids > gpurun_out/r03/fuzz_soak2.log; tail -4 gpurun_out/r03/fuzz_soak2.log
