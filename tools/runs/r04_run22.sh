#!/bin/bash
# Round 4, GPU call 22: the kernels on poisoned geometry (NaN / inf / 1e38 in the caller's arrays) — every launch must end, image = oracle's; then the whole -m gpu suite.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run22
mkdir -p $out
export PYTHONUNBUFFERED=1
echo "== poisoned geometry"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k non_finite 2>&1 | tee $out/poison.log | tail -5 || exit 1
echo "== pytest -m gpu"
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tee $out/pytest_gpu.log | tail -3
