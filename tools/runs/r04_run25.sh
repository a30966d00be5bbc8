#!/bin/bash
# Round 4, GPU call 25: hostile values in the scene tables through the kernels (own timeout), then the whole -m gpu suite on the rebuilt library.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run25
mkdir -p $out
export PYTHONUNBUFFERED=1
echo "== hostile tables"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "hostile" 2>&1 | tee $out/hostile.log | tail -5 || exit 1
echo "== pytest -m gpu"
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tee $out/pytest_gpu.log | tail -3 || exit 1
