#!/bin/bash
# Round 4, GPU call 23: non-finite rays and poisoned boxes through the kernels (each under its own timeout), then the whole -m gpu suite on the rebuilt library.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run23
mkdir -p $out
export PYTHONUNBUFFERED=1
echo "== non-finite rays and geometry"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k non_finite 2>&1 | tee $out/poison.log | tail -5 || exit 1
echo "== pytest -m gpu"
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tee $out/pytest_gpu.log | tail -3 || exit 1
echo "== abi tests on a box with a device"
timeout -k 10 300 python -m pytest tests/test_abi.py -x -q 2>&1 | tee $out/abi.log | tail -3
