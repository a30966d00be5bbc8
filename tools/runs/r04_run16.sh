#!/bin/bash
# Round 4, GPU call 16: a second, longer parity soak on the final libraries (the leaf-8 handles now on the reference builder's own trees).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run16
mkdir -p $out
export PYTHONUNBUFFERED=1
timeout -k 10 1000 python tools/fuzz_parity.py 900 53 > $out/fuzz_soak2.txt 2> $out/fuzz_soak2.err; echo "rc $?"; tail -3 $out/fuzz_soak2.txt; tail -3 $out/fuzz_soak2.err
