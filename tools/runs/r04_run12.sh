#!/bin/bash
# Round 4, GPU call 12: parity soak on the final libraries; the GPU builder's trees of staircase / veach-mis dumped for the CPU study; BASELINE.md §4's table.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run12
mkdir -p $out
export PYTHONUNBUFFERED=1
echo "== fuzz soak"
timeout -k 10 700 python tools/fuzz_parity.py 480 41 > $out/fuzz_soak.txt 2> $out/fuzz_soak.err; echo "rc $?"; tail -3 $out/fuzz_soak.txt
for sc in staircase veach-mis; do python tools/lbvh_dump.py $out/lbvh_$sc.npz $sc 2 2>&1 | grep --line-buffered -v amdgpu.ids; python tools/lbvh_dump.py $out/lbvh_${sc}_c16.npz $sc 2 16 2>&1 | grep --line-buffered -v amdgpu.ids; done
echo "== baseline table"
bash tools/baseline_table.sh > $out/baseline_table.md 2> $out/baseline_table.err; echo "rc $?"
cat $out/baseline_table.md | cut -c1-260
mkdir -p $out/baseline && cp gpurun_out/baseline/*.json $out/baseline/
