#!/bin/bash
# Round 4, GPU call 15: rocprofv3 evidence of the headline command on the final kernels (kernel stats + FETCH / WRITE passes), counters of staircase
# (k_shade in 256-thread blocks), and the -m gpu tests added last (leaf-loop values, device unchanged by the GPU builder).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run15
mkdir -p $out
export PYTHONUNBUFFERED=1
echo "== new tests"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lbvh.py -m gpu -q -x -k "own_leaf_size or config5_ten_million_triangles_built" 2>&1 | tail -3
test ${PIPESTATUS[0]} -eq 0 || exit 1
echo "== rocprof of the headline"
tools/prof.sh r04_back --steps 3 --warmup 1 --no-extra > $out/prof_back.log 2>&1; echo "prof rc $?"
python tools/pmc_summary.py gpurun_out/prof_r04_back r04_back_1080p_256spp back 1080 256 && cp profiles/r04_back_1080p_256spp_pmc_hbm_bytes.csv profiles/hbm_traffic_back_1080p_256spp.json $out/
f=$(find gpurun_out/prof_r04_back/stats -name "*kernel_stats.csv" | head -1); test -n "$f" && cp $f $out/r04_back_kernel_stats.csv; cp gpurun_out/prof_r04_back/bench_under_stats.json $out/r04_back_bench_under_rocprof_stats.json
head -8 $out/r04_back_kernel_stats.csv | cut -c1-160
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/run15/r04_back_bench_under_rocprof_stats.json").read().strip().splitlines()[-1])
print("bench under rocprof:", d["value"], d["ms_per_step"], {k: (v["avg_launch_ms"]) for k, v in d["roofline"]["by_kernel"].items()})
PY
echo "== counters: staircase 64 spp"
tools/roofs.sh r04_stair "--scene staircase --spp 64" > $out/roofs_stair.log 2>&1; head -16 gpurun_out/roofs_r04_stair/summary.txt
cp gpurun_out/roofs_r04_stair/summary.txt $out/r04_roofs_stair.txt
