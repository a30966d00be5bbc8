#!/bin/bash
# Round 4, GPU call 42: rocprofv3 evidence of the headline command on the FINAL library (kernel stats + FETCH / WRITE passes) — run 15's predates the special-ray
# routing and the non-temporal queue records.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run42
mkdir -p $out
export PYTHONUNBUFFERED=1
tools/prof.sh r04_back_final --steps 3 --warmup 1 --no-extra > $out/prof_back.log 2>&1; echo "prof rc $?"
python tools/pmc_summary.py gpurun_out/prof_r04_back_final r04_back_1080p_256spp back 1080 256 && cp profiles/r04_back_1080p_256spp_pmc_hbm_bytes.csv profiles/hbm_traffic_back_1080p_256spp.json $out/
f=$(find gpurun_out/prof_r04_back_final/stats -name "*kernel_stats.csv" | head -1); test -n "$f" && cp $f $out/r04_back_kernel_stats.csv; cp gpurun_out/prof_r04_back_final/bench_under_stats.json $out/r04_back_bench_under_rocprof_stats.json
head -9 $out/r04_back_kernel_stats.csv | cut -c1-150
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/run42/r04_back_bench_under_rocprof_stats.json").read().strip().splitlines()[-1])
print("bench under rocprof:", d["value"], d["ms_per_step"], {k: (v["avg_launch_ms"]) for k, v in d["roofline"]["by_kernel"].items()})
PY
