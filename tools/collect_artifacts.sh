#!/bin/bash
# Copies the judged summaries of tools/round_artifacts.sh from gpurun_out/ (scratch) to profiles/ (tracked).
# usage: tools/collect_artifacts.sh r02
r=$1
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root
for tag in back soup blob10m; do
  d=gpurun_out/prof_${r}_$tag
  [ -d $d ] || continue
  st=$(find $d/stats -name "*kernel_stats.csv" | head -1)
  [ -n "$st" ] && cp $st profiles/${r}_${tag}_kernel_stats.csv
  [ -f $d/bench_under_stats.json ] && cp $d/bench_under_stats.json profiles/${r}_${tag}_bench_under_rocprof_stats.json
done
python3 tools/pmc_summary.py gpurun_out/prof_${r}_back ${r}_back_1080p_256spp back 1080 256
python3 tools/pmc_summary.py gpurun_out/prof_${r}_soup ${r}_soup_1080p_16spp soup 1080 16
python3 tools/pmc_summary.py gpurun_out/prof_${r}_blob10m ${r}_blob10m_2160p_8spp blob 2160 8
for f in gpurun_out/${r}_bench_*.json; do cp $f profiles/; done
ls -la profiles | tail -20
