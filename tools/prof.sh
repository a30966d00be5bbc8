#!/bin/bash
# Collects the rocprofv3 evidence for one bench.py workload on the GPU box:
#   <tag>_stats   : --kernel-trace --stats (per-kernel time; must agree with bench.py's hipEvent numbers)
#   <tag>_fetch / <tag>_write : separate --pmc passes (FETCH_SIZE / WRITE_SIZE need all TCC slots each)
# usage: tools/prof.sh <tag> <bench.py args...>
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py "$@" --no-cpu-baseline --no-overlap-extra --no-traffic > $out/bench_under_stats.json 2> $out/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 $root/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-overlap-extra --no-traffic > $out/bench_under_fetch.json 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 $root/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-overlap-extra --no-traffic > $out/bench_under_write.json 2> $out/write.err
find $out -name "*.csv" | head -50
