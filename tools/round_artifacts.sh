#!/bin/bash
# Regenerates every measured artifact of a round on the GPU box (one gpurun call):
#   rocprofv3 kernel stats + FETCH/WRITE PMC of the headline workload, then the bench lines of all workloads.
# usage: tools/round_artifacts.sh r01     (outputs under gpurun_out/; copy to profiles/ with tools/collect_artifacts.sh)
set -e
r=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
tools/prof.sh ${r}_back --steps 3 --warmup 1 > gpurun_out/prof_${r}_back.log 2>&1
echo "prof done"
python bench.py --also-overlap > gpurun_out/${r}_bench_back.json 2> gpurun_out/${r}_bench_back.err
echo "back done"
python bench.py --overlap --no-cpu-baseline > gpurun_out/${r}_bench_back_overlap.json 2>/dev/null
python bench.py --scene veach-mis --no-cpu-baseline > gpurun_out/${r}_bench_veach.json 2>/dev/null
echo "veach done"
python bench.py --scene staircase --steps 2 --no-cpu-baseline > gpurun_out/${r}_bench_staircase.json 2>/dev/null
echo "staircase done"
python bench.py --scene soup --spp 64 --steps 2 --no-cpu-baseline > gpurun_out/${r}_bench_soup.json 2>/dev/null
python bench.py --scene blob --tris 2000000 --steps 2 --no-cpu-baseline > gpurun_out/${r}_bench_blob2m.json 2>/dev/null
echo "blob done"
python bench.py --scene blob --tris 10000000 --width 3840 --height 2160 --spp 16 --steps 1 --no-cpu-baseline > gpurun_out/${r}_bench_blob10m_4k.json 2>/dev/null
for f in gpurun_out/${r}_bench_*.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], d["value"], "Mrays/s", d["ms_per_step"], "ms", d["roofline"]["kernel"], d["roofline"]["frac"], {k: v["ms_per_step"] for k, v in d["kernels_rank0"].items() if v["ms_per_step"]})
PY
done
