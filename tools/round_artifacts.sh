#!/bin/bash
# Regenerates the measured artifacts of a round on the GPU box (one gpurun call):
#   rocprofv3 kernel stats + FETCH/WRITE PMC passes of the headline command, the default bench line (headline +
#   extra_workloads + CPU baseline), and the bench lines of the two blob scenes.
# usage: tools/round_artifacts.sh r02     (outputs under gpurun_out/; copy the summaries to profiles/ with tools/collect_artifacts.sh r02)
r=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
tools/prof.sh ${r}_back --steps 3 --warmup 1 --no-extra > gpurun_out/prof_${r}_back.log 2>&1
echo "prof back done"
tools/prof.sh ${r}_soup --scene soup --spp 16 --steps 2 --warmup 1 > gpurun_out/prof_${r}_soup.log 2>&1
echo "prof soup done"
tools/prof.sh ${r}_blob10m --scene blob --tris 10000000 --width 3840 --height 2160 --spp 8 --steps 2 --warmup 1 > gpurun_out/prof_${r}_blob10m.log 2>&1
echo "prof blob10m done"
python bench.py --also-overlap > gpurun_out/${r}_bench_default.json 2> gpurun_out/${r}_bench_default.err
echo "default bench done"
python bench.py --scene blob --tris 2000000 --steps 2 --no-cpu-baseline > gpurun_out/${r}_bench_blob2m.json 2>/dev/null
python bench.py --scene blob --tris 10000000 --width 3840 --height 2160 --spp 64 --steps 1 --no-cpu-baseline > gpurun_out/${r}_bench_blob10m_4k.json 2>/dev/null
echo "blob done"
for f in gpurun_out/${r}_bench_*.json; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["value"], "Mrays/s", d["ms_per_step"], "ms", d["roofline"]["kernel"], d["roofline"]["frac"], {k: v["ms_per_step"] for k, v in d["kernels_rank0"].items() if v["ms_per_step"]})
for e in d.get("extra_workloads") or []:
    print("   extra", e["config"]["scene"], e["value"], "Mrays/s", e["ms_per_step"], "ms", e["roofline"]["kernel"], e["roofline"]["frac"])
PY
done
