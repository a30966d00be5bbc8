#!/bin/bash
# round 3, run 17: where the tail kernel takes over (TRT_TAIL_N paths left in a pass), big scenes
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/r03
run() { # label envs args...
  label=$1; envs=$2; shift 2
  env $envs timeout -k 10 400 python bench.py "$@" --no-cpu-baseline --no-extra --no-overlap-extra > gpurun_out/r03/$label.json 2> gpurun_out/r03/$label.err || echo "$label failed"
  python - gpurun_out/r03/$label.json "$label" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = {a: b["ms_per_step"] for a, b in d["kernels_rank0"].items() if b["ms_per_step"]}
    print(f'{sys.argv[2]:26s} {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:9.2f} ms  closest {k.get("trace_closest", 0):8.2f} shade {k.get("shade", 0):7.2f} shadow {k.get("trace_shadow", 0):8.2f} tail {k.get("tail", 0):6.2f} launches {d["kernels_rank0"]["trace_closest"]["launches_per_step"]}', flush=True)
except Exception as e:
    print(sys.argv[2], "no result", e, flush=True)
PY
}
for tn in 131072 32768 8192 524288; do
  run s17_soup_tail$tn TRT_TAIL_N=$tn --scene soup --spp 64 --steps 2
  run s17_blob10m_tail$tn TRT_TAIL_N=$tn --scene blob --tris 10000000 --width 3840 --height 2160 --spp 64 --steps 2
  run s17_stair_tail$tn TRT_TAIL_N=$tn --scene staircase --spp 64 --steps 2
done
