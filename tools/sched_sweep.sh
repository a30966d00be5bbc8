#!/bin/bash
# Sweep of the scheduler driver's step-choice weights (TRT_SCHED_W=in:leaf: node step iff in * lanes-at-nodes >= leaf * lanes-at-leaves).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for w in "${@:-1:1 2:3 1:2 1:3}"; do
  for sc in "veach-mis --spp 256" "staircase --spp 64" "soup --spp 16" "blob --tris 2000000 --spp 64"; do
    TRT_SCHED_W=$w python bench.py --scene $sc --steps 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('w=$w', d['config']['scene'], d['value'], d['ms_per_step'], d['simd_utilisation_traversal'], {k:v['ms_per_step'] for k,v in d['kernels_rank0'].items() if v['ms_per_step']}, flush=True)"
  done
done
