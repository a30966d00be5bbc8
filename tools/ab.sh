#!/bin/bash
# A/B on the GPU box: each remaining argument is "label|ENV=val ...|bench args"; prints one line per run.
for spec in "$@"; do
  IFS='|' read -r label envs args <<< "$spec"
  env $envs timeout -k 10 400 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
line=sys.stdin.readline()
if not line.strip(): print('%-34s FAILED' % '$label'); sys.exit(0)
d=json.loads(line); k=d['kernels_rank0']
g=lambda n: k.get(n,{}).get('ms_per_step',0.0)
print('%-34s %9.1f Mrays/s %9.2f ms/step | closest %8.2f shadow %8.2f shade %8.2f tail %6.2f gen %5.2f | %s %.0f GB/s | rays %.3fG | util %s' % ('$label', d['value'], d['ms_per_step'], g('trace_closest'), g('trace_shadow'), g('shade'), g('tail'), g('gen_primary'), d['roofline']['kernel'], d['roofline']['achieved'], d['rays_per_step']/1e9, d.get('simd_utilisation_traversal')))"
done
