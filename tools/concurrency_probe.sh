#!/bin/bash
# Do two independent render pipelines on one GPU overlap (trace is VALU-bound, shade latency-bound)?
# Runs bench.py alone at 256 spp, then two processes at 128 spp each concurrently; compares total rays / wall.
run() { python bench.py --no-cpu-baseline --steps 80 "$@" 2>/dev/null; }
echo "single:"; run --spp 256 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
echo "two concurrent (128 spp each):"
run --spp 128 > /tmp/a.json & run --spp 128 > /tmp/b.json & wait
python - <<'PY'
import json
a=json.load(open('/tmp/a.json')); b=json.load(open('/tmp/b.json'))
print('each:', a['value'], a['ms_per_step'], '|', b['value'], b['ms_per_step'], ' sum Mrays/s (if fully concurrent):', a['value']+b['value'])
PY
