#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/ab
for f in 0 1; do
  TRT_FUSED=$f python bench.py --no-cpu-baseline --no-extra --steps 3 > gpurun_out/ab/back_fused$f.json 2> gpurun_out/ab/back_fused$f.err || echo "fused$f failed"
  python - gpurun_out/ab/back_fused$f.json fused$f <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("back", sys.argv[2], d["value"], "Mrays/s", d["ms_per_step"], "ms", {k: v["ms_per_step"] for k, v in d["kernels_rank0"].items() if v["ms_per_step"]}, flush=True)
PY
done
for b in 1024 2048 8192; do
  TRT_BOUNCE_BLOCKS=$b python bench.py --no-cpu-baseline --no-extra --steps 3 > gpurun_out/ab/back_bb$b.json 2> gpurun_out/ab/back_bb$b.err
  python - gpurun_out/ab/back_bb$b.json bb$b <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("back", sys.argv[2], d["value"], "Mrays/s", d["ms_per_step"], "ms", {k: v["ms_per_step"] for k, v in d["kernels_rank0"].items() if v["ms_per_step"]}, flush=True)
PY
done
