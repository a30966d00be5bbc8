#!/bin/bash
# round 3, run 9: the early end of parity-mode shadow rays (LightBox, trt_oct.h) — GPU parity tests, then A/B against TRT_SHADOW_STOP=0
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu9.log 2>&1 || { tail -30 gpurun_out/r03/pytest_gpu9.log; exit 1; }
tail -2 gpurun_out/r03/pytest_gpu9.log
run() { # tag envs args...
  tag=$1; envs=$2; shift 2
  env $envs timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-extra --no-overlap-extra > gpurun_out/r03/$tag.json 2> gpurun_out/r03/$tag.err || echo "$tag failed"
  python - gpurun_out/r03/$tag.json "$tag" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2].ljust(28), d["value"], "Mrays/s", d["ms_per_step"], "ms", {k: v["ms_per_step"] for k, v in d["kernels_rank0"].items() if v["ms_per_step"]}, flush=True)
except Exception as e:
    print(sys.argv[2], "no result", e, flush=True)
PY
}
for ss in 0 1; do
  run s9_veach_stop$ss TRT_SHADOW_STOP=$ss --scene veach-mis --steps 3
  run s9_stair_stop$ss TRT_SHADOW_STOP=$ss --scene staircase --spp 64 --steps 2
  run s9_soup_stop$ss TRT_SHADOW_STOP=$ss --scene soup --spp 16 --steps 2
  run s9_blob2m_stop$ss TRT_SHADOW_STOP=$ss --scene blob --tris 2000000 --spp 64 --steps 2
done
