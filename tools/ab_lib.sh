#!/bin/bash
# A/B of builds of the HIP library (tinyraytracing_amd/lib/variants/libtrt_hip_<name>.so, selected with TRT_HIP_LIB) on the
# three cg22 scenes and the soup.  usage: tools/ab_lib.sh name1 name2 ...
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/ab
run() { # tag envs args...
  tag=$1; envs=$2; shift 2
  env $envs python bench.py "$@" --no-cpu-baseline --no-extra > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err || echo "$tag failed"
  python - gpurun_out/ab/$tag.json "$tag" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2].ljust(28), d["value"], "Mrays/s", d["ms_per_step"], "ms", {k: v["ms_per_step"] for k, v in d["kernels_rank0"].items() if v["ms_per_step"]}, flush=True)
except Exception as e:
    print(sys.argv[2], "no result", e, flush=True)
PY
}
for v in "$@"; do
  lib=TRT_HIP_LIB=$root/tinyraytracing_amd/lib/variants/libtrt_hip_$v.so
  run back_$v $lib --steps 10 --warmup 2
  run veach_$v $lib --scene veach-mis --steps 2
  run stair_$v $lib --scene staircase --spp 64 --steps 2
  run soup_$v $lib --scene soup --spp 16 --steps 2
done
