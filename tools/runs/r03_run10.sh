#!/bin/bash
# round 3, run 10: what could a record diet of k_shade buy at most?  `short` = weight and throughput records stored as 8 B instead of 16 (wrong images, perf probe)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/r03
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
V=$root/tinyraytracing_amd/lib/variants
for rep in 1 2; do
  run s10_back_default_$rep X=1 --steps 5
  run s10_back_short_$rep TRT_HIP_LIB=$V/libtrt_hip_short.so --steps 5
  run s10_back_w8_$rep TRT_HIP_LIB=$V/libtrt_hip_w8.so --steps 5
done
run s10_veach_default X=1 --scene veach-mis --steps 3
run s10_veach_short TRT_HIP_LIB=$V/libtrt_hip_short.so --scene veach-mis --steps 3
run s10_veach_w8 TRT_HIP_LIB=$V/libtrt_hip_w8.so --scene veach-mis --steps 3
run s10_veach_refill40 TRT_REFILL_MIN=40 --scene veach-mis --steps 3
run s10_veach_refill56 TRT_REFILL_MIN=56 --scene veach-mis --steps 3
