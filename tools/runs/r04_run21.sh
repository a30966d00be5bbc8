#!/bin/bash
# Round 4, GPU call 21: two passes in flight — do the traversal kernels and k_shade of DIFFERENT passes share a CU when the traversal grid is capped below the
# chip's wave slots (TRT_TRACE_MAXB; 2048 blocks of 256 fill 8 waves per SIMD)?  Traversal is issue-bound, k_shade waits on memory: side by side they should overlap.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run21
mkdir -p $out
export PYTHONUNBUFFERED=1
one() {  # tag scene env...
  tag=$1; sc=$2; shift 2
  env "$@" timeout -k 10 200 python bench.py --scene $sc --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-traffic --no-overlap-extra $OV >$out/$tag.json 2>$out/$tag.err
  python - $out/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:28s} {d['value']:9.2f} Mrays/s {d['ms_per_step']:9.3f} ms")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for sc in back veach-mis staircase; do
  OV=""; one ${sc}_plain $sc TRT_DUMMY=1
  OV="--overlap"
  one ${sc}_ov $sc TRT_DUMMY=1
  for mb in 4096 2048 1536 1024 768; do one ${sc}_ov_maxb$mb $sc TRT_TRACE_MAXB=$mb; done
  OV=""; one ${sc}_plain_maxb2048 $sc TRT_TRACE_MAXB=2048
done 2>&1 | tee $out/summary.txt
