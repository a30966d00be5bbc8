#!/bin/bash
# Round 4, GPU call 4: the whole -m gpu suite on the new code; every rank's stripes of an N-GPU render on this one GPU (tools/stripe_balance.py:
# VERDICT r03 task 3); node kind A/B on leaf-8 trees on one box; the --gpus 2 line with its per-rank fields (gloo, one GPU).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run4
mkdir -p $out
export PYTHONUNBUFFERED=1
echo "== pytest -m gpu"
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1 || { tail -30 $out/pytest_gpu.log; exit 1; }
tail -2 $out/pytest_gpu.log
echo "== stripes"
for sc in back veach-mis staircase; do
  timeout -k 10 300 python tools/stripe_balance.py $sc --blocks 4,8,16 --json $out/stripes_$sc.json 2>$out/stripes_$sc.err | tee $out/stripes_$sc.md
done
timeout -k 10 400 python tools/stripe_balance.py staircase --spp 1024 --blocks 8 --reps 1 --json $out/stripes_config4.json 2>$out/stripes_config4.err | tee $out/stripes_config4.md
timeout -k 10 400 python tools/stripe_balance.py blob --tris 10000000 --width 3840 --height 2160 --spp 64 --blocks 8 --json $out/stripes_config5_64spp.json 2>$out/stripes_config5.err | tee $out/stripes_config5_64spp.md
timeout -k 10 300 python tools/stripe_balance.py soup --spp 64 --blocks 8 --json $out/stripes_config3.json 2>$out/stripes_config3.err | tee $out/stripes_config3.md
run() { # tag envs args...
  tag=$1; envs=$2; shift 2
  env $envs timeout -k 10 500 python bench.py "$@" --no-cpu-baseline --no-extra --no-overlap-extra > $out/$tag.json 2> $out/$tag.err || echo "$tag failed"
  python - $out/$tag.json "$tag" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = {a: b["ms_per_step"] for a, b in d["kernels_rank0"].items() if b["ms_per_step"]}
    u = d["simd_utilisation_traversal"]
    print(f'{sys.argv[2]:24s} {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:9.2f} ms  closest {k.get("trace_closest", 0):8.2f} shade {k.get("shade", 0):7.2f} shadow {k.get("trace_shadow", 0):8.2f} tail {k.get("tail", 0):6.2f} | node bytes {d["config"].get("inner_node_bytes")} lanes {u["inner_steps"]}/{u["leaf_steps"]} visits {u["visits_per_ray"]} tests {u["tri_tests_per_ray"]}', flush=True)
except Exception as e:
    print(sys.argv[2], "no result", e, flush=True)
PY
}
echo "== node kind on leaf-8 trees, one box"
for nk in 0 1; do
  run veach_leaf8_nk$nk "TRT_NODE_KIND=$nk" --scene veach-mis --steps 3 --leaf 8
  run stair_leaf8_nk$nk "TRT_NODE_KIND=$nk" --scene staircase --steps 2 --leaf 8
  run blob2m_leaf8_nk$nk "TRT_NODE_KIND=$nk" --scene blob --tris 2000000 --spp 64 --steps 3 --leaf 8
done
run veach_leaf2 "" --scene veach-mis --steps 3
run stair_leaf2 "" --scene staircase --steps 2
echo "== bench --gpus 2 (gloo, two ranks on the one GPU)"
TRT_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 > $out/bench_gpus2_gloo.json 2> $out/bench_gpus2_gloo.err; echo "rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/run4/bench_gpus2_gloo.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["per_rank"], d["timing"])
PY
