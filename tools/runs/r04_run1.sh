#!/bin/bash
# Round 4, GPU call 1: where the round starts — the default bench line, and the reference's own leaf size (main.cpp:76: buildBVH(..., 8))
# on the shipped code (leaves of > 3 triangles walk the exact 4-wide nodes) against this repo's leaf-2 default.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run1
mkdir -p $out
export PYTHONUNBUFFERED=1
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
echo "== default bench"
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err || { echo "default bench failed"; tail -5 $out/bench_default.err; exit 1; }
python - $out/bench_default.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("headline", d["value"], d["ms_per_step"], d["roofline"])
for w in d.get("extra_workloads", []):
    print(w["config"]["workload"], w["value"], w["ms_per_step"])
PY
echo "== leaf 2 (default) against leaf 8 (the reference's), shipped code"
for leaf in 2 8; do
  run veach_leaf$leaf "" --scene veach-mis --steps 2 --leaf $leaf
  run stair_leaf$leaf "" --scene staircase --steps 2 --leaf $leaf
done
run soup_leaf8 "" --scene soup --spp 64 --steps 2 --leaf 8
run soup_leaf1 "" --scene soup --spp 64 --steps 2
run blob2m_leaf8 "" --scene blob --tris 2000000 --spp 64 --steps 2 --leaf 8
run blob2m_leaf2 "" --scene blob --tris 2000000 --spp 64 --steps 2
