#!/bin/bash
# Round 4, GPU call 17: node kinds on SMALL trees with the reference's leaf size (veach-mis lost 7 % to the exact 4-wide nodes in run 4): where is the crossover?
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run17
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
    print(f'{sys.argv[2]:24s} {d["config"]["triangles"]:8d} tris {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:9.2f} ms  closest {k.get("trace_closest", 0):8.2f} shade {k.get("shade", 0):7.2f} shadow {k.get("trace_shadow", 0):8.2f} | visits {u["visits_per_ray"]} tests {u["tri_tests_per_ray"]}', flush=True)
except Exception as e:
    print(sys.argv[2], "no result", e, flush=True)
PY
}
for nk in 0 1; do
  run veach_l8_nk$nk "TRT_NODE_KIND=$nk" --scene veach-mis --spp 64 --steps 3 --leaf 8
  for n in 1000 4000 16000 64000; do
    run soup${n}_l8_nk$nk "TRT_NODE_KIND=$nk" --scene soup --tris $n --spp 16 --steps 3 --leaf 8
    run blob${n}_l8_nk$nk "TRT_NODE_KIND=$nk" --scene blob --tris $n --spp 32 --steps 3 --leaf 8
  done
  run stair_l8_nk$nk "TRT_NODE_KIND=$nk" --scene staircase --spp 32 --steps 3 --leaf 8
done
