#!/bin/bash
# Round 4, GPU call 2: leaves of 4..15 triangles on the 8-wide nodes (several slots with the leaf's own box): parity, then the leaf-8 rows
# again (run 1 has them on the exact 4-wide nodes); the exact decompositions of the staircase render for the fit against the
# reference's converged snapshot (tools/staircase_decomp.py).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run2
mkdir -p $out
export PYTHONUNBUFFERED=1
echo "== parity"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lbvh.py -m gpu -q -x -k "own_leaf_size or node_kind or compressed or tiny_scene or grazing or golden or incoherent or degenerate or (lbvh_tree and staircase)" 2>&1 | tail -5
test ${PIPESTATUS[0]} -eq 0 || exit 1
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
echo "== leaf 8 on the 8-wide nodes"
run veach_leaf8 "" --scene veach-mis --steps 2 --leaf 8
run stair_leaf8 "" --scene staircase --steps 2 --leaf 8
run stair_leaf4 "" --scene staircase --steps 2 --leaf 4
run stair_leaf3 "" --scene staircase --steps 2 --leaf 3
run blob2m_leaf8 "" --scene blob --tris 2000000 --spp 64 --steps 2 --leaf 8
run soup_leaf8 "" --scene soup --spp 64 --steps 2 --leaf 8
run back_leaf8_lane "TRT_TRACE_IMPL=3" --scene back --steps 2
echo "== staircase decompositions"
timeout -k 10 600 python tools/staircase_decomp.py $out/staircase_decomp.npz 256 8 2>&1 | tail -12
