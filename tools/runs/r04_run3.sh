#!/bin/bash
# Round 4, GPU call 3: triangles per leaf step (sc.leaf_loop) on trees with the reference's leaf size; rocprof kernel stats and the
# FETCH_SIZE / WRITE_SIZE passes of veach-mis and staircase at 1080p / 256 spp (VERDICT r03 task 7).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run3
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
echo "== leaf loop on leaf-8 trees"
for ll in 2 3 4 6 8; do
  run veach_leaf8_ll$ll "TRT_LEAF_LOOP=$ll" --scene veach-mis --steps 2 --leaf 8
  run stair_leaf8_ll$ll "TRT_LEAF_LOOP=$ll" --scene staircase --spp 64 --steps 2 --leaf 8
  run blob2m_leaf8_ll$ll "TRT_LEAF_LOOP=$ll" --scene blob --tris 2000000 --spp 64 --steps 2 --leaf 8
done
run stair_leaf2_ll2 "TRT_LEAF_LOOP=2" --scene staircase --spp 64 --steps 2
run stair_leaf2_ll3 "TRT_LEAF_LOOP=3" --scene staircase --spp 64 --steps 2
echo "== scheduler weights on leaf-8 trees (node step iff in_w * lanes at nodes >= lf_w * lanes at leaves)"
for w in 1:1 2:3 1:2 3:2; do
  run veach_leaf8_w$w "TRT_SCHED_W=$w" --scene veach-mis --steps 2 --leaf 8
  run stair_leaf8_w$w "TRT_SCHED_W=$w" --scene staircase --spp 64 --steps 2 --leaf 8
done
echo "== rocprof: veach-mis and staircase at 1080p / 256 spp"
tools/prof.sh r04_veach --scene veach-mis --steps 2 --warmup 1 --no-extra > $out/prof_veach.log 2>&1; echo "prof veach rc $?"
tools/prof.sh r04_stair --scene staircase --steps 2 --warmup 1 --no-extra > $out/prof_stair.log 2>&1; echo "prof stair rc $?"
python tools/pmc_summary.py gpurun_out/prof_r04_veach r04_veach-mis_1080p_256spp veach-mis 1080 256 && cp profiles/r04_veach-mis_1080p_256spp_pmc_hbm_bytes.csv profiles/hbm_traffic_veach-mis_1080p_256spp.json $out/
python tools/pmc_summary.py gpurun_out/prof_r04_stair r04_staircase_1080p_256spp staircase 1080 256 && cp profiles/r04_staircase_1080p_256spp_pmc_hbm_bytes.csv profiles/hbm_traffic_staircase_1080p_256spp.json $out/
for t in veach stair; do f=$(find gpurun_out/prof_r04_$t/stats -name "*kernel_stats.csv" | head -1); test -n "$f" && cp $f $out/r04_${t}_kernel_stats.csv; cp gpurun_out/prof_r04_$t/bench_under_stats.json $out/r04_${t}_bench_under_rocprof_stats.json; done
ls $out | head -80
