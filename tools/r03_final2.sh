#!/bin/bash
# Round 3, closing GPU call: the whole -m gpu suite (with the LBVH tests), the default bench line against the committed traffic files, the multi-rank
# rehearsals, BASELINE.md's table (tools/baseline_table.sh), start-up cost of config 5 with both builders.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/r03f
export PYTHONUNBUFFERED=1
echo "== pytest -m gpu"
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r03f/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/r03f/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r03f/pytest_gpu.log
echo "== default bench"
timeout -k 10 600 python bench.py > gpurun_out/r03f/r03_bench_default.json 2> gpurun_out/r03f/r03_bench_default.err; echo "rc $?"
TRT_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r03f/r03_bench_gpus2_gloo_one_gpu.json 2> gpurun_out/r03f/r03_bench_gpus2_gloo_one_gpu.err; echo "rc $?"
timeout -k 10 300 python bench.py --group 2 --steps 5 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/r03f/r03_bench_group2_one_gpu.json 2> gpurun_out/r03f/r03_bench_group2.err; echo "rc $?"
timeout -k 10 300 python bench.py --group 1 --steps 5 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/r03f/r03_bench_group1_one_gpu.json 2> gpurun_out/r03f/r03_bench_group1.err; echo "rc $?"
python - <<'PY'
import json
for f in ("r03_bench_default", "r03_bench_gpus2_gloo_one_gpu", "r03_bench_group2_one_gpu", "r03_bench_group1_one_gpu"):
    try:
        d = json.loads(open(f"gpurun_out/r03f/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], "Mrays/s", d["ms_per_step"], "ms", {k: v["ms_per_step"] for k, v in d["kernels_rank0"].items() if v["ms_per_step"]}, "overlap", d.get("with_pass_overlap"), "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
        for e in d.get("extra_workloads") or []:
            print("   extra", e["config"]["scene"], e["config"]["spp"], "spp", e["value"], "Mrays/s", e["ms_per_step"], "ms", e["roofline"]["kernel"], e["roofline"]["frac"], (e["roofline"].get("traffic_from_profiles") or {}).get("frac"))
    except Exception as e:
        print(f, "no result", e)
PY
echo "== start-up cost of config 5"
timeout -k 10 400 python tools/create_cost.py 10000000 2>&1 | grep -v amdgpu.ids | grep -v "^trt_create:" > gpurun_out/r03f/create_cost2.log; cat gpurun_out/r03f/create_cost2.log
timeout -k 10 400 python tools/lbvh_cost.py blob:10000000:3840:2160:16 2>&1 | grep -v amdgpu.ids > gpurun_out/r03f/lbvh_cost2.log; cat gpurun_out/r03f/lbvh_cost2.log
echo "== BASELINE.md table"
timeout -k 10 1100 bash tools/baseline_table.sh > gpurun_out/r03f/baseline_table.md 2> gpurun_out/r03f/baseline_table.err; cat gpurun_out/r03f/baseline_table.md
