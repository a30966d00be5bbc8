#!/bin/bash
# Round 3, GPU call 5: the whole -m gpu suite, the default bench line, the multi-rank rehearsals, rocprof kernel stats + FETCH/WRITE passes of the headline.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/r03
export PYTHONUNBUFFERED=1
echo "== pytest -m gpu"
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
test ${PIPESTATUS[0]} -eq 0 || exit 1
echo "== default bench"
timeout -k 10 600 python bench.py > gpurun_out/r03/r03_bench_default.json 2> gpurun_out/r03/r03_bench_default.err; echo "rc $?"
echo "== bench --gpus 2 (self-launched, gloo on one GPU) and --group 2"
TRT_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r03/r03_bench_gpus2_gloo_one_gpu.json 2> gpurun_out/r03/r03_bench_gpus2_gloo_one_gpu.err; echo "rc $?"
timeout -k 10 300 python bench.py --group 2 --steps 5 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/r03/r03_bench_group2_one_gpu.json 2> gpurun_out/r03/r03_bench_group2.err; echo "rc $?"
timeout -k 10 300 python bench.py --group 1 --steps 5 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/r03/r03_bench_group1_one_gpu.json 2> gpurun_out/r03/r03_bench_group1.err; echo "rc $?"
python - <<'PY'
import json
for f in ("r03_bench_default", "r03_bench_gpus2_gloo_one_gpu", "r03_bench_group2_one_gpu", "r03_bench_group1_one_gpu"):
    try:
        d = json.loads(open(f"gpurun_out/r03/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], "Mrays/s", d["ms_per_step"], "ms", {k: v["ms_per_step"] for k, v in d["kernels_rank0"].items() if v["ms_per_step"]}, "overlap", d.get("with_pass_overlap"), "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
        for e in d.get("extra_workloads") or []:
            print("   extra", e["config"]["scene"], e["config"]["spp"], "spp", e["value"], "Mrays/s", e["ms_per_step"], "ms", e["roofline"]["kernel"], e["roofline"]["frac"], {k: v["ms_per_step"] for k, v in e["kernels_rank0"].items() if v["ms_per_step"]})
    except Exception as e:
        print(f, "no result", e)
PY
echo "== rocprof of the headline"
tools/prof.sh r03_back --steps 3 --warmup 1 --no-extra > gpurun_out/r03/prof_back.log 2>&1; echo "prof rc $?"
