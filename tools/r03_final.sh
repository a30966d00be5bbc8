#!/bin/bash
# Round 3, final artifacts on the GPU box (one gpurun call): the -m gpu suite, the default bench line, the multi-rank rehearsals, rocprof kernel
# stats + FETCH/WRITE passes of the headline, the counter tables of staircase / veach-mis / config 3 / config 5, start-up cost of config 5.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/r03f
export PYTHONUNBUFFERED=1
echo "== pytest -m gpu"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r03f/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/r03f/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r03f/pytest_gpu.log
echo "== default bench"
timeout -k 10 600 python bench.py > gpurun_out/r03f/r03_bench_default.json 2> gpurun_out/r03f/r03_bench_default.err; echo "rc $?"
echo "== bench --gpus 2 (self-launched, gloo on one GPU) and --group 1 / 2"
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
            print("   extra", e["config"]["scene"], e["config"]["spp"], "spp", e["value"], "Mrays/s", e["ms_per_step"], "ms", e["roofline"]["kernel"], e["roofline"]["frac"], {k: v["ms_per_step"] for k, v in e["kernels_rank0"].items() if v["ms_per_step"]})
    except Exception as e:
        print(f, "no result", e)
PY
echo "== rocprof of the headline"
tools/prof.sh r03_back --steps 3 --warmup 1 --no-extra > gpurun_out/r03f/prof_back.log 2>&1; echo "prof rc $?"
echo "== counter tables"
tools/roofs.sh r03_stair "--scene staircase --spp 64" > gpurun_out/r03f/roofs_stair.log 2>&1; echo "roofs stair rc $?"
tools/roofs.sh r03_veach "--scene veach-mis --spp 64" > gpurun_out/r03f/roofs_veach.log 2>&1; echo "roofs veach rc $?"
tools/roofs.sh r03_soup_64spp "--scene soup --spp 64" > gpurun_out/r03f/roofs_soup.log 2>&1; echo "roofs soup rc $?"
tools/roofs.sh r03_blob10m_64spp "--scene blob --tris 10000000 --width 3840 --height 2160 --spp 64" > gpurun_out/r03f/roofs_blob10m.log 2>&1; echo "roofs blob10m rc $?"
echo "== start-up cost of config 5"
timeout -k 10 400 python tools/create_cost.py 10000000 2>&1 | grep -v amdgpu.ids > gpurun_out/r03f/create_cost.log; cat gpurun_out/r03f/create_cost.log
