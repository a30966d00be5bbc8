#!/bin/bash
# Round 4, GPU call 14: the final sources after a clean rebuild — smoke(), the whole -m gpu suite, the default bench line.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run14
mkdir -p $out
export PYTHONUNBUFFERED=1
echo "== smoke"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
echo "== pytest -m gpu"
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -s > $out/pytest_gpu.log 2>&1 || { tail -30 $out/pytest_gpu.log; exit 1; }
tail -2 $out/pytest_gpu.log; grep "staircase: visits\|blob-150k\|config 5, tree" $out/pytest_gpu.log | cut -c1-250
echo "== default bench"
timeout -k 10 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/run14/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], "overlap", d.get("with_pass_overlap"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "traffic", d["roofline"].get("traffic"), d["roofline"].get("traffic_note", "")[-8:])
print({k: (v.get("achieved_GBps"), v.get("traffic_GBps")) for k, v in d["roofline"]["by_kernel"].items()})
for e in d.get("extra_workloads") or []:
    print("   extra", e["config"]["scene"], e["value"], "Mrays/s", e["ms_per_step"], "ms")
PY
