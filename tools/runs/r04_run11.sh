#!/bin/bash
# Round 4, GPU call 11: the GPU builder's tests after the cluster rule; the default bench command with its live PMC passes (roofline.traffic);
# a parity soak on the final libraries (a quarter of the renders on leaf-8 trees, a fifth on trees of the GPU builder).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run11
mkdir -p $out
export PYTHONUNBUFFERED=1
echo "== tests/test_gpu_lbvh.py"
timeout -k 10 900 python -m pytest tests/test_gpu_lbvh.py -m gpu -q -x -s 2>&1 | grep -v amdgpu.ids | tail -8
test ${PIPESTATUS[0]} -eq 0 || exit 1
echo "== default bench (with live traffic)"
timeout -k 10 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/run11/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], "overlap", d.get("with_pass_overlap"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
print(json.dumps(d["roofline"], indent=0)[:1500])
for e in d.get("extra_workloads") or []:
    print("   extra", e["config"]["scene"], e["config"]["spp"], "spp", e["value"], "Mrays/s", e["ms_per_step"], "ms", e["roofline"]["kernel"], e["roofline"]["frac"], (e["roofline"].get("traffic_from_profiles") or {}).get("frac"))
PY
echo "== fuzz soak"
timeout -k 10 700 python tools/fuzz_parity.py 540 41 2>&1 | grep -v amdgpu.ids | tee $out/fuzz_soak.txt | tail -4
