#!/bin/bash
# Round 4, GPU call 37: the final library (non-temporal queue records in k_shade): -m gpu suite, smoke(), a soak, random scenes, the default bench for the record.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run37
mkdir -p $out
export PYTHONUNBUFFERED=1
echo "== pytest -m gpu"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tee $out/pytest_gpu.log | tail -3 || exit 1
echo "== smoke"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
echo "== soak"
timeout -k 10 300 python tools/fuzz_parity.py 120 77 > $out/fuzz_soak7.txt 2> $out/fuzz_soak7.err; echo "rc $?"; tail -1 $out/fuzz_soak7.txt
echo "== random scenes"
timeout -k 10 300 python tools/fuzz_scenes.py --gpu --seconds 120 --seed 123 > $out/fuzz_scenes.txt 2> $out/fuzz_scenes.err; echo "rc $?"; tail -1 $out/fuzz_scenes.txt
echo "== bench"
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err; echo "rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/run37/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("with_pass_overlap"), {k: v["ms_per_step"] for k, v in d["kernels_rank0"].items()} if isinstance(d.get("kernels_rank0"), dict) else "")
for e in d.get("extra_workloads", []): print("  ", e.get("config", {}).get("workload", "")[:40], e.get("value"), e.get("ms_per_step"))
PY
