#!/bin/bash
# Round 4, GPU call 24: after "no distance culling on trees whose boxes do not nest": the -m gpu suite, a parity soak that now includes hostile trees, the default bench.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run24
mkdir -p $out
export PYTHONUNBUFFERED=1
echo "== non-finite and foreign"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "non_finite or foreign" 2>&1 | tee $out/poison.log | tail -5 || exit 1
echo "== pytest -m gpu"
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tee $out/pytest_gpu.log | tail -3 || exit 1
echo "== soak"
timeout -k 10 400 python tools/fuzz_parity.py 240 11 > $out/fuzz_soak4.txt 2> $out/fuzz_soak4.err; echo "rc $?"; tail -2 $out/fuzz_soak4.txt
echo "== bench"
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err; echo "rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/run24/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("with_pass_overlap"), d["roofline"].get("traffic"))
PY
