#!/bin/bash
# Round 4, GPU call 32 (the script of call 26 on the final form of the special-ray handling): after "zero direction components take the literal slab test" and "hints end at the reference's infinity": the new regression tests first
# (own timeout), the -m gpu suite, random scenes through the kernels, the parity soak, the default bench (did the hot kernels keep their speed?).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run32
mkdir -p $out
export PYTHONUNBUFFERED=1
echo "== new tests"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "zero_direction or hostile or non_finite" 2>&1 | tee $out/new.log | tail -5 || exit 1
echo "== pytest -m gpu"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tee $out/pytest_gpu.log | tail -3 || exit 1
echo "== random scenes"
timeout -k 10 400 python tools/fuzz_scenes.py --gpu --seconds 240 --seed 21 > $out/fuzz_scenes.txt 2> $out/fuzz_scenes.err; echo "rc $?"; tail -2 $out/fuzz_scenes.txt
echo "== soak"
timeout -k 10 300 python tools/fuzz_parity.py 150 12 > $out/fuzz_soak5.txt 2> $out/fuzz_soak5.err; echo "rc $?"; tail -1 $out/fuzz_soak5.txt
echo "== bench"
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err; echo "rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/run32/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("with_pass_overlap"), d["timing"] if "timing" in d else "")
for e in d.get("extra", []): print("  ", e.get("scene"), e.get("value"), e.get("ms_per_step"))
PY
