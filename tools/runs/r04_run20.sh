#!/bin/bash
# Round 4, GPU call 20: TRT_FLAG_SPECULAR_KS on the product path — parity with the oracle, the HIP render against the reference's staircase snapshots at their
# noise floor, the whole -m gpu suite, the headline with the new (off) flag in the kernels, a short soak with the flag in the mix.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run20
mkdir -p $out
export PYTHONUNBUFFERED=1
echo "== the flag"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_png.py -m gpu -q -x -s -k "specular_ks" 2>&1 | grep -v amdgpu.ids | tail -8
test ${PIPESTATUS[0]} -eq 0 || exit 1
echo "== pytest -m gpu"
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1 || { tail -30 $out/pytest_gpu.log; exit 1; }
tail -2 $out/pytest_gpu.log
echo "== headline and extras"
timeout -k 10 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/run20/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels_rank0"].items()})
for e in d.get("extra_workloads") or []:
    print("   extra", e["config"]["scene"], e["value"], "Mrays/s", e["ms_per_step"], "ms")
PY
echo "== soak"
timeout -k 10 400 python tools/fuzz_parity.py 240 67 > $out/fuzz_soak3.txt 2> $out/fuzz_soak3.err; echo "rc $?"; tail -2 $out/fuzz_soak3.txt
