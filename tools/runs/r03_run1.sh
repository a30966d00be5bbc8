#!/bin/bash
# Round 3, GPU call 1: oct-kernel smoke (short timeout), the whole -m gpu suite, node-kind A/B, bench rehearsals.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out/r03
export PYTHONUNBUFFERED=1
echo "== smoke (oct nodes, veach-mis)"; 
TRT_NODE_KIND=1 TRT_DEBUG=1 timeout -k 5 180 python - <<'PY' 2>&1 | tee gpurun_out/r03/smoke_oct.log || exit 1
import sys, numpy as np
sys.path.insert(0, "tests")
import oracle_lib as O, tinyraytracing_amd as T
for name in ("veach-mis", "staircase"):
    s = T.Scene.named(name, 96, 54)
    p = T.make_params(96, 54, 4, 7, flags=T.TRT_FLAG_COUNT)
    r = T.Renderer(s, 0)
    img, st = r.render(p)
    ref, ost = O.render(s.flat, p)
    print(name, "node bytes", st.inner_node_bytes, "equal", np.array_equal(img, ref), "rays", st.rays, ost.rays, "redo", st.redo_rays, "visits/ray", (st.inner_visits[0] + st.inner_visits[1]) / st.rays, flush=True)
    assert np.array_equal(img, ref)
PY
echo "== pytest -m gpu"
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 | tee gpurun_out/r03/pytest_gpu.log
test ${PIPESTATUS[0]} -eq 0 || exit 1
echo "== node kind A/B"
timeout -k 10 900 tools/ab_nodes.sh 2>&1 | tee gpurun_out/r03/ab_nodes.log
echo "== bench --gpus 2 rehearsal over gloo on one GPU (self-launch)"
TRT_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r03/bench_gpus2_gloo.json 2> gpurun_out/r03/bench_gpus2_gloo.err; echo "rc $?"; tail -c 600 gpurun_out/r03/bench_gpus2_gloo.json
echo "== bench --group 2 / headline"
timeout -k 10 300 python bench.py --group 2 --steps 5 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/r03/bench_group2.json 2> gpurun_out/r03/bench_group2.err; echo "rc $?"
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/r03/bench_back.json 2> gpurun_out/r03/bench_back.err; echo "rc $?"
python - <<'PY'
import json
for f in ("bench_group2", "bench_back", "bench_gpus2_gloo"):
    try:
        d = json.loads(open(f"gpurun_out/r03/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], "Mrays/s", d["ms_per_step"], "ms", {k: v["ms_per_step"] for k, v in d["kernels_rank0"].items() if v["ms_per_step"]}, "overlap", d.get("with_pass_overlap"))
    except Exception as e:
        print(f, "no result", e)
PY
