#!/bin/bash
# Round 4, GPU call 10: the GPU builder with LARGE triangles as clusters of their own and smaller clusters on scenes below 4 M triangles:
# its tests, then quality against the host SAH tree (defaults, and the threshold K of "large": box half-area > bounds' / K).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run10
mkdir -p $out
export PYTHONUNBUFFERED=1
echo "== tests/test_gpu_lbvh.py"
timeout -k 10 900 python -m pytest tests/test_gpu_lbvh.py -m gpu -q -x 2>&1 | tail -3
test ${PIPESTATUS[0]} -eq 0 || exit 1
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/run10/quality.txt
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import lbvh_cluster_sweep as L
for spec in ["staircase", "veach-mis", "blob:150000", "blob:2000000", "soup:1000000", "blob:10000000"]:
    name, _, n = spec.partition(":")
    n = int(n) if n else None
    v0, t0, m0, _, nt = L.measure(name, n, "auto", None)
    print(f"{name} ({nt} triangles): host SAH: {v0:.2f} visits {t0:.2f} tests per ray, {m0:.0f} Mrays/s", flush=True)
    for K in (None, "0", "256", "1024", "16384"):
        if K is None: os.environ.pop("TRT_LBVH_LARGE", None)
        else: os.environ["TRT_LBVH_LARGE"] = K
        v, t, m, b, _ = L.measure(name, n, "lbvh", None)
        print(f"   large K {'4096 (default)' if K is None else K:>15}: visits {v:6.2f} ({(v / v0 - 1) * 100:+5.1f} %)  tests {t:6.2f} ({(t / t0 - 1) * 100:+5.1f} %)  {m:6.0f} Mrays/s ({(m / m0 - 1) * 100:+5.1f} %)  build: device {b[0]:.1f} ms, call {b[1]:.1f} ms", flush=True)
    os.environ.pop("TRT_LBVH_LARGE", None)
PY
echo "== rays per second at full size, host SAH against the GPU builder"
for sc in "staircase --spp 64" "veach-mis --spp 64" "blob --tris 2000000 --spp 64"; do
  for b in auto lbvh; do
    timeout -k 10 400 python bench.py --scene $sc --steps 3 --builder $b --no-cpu-baseline --no-extra --no-overlap-extra > $out/q.json 2>$out/q.err
    python - "$sc $b" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r04/run10/q.json")); u = d["simd_utilisation_traversal"]
print(f'{sys.argv[1]:40s} {d["value"]:9.1f} Mrays/s {d["ms_per_step"]:9.2f} ms  visits {u["visits_per_ray"]} tests {u["tri_tests_per_ray"]}', flush=True)
PY
  done
done
