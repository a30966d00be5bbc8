#!/bin/bash
# Round 4, GPU call 13: the GPU builder with clusters of 4 / 8 triangles on the small scenes (the CPU emulation of tools/lbvh_study.py says the loss of staircase
# sits between groups of 8 and groups of 16 Morton-adjacent triangles).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
out=gpurun_out/r04/run13
mkdir -p $out
export PYTHONUNBUFFERED=1
python - <<'PY' 2>&1 | grep --line-buffered -v amdgpu.ids | tee gpurun_out/r04/run13/small_clusters.txt
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import lbvh_cluster_sweep as L
for spec in ["staircase", "veach-mis", "blob:150000", "blob:60000", "soup:50000"]:
    name, _, n = spec.partition(":")
    n = int(n) if n else None
    v0, t0, m0, _, nt = L.measure(name, n, "auto", None)
    print(f"{name} ({nt} triangles): host SAH: {v0:.2f} visits {t0:.2f} tests per ray, {m0:.0f} Mrays/s", flush=True)
    for cl in (None, 2, 4, 8, 12, 16):
        v, t, m, b, _ = L.measure(name, n, "lbvh", cl)
        print(f"   cluster {'default' if cl is None else cl:>7}: visits {v:6.2f} ({(v / v0 - 1) * 100:+5.1f} %)  tests {t:6.2f} ({(t / t0 - 1) * 100:+5.1f} %)  {m:6.0f} Mrays/s ({(m / m0 - 1) * 100:+5.1f} %)  build: device {b[0]:.1f} ms, call {b[1]:.1f} ms", flush=True)
PY
