#!/bin/bash
# variants x workloads matrix on the GPU box; prints one line per run as it goes.
V=tinyraytracing_amd/lib/variants
for name in "$@"; do
  lib=$V/libtrt_hip_$name.so
  ok=$(TRT_HIP_LIB=$lib timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or incoherent or soup_deep" 2>&1 | tail -1)
  echo "## $name parity: $ok"
  tools/ab.sh "$name back|TRT_HIP_LIB=$lib|--steps 2" "$name veach l4|TRT_HIP_LIB=$lib|--scene veach-mis --steps 1 --leaf 4" "$name stair l4|TRT_HIP_LIB=$lib|--scene staircase --spp 64 --steps 1 --leaf 4" "$name soup l4|TRT_HIP_LIB=$lib|--scene soup --spp 16 --steps 1 --leaf 4"
done
