#!/bin/bash
# variants x workloads on the GPU box.  usage: tools/matrix.sh "<variant names>" ["label|bench args" ...]
# a variant name "default" means the in-tree libtrt_hip.so
V=tinyraytracing_amd/lib/variants
names=$1; shift
if [ $# -eq 0 ]; then set -- "back|--steps 3" "veach|--scene veach-mis --steps 2" "stair64|--scene staircase --spp 64 --steps 1" "soup16|--scene soup --spp 16 --steps 1"; fi
for name in $names; do
  if [ "$name" = default ]; then lib=tinyraytracing_amd/lib/libtrt_hip.so; else lib=$V/libtrt_hip_$name.so; fi
  for w in "$@"; do
    tools/ab.sh "$name ${w%%|*}|TRT_HIP_LIB=$lib|${w#*|}"
  done
done
