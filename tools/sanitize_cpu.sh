#!/bin/bash
# AddressSanitizer + UBSan over all CPU-side code (host loaders/builder/JPEG/PNG, oracle, and the device path
# functions compiled for the CPU by tests/hostsim) under the whole `-m "not gpu"` test suite.
# GPU ASan is not available on the pool; the kernels' per-lane code is exactly what hostsim compiles.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root"
san=/tmp/trt_san; mkdir -p $san/bak
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1 -std=c++17 -fPIC -fopenmp -ffp-contract=off -march=x86-64-v3 -Iinclude"
g++ $SAN -Itinyraytracing_amd/host -shared -o $san/libtrt_host.so tinyraytracing_amd/host/{scene,bvh,synth,image_out,jpeg,png,capi}.cpp
g++ $SAN -shared -o $san/liboracle.so oracle/oracle.cpp oracle/oracle_literal.cpp
g++ $SAN -Itinyraytracing_amd/csrc -shared -o $san/libhostsim.so tests/hostsim/hostsim.cpp
cp tinyraytracing_amd/lib/libtrt_host.so oracle/liboracle.so tests/hostsim/libhostsim.so $san/bak/
restore() { cp $san/bak/libtrt_host.so tinyraytracing_amd/lib/; cp $san/bak/liboracle.so oracle/; cp $san/bak/libhostsim.so tests/hostsim/; }
trap restore EXIT
cp $san/libtrt_host.so tinyraytracing_amd/lib/; cp $san/liboracle.so oracle/; cp $san/libhostsim.so tests/hostsim/
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  UBSAN_OPTIONS=print_stacktrace=1 OMP_NUM_THREADS=4 python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider ${SAN_PYTEST_ARGS:-} 2>&1 | tee $san/log.txt | tail -5; grep -q " failed" $san/log.txt && { echo "tests failed under the sanitizers"; exit 1; } || true
if grep -q "runtime error" $san/log.txt; then echo "UBSan findings:"; grep "runtime error" $san/log.txt | sort | uniq -c; exit 1; fi
echo "sanitizers: clean"
