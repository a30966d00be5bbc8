# Builds everything in-tree (the .so files travel to the GPU box with the snapshot).
#   libtrt_hip.so   — the C-ABI hot path (include/trt.h), HIP for gfx950
#   libtrt_lbvh.so  — GPU BVH builder (include/trt_build.h), HIP for gfx950; not needed by the render path
#   libtrt_host.so  — loaders / BVH / PNG (include/trt_host.h), plain C++
#   tinyrt          — CLI: loaders -> render() -> PNG (the reference's main())
#   oracle/liboracle.so — CPU restatement used by tests/ and bench.py only
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
PKG := tinyraytracing_amd
OUT := $(PKG)/lib

# Bit-parity rule (include/trt_prims.h): no FP contraction on either side;
# FMA only where the source says fmaf.  x86-64-v3 = AVX2 + hardware FMA.
CPU_FP := -ffp-contract=off -march=x86-64-v3
CXXFLAGS := -O2 -g0 -std=c++17 -fPIC -Wall -Wextra $(CPU_FP) -Iinclude -I$(PKG)/host
# The LLVM atomic optimizer turns the one-lane queue reservation of k_shade (blockStage) into "atomic; s_waitcnt vmcnt(0);
# readfirstlane", i.e. waits for the round trip (and every store before it) on the spot; the kernel consumes the value one
# stage later.  No kernel here relies on that pass: every atomic is issued by one lane per wave already.
# -fno-slp-vectorize: left to itself the SLP vectoriser pairs the scalar fp32 operations of the triangle test and of the shading
# code into v_pk_* instructions; on gfx950 a v_pk_mul/add costs what two v_mul/add cost (tools/valu_probe.hip: 4.6 against
# 2 x 2.4 clocks per wave) and the pairs have to be assembled with v_mov first: trace_closest on `back` -9 %, k_shade -3 %.
# (The explicit two-wide code of innerStep() is not affected: its operands are loaded as pairs.)
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
            -mllvm -amdgpu-atomic-optimizer-strategy=None \
            -Wall -Wextra -Wno-unused-parameter -Iinclude -I$(PKG)/csrc

HOST_SRC := $(PKG)/host/scene.cpp $(PKG)/host/bvh.cpp $(PKG)/host/synth.cpp $(PKG)/host/image_out.cpp $(PKG)/host/jpeg.cpp $(PKG)/host/png.cpp $(PKG)/host/capi.cpp
HOST_HDR := $(wildcard $(PKG)/host/*.h) include/trt.h include/trt_host.h include/trt_prims.h
HIP_SRC := $(PKG)/csrc/trt_api.hip
HIP_HDR := $(wildcard $(PKG)/csrc/*.h) include/trt.h include/trt_prims.h include/trt_exact.h

.PHONY: all host hip lbvh oracle cli hostsim variants probe exactcheck clean
all: host hip lbvh oracle hostsim cli exactcheck

host: $(OUT)/libtrt_host.so
hip: $(OUT)/libtrt_hip.so
lbvh: $(OUT)/libtrt_lbvh.so
cli: $(OUT)/tinyrt
oracle:
	$(MAKE) -C oracle
# CPU compile of the device-side path functions, for tests only (tests/hostsim)
hostsim: tests/hostsim/libhostsim.so
tests/hostsim/libhostsim.so: tests/hostsim/hostsim.cpp $(HIP_HDR)
	$(CXX) $(CXXFLAGS) -fopenmp -I$(PKG)/csrc -shared -o $@ tests/hostsim/hostsim.cpp

$(OUT)/libtrt_host.so: $(HOST_SRC) $(HOST_HDR)
	@mkdir -p $(OUT)
	$(CXX) $(CXXFLAGS) -fopenmp -shared -o $@ $(HOST_SRC)

$(OUT)/libtrt_hip.so: $(HIP_SRC) $(HIP_HDR)
	@mkdir -p $(OUT)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_SRC)

# -ffp-contract=off like every other library here: contraction would only move Morton codes and SAH costs, i.e. topology, but the tree is part of
# the checkpoint's scene hash and should not depend on the compiler's choice of fused operations.
$(OUT)/libtrt_lbvh.so: $(PKG)/csrc/trt_lbvh.hip include/trt.h include/trt_build.h
	@mkdir -p $(OUT)
	$(HIPCC) -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wextra -Wno-unused-parameter -Iinclude -shared -o $@ $(PKG)/csrc/trt_lbvh.hip

# tinyrt does not link libtrt_lbvh.so: render() dlopens it when --gpu-bvh asks for the device builder (include/trt_build.h)
$(OUT)/tinyrt: $(PKG)/host/main.cpp $(PKG)/host/render.cpp $(HOST_HDR) include/trt_build.h $(OUT)/libtrt_host.so $(OUT)/libtrt_hip.so
	$(CXX) $(CXXFLAGS) -o $@ $(PKG)/host/main.cpp $(PKG)/host/render.cpp -L$(OUT) -ltrt_host -ltrt_hip -fopenmp -ldl -Wl,-rpath,'$$ORIGIN'

# A/B builds of the HIP library for tuning on the GPU box: TRT_HIP_LIB=<path> selects one at run time.
# name=defines, "+" separating the -D options
# (round 4's k_shade occupancy arms — block sizes 256 / 512 / 640 / 768 at 4, 5 and 6 waves per SIMD — are on record in profiles/r04_ab_shade_tail.txt; what
# won is in the default build: TRT_SHADE1_* / TRT_SHADEN_* in trt_kernels.h.  shade_old = round 3's configuration for the A/B.)
VARIANTS := w8=-DTRT_TRACE_MINWAVES=8 nt0=-DTRT_NT=0 nt15=-DTRT_NT=15 shade_old=-DTRT_SHADE1_BLOCK=512+-DTRT_SHADE1_WAVES=4+-DTRT_SHADEN_BLOCK=512+-DTRT_SHADEN_WAVES=4
variants: $(HIP_SRC) $(HIP_HDR)
	@mkdir -p $(OUT)/variants
	@for v in $(VARIANTS); do name=$${v%%=*}; defs=$$(echo "$${v#*=}" | tr '+' ' '); \
	  echo "variant $$name: $$defs"; $(HIPCC) $(HIPFLAGS) $$defs -shared -o $(OUT)/variants/libtrt_hip_$$name.so $(HIP_SRC) || exit 1; done

clean:
	rm -rf $(OUT) tests/hostsim/libhostsim.so
	$(MAKE) -C oracle clean

# exhaustive (2^32 inputs) proof that include/trt_exact.h returns the bits of sqrtf / 1.0f / sqrtf: run by tests/test_gpu_parity.py
exactcheck: tools/exact_unary_check
tools/exact_unary_check: tools/exact_unary_check.hip include/trt_exact.h include/trt_prims.h
	$(HIPCC) -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Iinclude -o $@ tools/exact_unary_check.hip

# chip-ceiling probe + TCC counter calibration workload (tools/calibrate_counters.sh runs it on the GPU box)
probe: tools/gather_probe
tools/gather_probe: tools/gather_probe.hip
	$(HIPCC) -O3 --offload-arch=gfx950 -o $@ tools/gather_probe.hip
