// gather_probe.hip — what a CU of gfx950 sustains for the access patterns of BVH traversal, and a known-byte-count
// workload to calibrate the rocprofv3 TCC counters on.  Stand-alone: no library code.
//
//   k_gather<LOADS>   every lane fetches LOADS consecutive 16-B words of a 128-B record whose index is a hash of
//                     (lane, iteration): the per-lane divergent gather of a wide-node visit (LOADS = 7), of a
//                     quantised node (4), of a triangle record (3), of one word (1).  `coherent` makes all 64 lanes
//                     of a wave pick the SAME record (what a wave of coherent rays does).
//   k_gather_lds      the same gather from a table resident in LDS (ds_read_b128 at per-lane addresses).
//   k_valu            independent v_fma_f32 chains: wave-instructions per clock and SIMD at the launch's occupancy.
//
// Output: one line per case with the known request/byte counts, so that `rocprofv3 --pmc FETCH_SIZE ...` of the same
// run can be divided by them (tools/calibrate_counters.sh).
//
// build: hipcc -O3 --offload-arch=gfx950 -o tools/gather_probe tools/gather_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } \
    } while (0)

struct alignas(16) f4 {
    float x, y, z, w;
};

__device__ inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

template <int LOADS>
__global__ __launch_bounds__(256) void k_gather(const f4* __restrict__ table, uint32_t n_records, uint32_t iters, uint32_t coherent, float* __restrict__ sink)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t key = coherent ? (gid >> 6) : gid;  // one record per wave, or per lane
    float acc = 0.f;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t rec = mix32(key * 0x9E3779B9u + it * 0x85EBCA6Bu) % n_records;
        const f4* p = table + (size_t)rec * 8;  // 128-B records
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            const f4 v = p[k];
            acc += v.x + v.w;
        }
    }
    if (acc == 123.456f) sink[gid] = acc;  // never true: keeps the loads
}

// dependent variant: the next record index comes out of the loaded data (pointer chasing, as a traversal does)
template <int LOADS>
__global__ __launch_bounds__(256) void k_chase(const f4* __restrict__ table, uint32_t n_records, uint32_t iters, float* __restrict__ sink)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t rec = mix32(gid) % n_records;
    float acc = 0.f;
    for (uint32_t it = 0; it < iters; ++it) {
        const f4* p = table + (size_t)rec * 8;
        uint32_t nxt = 0;
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            const f4 v = p[k];
            acc += v.x;
            nxt ^= __float_as_uint(v.w);
        }
        rec = mix32(nxt + gid + it) % n_records;
    }
    if (acc == 123.456f) sink[gid] = acc;
}


// Wave-cooperative fetch of 64 per-lane records (what a wide-node visit of 64 divergent rays needs): in load k, lane j
// requests 16-B piece (j & 7) of the record owned by lane 8 k + (j >> 3), so eight neighbouring lanes cover one 128-B line
// and the load is as coalesced as the hardware can see; the pieces go through LDS ([owner][piece], 112-B stride: conflict
// free for b128) to their owners.  PIECES = 7 (a WideNode).  `chase`: the next record index depends on the data.
__global__ __launch_bounds__(256) void k_coop(const f4* __restrict__ table, uint32_t n_records, uint32_t iters, uint32_t chase, float* __restrict__ sink)
{
    __shared__ f4 stage[4][64 * 7 + 8];  // + a dump for the eighth lane of every group (no branch around the loads)
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    f4* st = stage[wave];
    const uint32_t piece = lane & 7u, sub = lane >> 3;
    uint32_t rec = mix32(gid) % n_records;
    float acc = 0.f;
    for (uint32_t it = 0; it < iters; ++it) {
        f4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t r = (uint32_t)__shfl((int)rec, 8 * k + (int)sub);
            v[k] = table[(size_t)r * 8 + (piece < 7u ? piece : 6u)];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) st[piece < 7u ? (8 * k + sub) * 7 + piece : 64 * 7 + sub] = v[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t nxt = 0;
#pragma unroll
        for (int p = 0; p < 7; ++p) {
            const f4 w = st[lane * 7 + p];
            acc += w.x;
            nxt ^= __float_as_uint(w.w);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        rec = chase ? mix32(nxt + gid + it) % n_records : mix32(gid * 0x9E3779B9u + it * 0x85EBCA6Bu) % n_records;
    }
    if (acc == 123.456f) sink[gid] = acc;
}

template <int LOADS>
__global__ __launch_bounds__(256) void k_gather_lds(const f4* __restrict__ table, uint32_t n_records, uint32_t iters, uint32_t coherent, float* __restrict__ sink)
{
    extern __shared__ f4 lds[];
    for (uint32_t w = threadIdx.x; w < n_records * 8; w += blockDim.x) lds[w] = table[w];
    __syncthreads();
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t key = coherent ? (gid >> 6) : gid;
    float acc = 0.f;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t rec = mix32(key * 0x9E3779B9u + it * 0x85EBCA6Bu) % n_records;
        const f4* p = lds + (size_t)rec * 8;
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            const f4 v = p[k];
            acc += v.x + v.w;
        }
    }
    if (acc == 123.456f) sink[gid] = acc;
}

__global__ __launch_bounds__(256) void k_valu(uint32_t iters, float* __restrict__ sink)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 1e-9f;
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
            a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
        }
    }
    const float s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    if (s == 123.456f) sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 32-bit integer multiplies (the counter RNG of trt_prims.h spends five per draw) and IEEE divisions (43 per path vertex in k_shade)
__global__ __launch_bounds__(256) void k_imul(uint32_t iters, float* __restrict__ sink)
{
    uint32_t a0 = threadIdx.x | 1u, a1 = a0 + 2, a2 = a0 + 4, a3 = a0 + 6, a4 = a0 + 8, a5 = a0 + 10, a6 = a0 + 12, a7 = a0 + 14;
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // the xor keeps the compiler from folding the chain into one multiply by a power of the constant
            a0 = (a0 * 0x7feb352du) ^ it; a1 = (a1 * 0x846ca68bu) ^ it; a2 = (a2 * 0x7feb352du) ^ it; a3 = (a3 * 0x846ca68bu) ^ it;
            a4 = (a4 * 0x7feb352du) ^ it; a5 = (a5 * 0x846ca68bu) ^ it; a6 = (a6 * 0x7feb352du) ^ it; a7 = (a7 * 0x846ca68bu) ^ it;
        }
    }
    const uint32_t s = ((a0 ^ a1) ^ (a2 ^ a3)) ^ ((a4 ^ a5) ^ (a6 ^ a7));
    if (s == 123456u) sink[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
__global__ __launch_bounds__(256) void k_fdiv(uint32_t iters, float* __restrict__ sink)
{
    float a0 = 1.0f + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const float d = 1.0000001f + 1e-7f * threadIdx.x;
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { a0 = a0 / d; a1 = a1 / d; a2 = a2 / d; a3 = a3 / d; }
    }
    const float s = (a0 + a1) + (a2 + a3);
    if (s == 123.456f) sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double timeIt(hipEvent_t e0, hipEvent_t e1)
{
    float ms = 0.f;
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main(int argc, char** argv)
{
    const char* only = argc > 1 ? argv[1] : "";  // substring filter on the case name
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;  // Hz (nominal)
    std::printf("# device %s, %d CUs, nominal %.0f MHz\n", prop.gcnArchName, cus, clk / 1e6);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float* sink = nullptr;
    CK(hipMalloc(&sink, (size_t)1 << 26));

    const size_t max_bytes = (size_t)2 << 30;
    f4* table = nullptr;
    CK(hipMalloc(&table, max_bytes));
    {
        std::vector<uint32_t> h(max_bytes / 4);
        uint32_t s = 12345u;
        for (auto& w : h) { s = s * 1664525u + 1013904223u; w = (s >> 9) | 0x3f000000u; }  // floats in [0.5, 1)
        CK(hipMemcpy(table, h.data(), max_bytes, hipMemcpyHostToDevice));
    }
    const uint32_t blocks = (uint32_t)cus * 8;  // 8 blocks of 256 threads per CU = 8 waves per SIMD
    const uint32_t threads = blocks * 256;

    struct Size { const char* name; size_t bytes; };
    const Size sizes[] = {{"16KiB(L1)", (size_t)16 << 10}, {"2MiB(L2)", (size_t)2 << 20}, {"64MiB(MALL)", (size_t)64 << 20}, {"2GiB(HBM)", (size_t)2 << 30}};
    auto runGather = [&](int loads, const Size& sz, uint32_t coherent, bool chase) {
        char name[128];
        std::snprintf(name, sizeof name, "%s loads=%d table=%s %s", chase ? "chase" : "gather", loads, sz.name, coherent ? "coherent" : "divergent");
        if (only[0] && !std::strstr(name, only)) return;
        const uint32_t n_records = (uint32_t)(sz.bytes / 128);
        const uint32_t iters = chase ? 64 : 256;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (!chase) {
                if (loads == 1) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, table, n_records, iters, coherent, sink);
                else if (loads == 3) hipLaunchKernelGGL(k_gather<3>, dim3(blocks), dim3(256), 0, 0, table, n_records, iters, coherent, sink);
                else if (loads == 4) hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(256), 0, 0, table, n_records, iters, coherent, sink);
                else hipLaunchKernelGGL(k_gather<7>, dim3(blocks), dim3(256), 0, 0, table, n_records, iters, coherent, sink);
            } else {
                if (loads == 4) hipLaunchKernelGGL(k_chase<4>, dim3(blocks), dim3(256), 0, 0, table, n_records, iters, sink);
                else hipLaunchKernelGGL(k_chase<7>, dim3(blocks), dim3(256), 0, 0, table, n_records, iters, sink);
            }
            CK(hipEventRecord(e1));
            const double ms = timeIt(e0, e1);
            if (rep == 0) continue;
            const double lane_loads = (double)threads * iters * loads;
            const double bytes = lane_loads * 16.0;
            const double records = (double)threads * iters / (coherent ? 64.0 : 1.0);
            std::printf("%-52s %8.3f ms  %7.1f GB/s requested  %6.3f lane-loads/clk/CU  distinct 128-B records %.3e (%.3e B if each fetched once)\n", name, ms,
                        bytes / ms / 1e6, lane_loads / (ms * 1e-3 * clk) / cus, records, records * (loads > 4 ? 128.0 : 64.0));
        }
    };
    const int loadv[] = {1, 3, 4, 7};
    for (const Size& sz : sizes)
        for (int loads : loadv) {
            runGather(loads, sz, 0u, false);
            if (sz.bytes <= ((size_t)2 << 20)) runGather(loads, sz, 1u, false);
        }
    for (const Size& sz : sizes) { runGather(4, sz, 0u, true); runGather(7, sz, 0u, true); }


    // cooperative fetch against the per-lane gather / chase of 7 pieces, at the occupancy its LDS staging leaves (5 blocks per CU)
    for (const Size& sz : sizes)
        for (int mode = 0; mode < 3; ++mode) {  // 0: k_chase<7> at 5 waves/SIMD, 1: coop chase, 2: coop independent
            char name[128];
            std::snprintf(name, sizeof name, "%s table=%s waves/SIMD=5", mode == 0 ? "chase loads=7 (per lane)" : mode == 1 ? "coop chase pieces=7" : "coop gather pieces=7", sz.name);
            if (only[0] && !std::strstr(name, only)) continue;
            const uint32_t n_records = (uint32_t)(sz.bytes / 128), iters = 64, nb = (uint32_t)cus * 5;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_chase<7>, dim3(nb), dim3(256), 0, 0, table, n_records, iters, sink);
                else hipLaunchKernelGGL(k_coop, dim3(nb), dim3(256), 0, 0, table, n_records, iters, mode == 1 ? 1u : 0u, sink);
                CK(hipEventRecord(e1));
                const double ms = timeIt(e0, e1);
                if (rep == 0) continue;
                const double visits = (double)nb * 256 * iters;
                std::printf("%-52s %8.3f ms  %6.3f record visits/clk/CU  (%.1f clk per visit and CU)\n", name, ms, visits / (ms * 1e-3 * clk) / cus, (ms * 1e-3 * clk) * cus / visits);
            }
        }

    for (int coherent = 0; coherent < 2; ++coherent)
        for (int loads : {4, 7}) {
            char name[128];
            std::snprintf(name, sizeof name, "lds loads=%d table=32KiB %s", loads, coherent ? "coherent" : "divergent");
            if (only[0] && !std::strstr(name, only)) continue;
            const uint32_t n_records = 256, iters = 256;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (loads == 4) hipLaunchKernelGGL(k_gather_lds<4>, dim3(blocks / 2), dim3(256), 32768, 0, table, n_records, iters, (uint32_t)coherent, sink);
                else hipLaunchKernelGGL(k_gather_lds<7>, dim3(blocks / 2), dim3(256), 32768, 0, table, n_records, iters, (uint32_t)coherent, sink);
                CK(hipEventRecord(e1));
                const double ms = timeIt(e0, e1);
                if (rep == 0) continue;
                const double lane_loads = (double)(threads / 2) * iters * loads;
                std::printf("%-52s %8.3f ms  %7.1f GB/s requested  %6.3f lane-loads/clk/CU\n", name, ms, lane_loads * 16.0 / ms / 1e6, lane_loads / (ms * 1e-3 * clk) / cus);
            }
        }

    for (int bpc : {1, 2, 4, 8}) {  // blocks of 256 threads per CU = waves per SIMD
        char name[64];
        std::snprintf(name, sizeof name, "valu fma waves/SIMD=%d", bpc);
        if (only[0] && !std::strstr(name, only)) continue;
        const uint32_t iters = 4096;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_valu, dim3((uint32_t)cus * bpc), dim3(256), 0, 0, iters, sink);
            CK(hipEventRecord(e1));
            const double ms = timeIt(e0, e1);
            if (rep == 0) continue;
            const double wave_instr = (double)cus * bpc * 4 * iters * 64.0;
            std::printf("%-52s %8.3f ms  %6.3f wave-instr/clk/SIMD (nominal clock)  %7.1f TFLOP/s\n", name, ms, wave_instr / (ms * 1e-3 * clk) / (cus * 4.0),
                        wave_instr * 64 * 2 / ms / 1e9);
        }
    }
    for (int which = 0; which < 2; ++which) {
        const char* name = which == 0 ? "valu v_mul_lo_u32 waves/SIMD=8" : "valu IEEE fp32 division waves/SIMD=8";
        if (only[0] && !std::strstr(name, only)) continue;
        const uint32_t iters = which == 0 ? 2048 : 512;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (which == 0) hipLaunchKernelGGL(k_imul, dim3((uint32_t)cus * 8), dim3(256), 0, 0, iters, sink);
            else hipLaunchKernelGGL(k_fdiv, dim3((uint32_t)cus * 8), dim3(256), 0, 0, iters, sink);
            CK(hipEventRecord(e1));
            const double ms = timeIt(e0, e1);
            if (rep == 0) continue;
            const double ops = (double)cus * 8 * 4 * iters * (which == 0 ? 64.0 : 32.0);  // wave-level operations (imul: each paired with one v_xor)
            std::printf("%-52s %8.3f ms  %6.3f wave-ops/clk/SIMD (nominal clock) = one per %.1f clocks\n", name, ms, ops / (ms * 1e-3 * clk) / (cus * 4.0),
                        (ms * 1e-3 * clk) * (cus * 4.0) / ops);
        }
    }
    CK(hipDeviceSynchronize());
    return 0;
}
