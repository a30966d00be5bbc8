// trt_kernels.h — the wavefront kernels (gfx950, wave64).
//
// One render pass keeps N = pixels x samples-in-chunk paths in flight.  Path
// state lives in HBM as struct-of-float4-arrays (16 B per lane per load: the
// widest coalesced access, 1 KiB per wave instruction):
//   ray queue   ra = (o.x, o.y, o.z, d.x)   rb = (d.y, d.z, bits(path id), bits(meta))
//               bt = (beta.r, beta.g, beta.b, -)
//   hit buffer  (t, bits(tri), u, v)
//   shadow queue per light   sa = (o.xyz, d.x)  sb = (d.y, d.z, bits(path id), -)
//                            sw = (w.r, w.g, w.b, -)   w = beta * unoccluded contribution
//   accumulator Lacc[path id] = (L.r, L.g, L.b, -)
// Per bounce: trace_closest -> shade (emits <= 1 extension ray and <= 1 shadow
// ray per light, compacted with __ballot/popcount ranks + one atomic per block)
// -> trace_shadow per light (in light order, so every path's sum has a fixed
// order and the image is bit-reproducible).
//
// Traversal stack: per-lane stack of node references in LDS, laid out
// [level][lane] so a wave's accesses hit 64 consecutive banks; levels beyond
// TRT_LDS_STACK spill to a per-thread global area.
#pragma once
#include <hip/hip_runtime.h>

#include "trt_path.h"

namespace trtd {

constexpr int TRT_TRACE_BLOCK = 256;
constexpr int TRT_LDS_STACK = 24;       // 24 levels x 256 lanes x 4 B = 24 KiB per block
constexpr int TRT_SHADE_BLOCK = 512;
constexpr int TRT_MAX_LIGHTS = 8;

struct RayQueue {
    f4* ra;
    f4* rb;
    f4* bt;
};
struct ShadowQueue {
    f4* sa;
    f4* sb;
    f4* sw;
};

struct DeviceStats {
    unsigned long long inner_visits[2];
    unsigned long long tri_tests[2];
    unsigned long long shaded_hits;
    unsigned int max_depth_hit;
    unsigned int pad;
};

struct LdsStack {
    uint32_t* lds;    // &smem[threadIdx.x]
    uint32_t* spill;  // &spill[global thread id]
    uint32_t spill_stride;
    __device__ void push(int sp, uint32_t v)
    {
        if (sp < TRT_LDS_STACK) lds[sp * TRT_TRACE_BLOCK] = v;
        else spill[(size_t)(sp - TRT_LDS_STACK) * spill_stride] = v;
    }
    __device__ uint32_t pop(int sp) const
    {
        return sp < TRT_LDS_STACK ? lds[sp * TRT_TRACE_BLOCK] : spill[(size_t)(sp - TRT_LDS_STACK) * spill_stride];
    }
};

// Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a
// contiguous range of the queue so rays that are neighbours in the queue
// (spatially coherent) meet in one L2.  Speed only, never correctness.
__device__ inline uint32_t xcdSwizzle(uint32_t bid, uint32_t nblocks)
{
    return (nblocks & 7u) ? bid : (bid & 7u) * (nblocks >> 3) + (bid >> 3);
}

__device__ inline unsigned long long waveSum(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ---------------------------------------------------------------- K1 ----
// Primary rays: main.cpp:88-95 + Camera::getRay (camera.cpp:19-28).
// path id i = s_local * npix + pixel-in-tile: a wave covers 64 neighbouring pixels.

__global__ __launch_bounds__(256) void k_gen_primary(SceneDev sc, TileDesc td, RayQueue q, f4* __restrict__ Lacc, uint32_t s0, uint32_t n)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t s_local = i / td.npix, pl = i - s_local * td.npix;
        const uint32_t r = pl / (uint32_t)td.tile_w, c = pl - r * (uint32_t)td.tile_w;
        const int y = td.rows[r], x = td.x0 + (int)c;
        Stream rng;
        rng.key = trt_rng_make_key(td.seed, (uint32_t)y * (uint32_t)td.width + (uint32_t)x, s0 + s_local);
        rng.ctr = 0;
        const float u1 = rng.next(), u2 = rng.next();
        f3 o, d;
        cameraRay(sc.cam, td.width, td.height, y, x, u1, u2, o, d);
        q.ra[i] = mk4(o.x, o.y, o.z, d.x);
        q.rb[i] = mk4(d.y, d.z, u2f(i), u2f(packMeta(rng.ctr, TRT_META_CAMERA, 0)));
        q.bt[i] = mk4(1.0f, 1.0f, 1.0f, 0.0f);
        Lacc[i] = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

// ---------------------------------------------------------------- K2 ----
// traverseBVH (bvh.cpp:146-245) for every queued ray.
template <bool COUNT>
__global__ __launch_bounds__(TRT_TRACE_BLOCK) void k_trace_closest(SceneDev sc, const f4* __restrict__ ra, const f4* __restrict__ rb, f4* __restrict__ hit, uint32_t n,
                                                                   uint32_t* __restrict__ spill, uint32_t spill_stride, DeviceStats* stats)
{
    __shared__ uint32_t smem[TRT_LDS_STACK * TRT_TRACE_BLOCK];
    LdsStack stk;
    stk.lds = smem + threadIdx.x;
    stk.spill = spill + (size_t)blockIdx.x * TRT_TRACE_BLOCK + threadIdx.x;
    stk.spill_stride = spill_stride;
    uint32_t n_inner = 0, n_tri = 0;
    const uint32_t lb = xcdSwizzle(blockIdx.x, gridDim.x);
    const uint32_t stride = gridDim.x * TRT_TRACE_BLOCK;
    for (uint32_t i = lb * TRT_TRACE_BLOCK + threadIdx.x; i < n; i += stride) {
        const f4 a = ra[i], b = rb[i];
        const Hit h = traceClosest<LdsStack, COUNT>(sc, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), stk, n_inner, n_tri);
        hit[i] = mk4(h.t, u2f((uint32_t)h.tri), h.u, h.v);
    }
    if (COUNT) {
        const unsigned long long si = waveSum(n_inner), st = waveSum(n_tri);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&stats->inner_visits[0], si);
            atomicAdd(&stats->tri_tests[0], st);
        }
    }
}

// ---------------------------------------------------------------- K4 ----
// Shadow test of shade() (pathTracing.cpp:51-58): CLOSEST hit, visible iff
// its material is the light's (Q5); then L += w.  One launch per light, in
// light order; each path has at most one ray per launch, so the read-modify-
// write of Lacc needs no atomic and the sum order is fixed.
template <bool COUNT>
__global__ __launch_bounds__(TRT_TRACE_BLOCK) void k_trace_shadow(SceneDev sc, ShadowQueue sq, uint32_t n, uint32_t light_mat, f4* __restrict__ Lacc,
                                                                  uint32_t* __restrict__ spill, uint32_t spill_stride, DeviceStats* stats)
{
    __shared__ uint32_t smem[TRT_LDS_STACK * TRT_TRACE_BLOCK];
    LdsStack stk;
    stk.lds = smem + threadIdx.x;
    stk.spill = spill + (size_t)blockIdx.x * TRT_TRACE_BLOCK + threadIdx.x;
    stk.spill_stride = spill_stride;
    uint32_t n_inner = 0, n_tri = 0;
    const uint32_t lb = xcdSwizzle(blockIdx.x, gridDim.x);
    const uint32_t stride = gridDim.x * TRT_TRACE_BLOCK;
    for (uint32_t i = lb * TRT_TRACE_BLOCK + threadIdx.x; i < n; i += stride) {
        const f4 a = sq.sa[i], b = sq.sb[i];
        const Hit h = traceClosest<LdsStack, COUNT>(sc, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), stk, n_inner, n_tri);
        if (h.tri >= 0 && (h.flags >> 8) == light_mat) {
            const f4 w = sq.sw[i];
            const uint32_t pid = f2u(b.z);
            f4 L = Lacc[pid];
            L.x = L.x + w.x;
            L.y = L.y + w.y;
            L.z = L.z + w.z;
            Lacc[pid] = L;
        }
    }
    if (COUNT) {
        const unsigned long long si = waveSum(n_inner), st = waveSum(n_tri);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&stats->inner_visits[1], si);
            atomicAdd(&stats->tri_tests[1], st);
        }
    }
}

// Block-wide stream compaction slot: every thread calls it; threads with `flag`
// receive consecutive slots of the output queue.  Wave rank from
// __ballot/popcount, wave offsets through LDS, ONE global atomic per block.
// s_cnt must hold 2 * (NW + 1) words; `parity` alternates between calls so two
// barriers per call suffice.
template <int BLOCK>
__device__ inline uint32_t blockReserve(bool flag, uint32_t* counter, uint32_t* s_cnt, int parity)
{
    constexpr int NW = BLOCK / 64;
    uint32_t* s = s_cnt + parity * (NW + 1);
    const unsigned long long ballot = __ballot(flag);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t rank = (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) s[wave] = (uint32_t)__popcll(ballot);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < NW; ++w) { const uint32_t c = s[w]; s[w] = tot; tot += c; }
        s[NW] = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    return s[NW] + s[wave] + rank;
}

// ---------------------------------------------------------------- K3 ----
// shade() for one path vertex (pathTracing.cpp:3-102) in its iterative form:
// emissive early-out with the Q9 rules, vertex set-up, one NEE sample per light
// (shadow ray + weight emitted), Russian roulette, nextRay, beta update.
struct ShadeArgs {
    RayQueue qin;
    const f4* hit;
    uint32_t n;
    RayQueue qout;
    ShadowQueue sq[TRT_MAX_LIGHTS];
    uint32_t* next_count;     // survivors -> qout
    uint32_t* shadow_counts;  // [n_lights]
    f4* Lacc;
    TileDesc td;
    uint32_t s0;
    int32_t max_depth;
    DeviceStats* stats;
};

__global__ __launch_bounds__(TRT_SHADE_BLOCK) void k_shade(SceneDev sc, ShadeArgs A)
{
    __shared__ uint32_t s_cnt[2 * (TRT_SHADE_BLOCK / 64 + 1)];
    __shared__ uint32_t s_shaded, s_anyhit;
    if (threadIdx.x == 0) { s_shaded = 0; s_anyhit = 0; }
    __syncthreads();
    int parity = 0;
    uint32_t bounce_depth = 0;
    const uint32_t per_grid = gridDim.x * TRT_SHADE_BLOCK;
    // uniform trip count per block: every thread reaches every barrier
    for (uint32_t base = blockIdx.x * TRT_SHADE_BLOCK; base < A.n; base += per_grid) {
        const uint32_t i = base + threadIdx.x;
        ShadeCtx c;
        c.had_hit = c.shade_ok = c.add_L = false;
        if (i < A.n) {
            shadeBegin(sc, A.td, A.s0, A.qin.ra[i], A.qin.rb[i], A.qin.bt[i], A.hit[i], c);
            if (c.had_hit) bounce_depth = c.depth;
            if (c.add_L) {
                f4 L = A.Lacc[c.pid];
                L.x = L.x + c.addL.x; L.y = L.y + c.addL.y; L.z = L.z + c.addL.z;
                A.Lacc[c.pid] = L;
            }
        }
        const unsigned long long ok_ballot = __ballot(c.shade_ok);
        if ((threadIdx.x & 63u) == 0 && ok_ballot) atomicAdd(&s_shaded, (uint32_t)__popcll(ok_ballot));
        if (__ballot(c.had_hit) && (threadIdx.x & 63u) == 0) s_anyhit = 1;

        // direct illumination: one shadow ray per light (pathTracing.cpp:34-74)
        for (uint32_t li = 0; li < sc.n_lights; ++li) {
            bool emit = false;
            f3 wo = mk3(0, 0, 0), contrib = mk3(0, 0, 0);
            if (c.shade_ok) emit = lightSample(sc, c.vx, c.m, li, c.rng, wo, contrib);
            const uint32_t slot = blockReserve<TRT_SHADE_BLOCK>(emit, A.shadow_counts + li, s_cnt, parity);
            parity ^= 1;
            if (emit) {
                const f3 w = c.beta * contrib;
                A.sq[li].sa[slot] = mk4(c.vx.P.x, c.vx.P.y, c.vx.P.z, wo.x);  // Q6: origin = hit point, no offset
                A.sq[li].sb[slot] = mk4(wo.y, wo.z, u2f(c.pid), 0.0f);
                A.sq[li].sw[slot] = mk4(w.x, w.y, w.z, 0.0f);
            }
        }

        // indirect illumination: RR(0.8) then nextRay (pathTracing.cpp:78-99)
        f4 nra, nrb, nbt;
        const bool emit_next = shadeNext(c, A.max_depth, nra, nrb, nbt);
        const uint32_t slot = blockReserve<TRT_SHADE_BLOCK>(emit_next, A.next_count, s_cnt, parity);
        parity ^= 1;
        if (emit_next) {
            A.qout.ra[slot] = nra;
            A.qout.rb[slot] = nrb;
            A.qout.bt[slot] = nbt;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_shaded) atomicAdd(&A.stats->shaded_hits, (unsigned long long)s_shaded);
        if (s_anyhit) atomicMax(&A.stats->max_depth_hit, bounce_depth);
    }
}

// ---------------------------------------------------------------- K6 ----
// Accumulation of main.cpp:101-108: color = L / (float)SAMPLE, image += color,
// summed per pixel in sample order into the double accumulator (main.cpp:74).
__global__ __launch_bounds__(256) void k_resolve(const f4* __restrict__ Lacc, double* __restrict__ acc, uint32_t npix, uint32_t s_count, float spp)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += stride) {
        double r = acc[(size_t)p * 3 + 0], g = acc[(size_t)p * 3 + 1], b = acc[(size_t)p * 3 + 2];
        for (uint32_t s = 0; s < s_count; ++s) {
            const f4 L = Lacc[(size_t)s * npix + p];
            r += (double)(L.x / spp);
            g += (double)(L.y / spp);
            b += (double)(L.z / spp);
        }
        acc[(size_t)p * 3 + 0] = r;
        acc[(size_t)p * 3 + 1] = g;
        acc[(size_t)p * 3 + 2] = b;
    }
}

__global__ __launch_bounds__(256) void k_finalize(const double* __restrict__ acc, float* __restrict__ out, uint32_t n)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (float)acc[i];
}

// ray-batch entry (trt_trace_closest): SoA repack of host org/dir arrays
__global__ __launch_bounds__(256) void k_pack_rays(const float* __restrict__ org, const float* __restrict__ dir, f4* __restrict__ ra, f4* __restrict__ rb, uint32_t n)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        ra[i] = mk4(org[(size_t)i * 3], org[(size_t)i * 3 + 1], org[(size_t)i * 3 + 2], dir[(size_t)i * 3]);
        rb[i] = mk4(dir[(size_t)i * 3 + 1], dir[(size_t)i * 3 + 2], u2f(i), 0.0f);
    }
}

}  // namespace trtd
