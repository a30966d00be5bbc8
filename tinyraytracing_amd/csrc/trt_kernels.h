// trt_kernels.h — the wavefront kernels (gfx950, wave64).
//
// One render pass keeps N = pixels x samples-in-chunk paths in flight.  Path
// state lives in HBM as struct-of-float4-arrays (16 B per lane per load: the
// widest coalesced access, 1 KiB per wave instruction):
//   ray queue   ra = (o.x, o.y, o.z, d.x)   rb = (d.y, d.z, bits(path id), bits(meta))
//               bt = (beta.r, beta.g, beta.b, -)
//   hit buffer  (t, bits(tri), u, v)
//   shadow queue per light   sa = (o.xyz, d.x)  sb = (d.y, d.z, bits(path id), t_max: search hint / occlusion range)
//                            sw = (w.r, w.g, w.b, -)   w = beta * unoccluded contribution
//   accumulator Lacc[path id] = (L.r, L.g, L.b, -)
// Bounce 0 generates its camera rays in registers (no queue).  Per bounce: trace_closest -> shade (emits <= 1 extension ray and <= 1 shadow
// ray per light, compacted with __ballot/popcount ranks + one atomic per block)
// -> trace_shadow per light (in light order, so every path's sum has a fixed
// order and the image is bit-reproducible).
//
// Traversal stack: per-lane stack of node references in LDS, laid out
// [level][lane] so a wave's accesses hit 64 consecutive banks; levels beyond
// the kernel's LDS depth (8 or 16 levels; 16 + spill when the wide tree can need more) spill to a per-thread
// global area.  The per-lane traversals walk 4-wide nodes (trt_wide.h); tiny scenes are walked wave-uniformly.
#pragma once
#include <hip/hip_runtime.h>

#include "trt_path.h"
#include "trt_oct.h"

namespace trtd {

// Build-time tuning knobs (A/B variants are built by `make variants` and picked with TRT_HIP_LIB):
//   TRT_TRACE_MINWAVES  second __launch_bounds__ argument of the traversal kernels (0 = leave it to the compiler)
//   TRT_PREFETCH        (trt_path.h) fetch the next triangle record of a leaf while the current one is tested
#ifndef TRT_TRACE_MINWAVES
#define TRT_TRACE_MINWAVES 0
#endif
#if TRT_TRACE_MINWAVES > 0
#define TRT_TRACE_BOUNDS __launch_bounds__(256, TRT_TRACE_MINWAVES)
#else
#define TRT_TRACE_BOUNDS __launch_bounds__(256)
#endif
constexpr int TRT_TRACE_BLOCK = 256;
#ifndef TRT_LDS_STACK_MAX_LEVELS
#define TRT_LDS_STACK_MAX_LEVELS 16
#endif
constexpr int TRT_LDS_STACK_MAX = TRT_LDS_STACK_MAX_LEVELS;   // deepest LDS stack (x 256 lanes x 4 B per block); deeper levels spill to global
// k_shade comes in two flavours, chosen by the scene's light count (round 4; profiles/r04_ab_shade_tail.txt).  The kernel waits on memory at low
// occupancy (one block per CU instead of two: 1.67x its time), and a CU takes whole blocks whose waves divide evenly over its 4 SIMDs:
//   ONE light (back, soup, blob: nl known at compile time, no per-light stage pipeline): 80 VGPRs without scratch -> 512-thread blocks, THREE per CU
//     = 6 waves per SIMD (was 108 VGPRs, 2 blocks, 4 waves): k_shade -13 % on `back`, -14 % on the 10 M mesh;
//   SEVERAL lights: 96 VGPRs without scratch -> 256-thread blocks, FIVE per CU = 5 waves per SIMD, and five independently phased blocks
//     per SIMD instead of two: -14 % on veach-mis, -10 % on staircase.  (256-thread blocks with one light LOSE 60 %: twice the queue
//     reservations on ONE counter — same-address atomics serialise at ~6 ns —; with several lights they spread over the lights' counters
//     and a tile takes longer.  640-thread blocks: 10 waves do not divide over 4 SIMDs and only one block lands on a CU: +55 %.)
#ifndef TRT_SHADE1_BLOCK
#define TRT_SHADE1_BLOCK 512
#endif
#ifndef TRT_SHADE1_WAVES
#define TRT_SHADE1_WAVES 6
#endif
#ifndef TRT_SHADEN_BLOCK
#define TRT_SHADEN_BLOCK 256
#endif
#ifndef TRT_SHADEN_WAVES
#define TRT_SHADEN_WAVES 5
#endif
#ifndef TRT_SHADE_PIPE
#define TRT_SHADE_PIPE 0
#endif
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
    unsigned long long tail_rays_shadow, tail_rays_indirect;  // rays traced inside k_tail
    unsigned long long wave_inner_steps, wave_leaf_steps;     // wave-level iterations of the two traversal phases (COUNT)
    unsigned int max_depth_hit;
    unsigned int redo_rays;  // rays whose result failed the leaf-box check and went through k_trace_fix
    // COUNT builds, persistent drivers: lane census summed over all wave iterations — lanes waiting for a node step, for a leaf step,
    // finished rays waiting for the refill batch (the rest hold no ray) — and the iterations themselves (x 64 = lane slots)
    unsigned long long census_inner, census_leaf, census_done, census_iters;
};

// DEPTH levels live in LDS ([level][lane]: conflict-free, one ds_read/ds_write per access).  With
// SPILL the levels beyond DEPTH, which only a BVH deeper than DEPTH can reach, go to a per-thread
// global area; without it the host guarantees depth <= DEPTH.
template <int DEPTH, bool SPILL>
struct LdsStack {
    uint32_t* lds;    // &smem[threadIdx.x]
    uint32_t* spill;  // &spill[global thread id]
    uint32_t spill_stride;
    __device__ void push(int sp, uint32_t v)
    {
        if (!SPILL || sp < DEPTH) lds[sp * TRT_TRACE_BLOCK] = v;
        else spill[(size_t)(sp - DEPTH) * spill_stride] = v;
    }
    __device__ uint32_t pop(int sp) const
    {
        if (!SPILL) return lds[sp * TRT_TRACE_BLOCK];
        uint32_t v = lds[(sp < DEPTH ? sp : DEPTH - 1) * TRT_TRACE_BLOCK];
        asm volatile("" : "+v"(v));  // keep this a ds_read: without it the two address spaces merge into a flat load
        if (sp >= DEPTH) v = spill[(size_t)(sp - DEPTH) * spill_stride];
        return v;
    }
};

// The same for the 8-byte groups of the oct traversal (trt_oct.h): level `sp` lives at words (2 sp, 2 sp + 1) x [lane].
template <int DEPTH, bool SPILL>
struct OctLdsStack {
    uint32_t* lds;    // &smem[threadIdx.x]
    uint32_t* spill;  // &spill[global thread id]
    uint32_t spill_stride;
    __device__ void push(int sp, OctGroup g)
    {
        if (!SPILL || sp < DEPTH) { lds[(2 * sp) * TRT_TRACE_BLOCK] = g.x; lds[(2 * sp + 1) * TRT_TRACE_BLOCK] = g.y; }
        else { spill[(size_t)(2 * (sp - DEPTH)) * spill_stride] = g.x; spill[(size_t)(2 * (sp - DEPTH) + 1) * spill_stride] = g.y; }
    }
    __device__ OctGroup pop(int sp) const
    {
        OctGroup g;
        const int l = (!SPILL || sp < DEPTH) ? sp : DEPTH - 1;
        g.x = lds[(2 * l) * TRT_TRACE_BLOCK];
        g.y = lds[(2 * l + 1) * TRT_TRACE_BLOCK];
        if (SPILL) {
            asm volatile("" : "+v"(g.x), "+v"(g.y));  // keep these ds_reads (see LdsStack::pop)
            if (sp >= DEPTH) { g.x = spill[(size_t)(2 * (sp - DEPTH)) * spill_stride]; g.y = spill[(size_t)(2 * (sp - DEPTH) + 1) * spill_stride]; }
        }
        return g;
    }
};

// Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a
// contiguous range of the queue so rays that are neighbours in the queue
// (spatially coherent) meet in one L2.  Speed only, never correctness.
__device__ inline uint32_t xcdSwizzle(uint32_t bid, uint32_t nblocks)
{
    return (nblocks & 7u) ? bid : (bid & 7u) * (nblocks >> 3) + (bid >> 3);
}

// __ballot() takes an int: the predicate is widened to 0 / 1 (v_cndmask) and compared again (v_cmp_ne), two VALU instructions
// per vote that the compiler does not always fold away; the builtin takes the lane mask as it is.
__device__ __forceinline__ unsigned long long ballotb(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ inline unsigned long long waveSum(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ---------------------------------------------------------------- K1 ----
// Primary rays (main.cpp:88-95 + Camera::getRay, camera.cpp:19-28) have no kernel of their own:
// primaryRay() (trt_path.h) is a pure function of the path id, evaluated in registers by the bounce-0
// traversal kernel (PRIMARY) and again by k_shade, so no camera-ray queue is written to or read from HBM.
// path id i = s_local * npix + pixel-in-tile: a wave covers 64 neighbouring pixels.

// ------------------------------------------------------------ K2 / K4 ----
// traverseBVH (bvh.cpp:146-245) for every queued ray.  Per ray, the node/leaf
// visiting order, the culling rule and the tie rules are exactly
// traceClosest()'s (trt_path.h), whichever driver below runs it.
//
// IMPL selects the wave-level driver (trt_create picks it per scene):
//   0  wave-uniform walk of a tiny tree (<= 32 inner nodes, <= 64 triangles): scalar loads, no stack
//   3  persistent wave with a per-step scheduler: each wave owns a contiguous queue slice and refills finished lanes from
//      it (__ballot of free lanes, rank = popcount of the lower free lanes); each iteration runs the step kind (inner
//      node / one triangle) that more lanes are waiting for
// (Round 2 also carried a static driver, a while-while driver and a scheduler with a postponed leaf; they lost on every
// shipped scene — DESIGN.md §4.1 has the matrix — and are gone.)
// sc.refill_min: finished lanes are written back and refilled in batches of at least this many lanes:
// ray set-up (three IEEE reciprocals) and result write-back (the winner's barycentrics) then run at decent lane
// utilisation.  Chosen per scene in trt_create: short traversals want large batches (staircase 454 -> 434 ms/step at 48
// instead of 16), deep trees smaller ones (blob, soup: 32).

// Where a traversal kernel takes ray `i` from: the queue in HBM, or (PRIMARY) the camera-ray generator.
struct RaySource {
    const f4* ra;
    const f4* rb;
    TileDesc td;
    uint32_t s0;
};
// Queue records are written once by one kernel and read once by the next, far more of them than any cache holds: TRT_NT (bit mask) marks their accesses
// non-temporal — 1: k_shade's loads, 2: k_shade's stores, 4: the traversal kernels' ray loads, 8: their hit stores, 16: the shadow kernels' weight loads,
// 32: their read-modify-write of the radiance sums.  Measured (profiles/r04_ab_nt.txt, one box,
// two rounds): 2 shortens k_shade by 3-5 % where it has several queues to feed (veach-mis, soup, the 10 M mesh); 1 + 2 also the shadow kernel of the Cornell
// box by 3.5 % (22.9 -> 22.1 ms); 4 and 8 change nothing or take that back; 16 changes nothing; 32 costs the Cornell box's shadow kernel 11 % (the sums ARE reused: one per path and light).
// Default 3: +0.5 to +0.8 % rays per second on every workload.
#ifndef TRT_NT
#define TRT_NT 3
#endif
typedef float trt_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 ldNT(const f4* p)
{
    const trt_v4f v = __builtin_nontemporal_load(reinterpret_cast<const trt_v4f*>(p));
    return mk4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stNT(f4* p, f4 v)
{
    const trt_v4f q = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(q, reinterpret_cast<trt_v4f*>(p));
}
#define TRT_LDQ(bit, ptr) (((TRT_NT) & (bit)) ? ldNT(ptr) : *(ptr))
#define TRT_STQ(bit, ptr, val) do { if ((TRT_NT) & (bit)) stNT((ptr), (val)); else *(ptr) = (val); } while (0)

template <bool PRIMARY>
__device__ __forceinline__ void fetchRay(const SceneDev& sc, const RaySource& src, uint32_t i, f4& a, f4& b)
{
    if (PRIMARY) primaryRay(sc, src.td, src.s0, i, a, b);
    else { a = TRT_LDQ(4, src.ra + i); b = TRT_LDQ(4, src.rb + i); }
}

// Rays whose result failed the check of traceClosest() (a hit in front of the box of its own leaf; one in ~10^7): the traversal
// kernels do not store their result but append the queue index here, and k_trace_fix, launched behind every traversal launch,
// traces them again with the form that checks every triangle hit.  (Inlined into the traversal kernels that form costs 19 VGPRs,
// i.e. two waves per SIMD.)
struct RedoList {
    uint32_t* count;
    uint32_t* idx;
};

constexpr uint32_t TRT_REF_IDLE = 0xFFFFFFFFu;  // lane holds no ray        } both have the leaf bit set and
constexpr uint32_t TRT_REF_DONE = 0xFFFFFFFEu;  // ray finished, not stored } first >= 2^27 - 2: beyond TRT_MAX_TRIS, no builder emits them

struct TraceProbe {  // COUNT builds only: work and SIMD utilisation of the two step kinds
    uint32_t n_inner = 0, n_tri = 0;            // per lane: inner nodes visited, triangles tested
    uint32_t wave_inner = 0, wave_tri = 0;      // counted by the first participating lane: wave-level steps
    uint32_t c_in = 0, c_lf = 0, c_done = 0, c_it = 0;  // lane 0: the census of DeviceStats
};

// result of one ray: hit record (closest) or the NEE accumulation (shadow)
template <bool SHADOW>
__device__ __forceinline__ void storeResult(const SceneDev& sc, f3 o, f3 d, float best_t, int32_t best_tri, uint32_t best_flags, uint32_t idx, uint32_t pid,
                                            f4* __restrict__ hit, const f4* __restrict__ sw, uint32_t light_mat, f4* __restrict__ Lacc, bool any)
{
    if (!SHADOW) {
        // barycentric weights of v1, v2 (what findBaryCor feeds bvh.cpp:224): re-evaluated on the winning
        // triangle once per ray instead of carrying three more registers through the traversal
        float u = 0.f, v = 0.f;
        if (best_tri >= 0) {
            float t, un, vn, det;
            if (triTest(sc.tri_isect[best_tri], o, d, t, un, vn, det)) { u = un / det; v = vn / det; }
        }
        TRT_STQ(8, hit + idx, mk4(best_t, u2f((uint32_t)best_tri), u, v));
    } else if (any ? best_tri < 0 : (best_tri >= 0 && (best_flags >> 8) == light_mat)) {
        // pathTracing.cpp:55-58 (Q5): visible iff the CLOSEST hit carries the light's material;
        // TRT_FLAG_FIXED_NEE (`any`): visible iff nothing lies in front of the light sample
        const f4 w = TRT_LDQ(16, sw + idx);
        f4 L = TRT_LDQ(32, Lacc + pid);
        L.x = L.x + w.x; L.y = L.y + w.y; L.z = L.z + w.z;
        TRT_STQ(32, Lacc + pid, L);
    }
}

// The check of a ray's result (a hit in front of the box of its own leaf does not count: leafEntry(), trt_path.h) and the store of
// that result in ONE memory round trip: the leaf's box is requested together with what the store needs (the winner's record, or the
// shadow ray's weight and the sample's radiance), and only then is the entry distance formed.  Returns true when the result fails
// the check: nothing is stored then and the caller puts the ray on the redo list.
// OCT: the result of a traversal of the quantised 8-wide nodes (trt_oct.h) counts only if the ray also passes the reference's test of the
// exact box of the triangle's leaf.
template <bool SHADOW, bool OCT = false>
__device__ __forceinline__ bool checkedStore(const SceneDev& sc, f3 o, f3 d, f3 inv, float best_t, int32_t best_tri, uint32_t best_flags, uint32_t idx, uint32_t pid,
                                             f4* __restrict__ hit, const f4* __restrict__ sw, uint32_t light_mat, f4* __restrict__ Lacc, bool any)
{
    // a zero direction component: the literal slab test may see a NaN (trt_path.h boxTestGlm) — k_trace_fix walks the BVH2 with it.  Folded into the conditions
    // below, behind the loads: a branch in front of them cost 3 % of the traversal kernels' time.
    const bool special = raySpecial(inv);
    const uint32_t tri = best_tri >= 0 ? (uint32_t)best_tri : 0u;
    const f4 ba = sc.leaf_box[2 * (size_t)tri], bb = sc.leaf_box[2 * (size_t)tri + 1];
    if (!SHADOW) {
        const TriIsect T = sc.tri_isect[tri];
        float e;
        const bool pass = boxTest(ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, o, inv, e);
        if (special || (best_tri >= 0 && ((OCT && !pass) || best_t < trt_leaf_floor(e, sc.leaf_alpha)))) return true;
        float u = 0.f, v = 0.f;
        if (best_tri >= 0) {
            float t, un, vn, det;
            if (triTest(T, o, d, t, un, vn, det)) { u = un / det; v = vn / det; }
        }
        TRT_STQ(8, hit + idx, mk4(best_t, u2f((uint32_t)best_tri), u, v));
    } else {
        const bool vis = any ? best_tri < 0 : (best_tri >= 0 && (best_flags >> 8) == light_mat);
        f4 w = mk4(0, 0, 0, 0), L = w;
        if (vis) { w = TRT_LDQ(16, sw + idx); L = TRT_LDQ(32, Lacc + pid); }
        float e;
        const bool pass = boxTest(ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, o, inv, e);
        if (special || (best_tri >= 0 && ((OCT && !pass) || best_t < trt_leaf_floor(e, sc.leaf_alpha)))) return true;
        if (vis) {
            L.x = L.x + w.x; L.y = L.y + w.y; L.z = L.z + w.z;
            TRT_STQ(32, Lacc + pid, L);
        }
    }
    return false;
}

// IMPL 0 — wave-uniform evaluation of a tiny BVH (<= 32 inner nodes): no stack, no divergent control
// flow, no per-lane addresses.  The tree is walked in node-index order (the builders emit parents
// before children) by the whole wave at once; a per-lane bit mask records which inner nodes the lane's
// ray reaches (parent reached AND child box hit, bvh.cpp:162-166), node and triangle records are
// fetched at wave-uniform addresses (scalar loads), and a leaf is intersected when any lane reaches it.
// This is the reference's own visit set — both children, no culling by the best hit (bvh.cpp:146-175) —
// so the counters equal the oracle's, and the result is the same as the ordered traversal's because
// the leaf rule and the between-leaves rule are applied unchanged.
// Deferred divisions: a candidate triangle (acceptance test passed, t = tn / det not formed yet) is parked in a
// small per-lane LDS queue [slot][lane]; the division, the t < 0.0005 cut and the fold into the best hit run
// once per queue slot at the end of the ray (or when a lane's queue is full) instead of once per triangle that
// ANY lane of the wave is a candidate for — with 64 incoherent rays that is nearly every triangle, while one
// ray has only a few candidates.  The fold is interactBVHNode's and traverseBVH's two-level rule (bvh.cpp:219,
// 168-172) in one step per candidate: nearer wins; at equal distance a candidate of the SAME leaf as the
// current best (best index inside the candidate's leaf range; candidates arrive in index order) wins iff it is
// emissive, one of ANOTHER leaf by "leftmost emissive, else rightmost" — what the leaf-local fold followed by
// the between-leaves merge yields, since the leaves are disjoint index ranges.
constexpr int TRT_PEND_SLOTS = 4;
// The walk itself for the rays the active lanes of a wave hold (`valid`: this lane has one).  best_t comes in as the
// bound of the search (TRT_INF, or an occlusion range) and goes out with best_tri / best_flags as the closest hit.
// GLM: this wave holds a ray with a zero direction component (`special` lanes): those lanes take the literal slab test (trt_path.h boxTestGlm: a NaN from 0 * inf
// counts as the reference counts it); the walk is the reference's own visit set already, so nothing else changes for them.
template <bool COUNT, int STRIDE, bool GLM>
__device__ __forceinline__ void uniformWalkImpl(const SceneDev& sc, f3 o, f3 d, f3 inv, bool valid, bool special, f4* __restrict__ my_pend, float& best_t, int32_t& best_tri,
                                                uint32_t& best_flags, uint32_t& n_inner, uint32_t& n_tri)
{
    const uint32_t n_nodes = sc.n_nodes;
    uint32_t reach = valid ? 1u : 0u;  // bit k: the ray reaches inner node k
    uint32_t n_pend = 0;
    // division + cut + fold of every parked candidate, slot by slot (= in the order they were found)
    auto flush = [&]() {
        for (uint32_t s = 0; ballotb(s < n_pend) != 0ull; ++s) {
            if (s < n_pend) {
                const f4 e = my_pend[s * STRIDE];
                const float t = e.x / e.y;
                if (!(t < TRT_T_MIN) && !(t < e.w)) {  // bvh.cpp:189; and not in front of the box of its leaf (leafFloor(), trt_path.h)
                    const uint32_t pk = f2u(e.z);  // triangle index | first triangle of its leaf << 8 | triangles in the leaf << 16 (<= 64 triangles here)
                    const int32_t j = (int32_t)(pk & 0xFFu);
                    const uint32_t fl = f2u(sc.tri_isect[j].c.z);
                    bool take = t < best_t;
                    if (t == best_t && best_tri >= 0) {
                        const bool em = (fl & 1u) != 0, bem = (best_flags & 1u) != 0;
                        const uint32_t first = (pk >> 8) & 0xFFu, cnt = pk >> 16;
                        const bool same_leaf = (uint32_t)best_tri >= first && (uint32_t)best_tri < first + cnt;
                        take = same_leaf ? em : (em ? (!bem || j < best_tri) : (!bem && j > best_tri));
                    }
                    if (take) { best_t = t; best_tri = j; best_flags = fl; }
                }
            }
        }
        n_pend = 0;
    };
    for (uint32_t ni = 0; ni < n_nodes; ++ni) {
        const bool at = (reach >> ni) & 1u;
        if (ballotb(at) == 0ull) continue;
        const f4* np4 = reinterpret_cast<const f4*>(sc.nodes + ni);  // wave-uniform address
        const f4 q0 = np4[0], q1 = np4[1], q2 = np4[2], q3 = np4[3];
        if (COUNT && at) n_inner++;
        float e0 = 0.0f, e1 = 0.0f;
        bool h0 = at && boxTest(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, inv, e0);
        bool h1 = at && boxTest(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, inv, e1);
        if (GLM) {
            float g0, g1;
            const bool s0 = boxTestGlm(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, inv, g0), s1 = boxTestGlm(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, inv, g1);
            if (special) { h0 = at && s0; h1 = at && s1; e0 = g0; e1 = g1; }
        }
        const float f0 = trt_leaf_floor(e0, sc.leaf_alpha), f1 = trt_leaf_floor(e1, sc.leaf_alpha);  // used only where the child is a leaf
        const uint32_t child[2] = {f2u(q3.x), f2u(q3.y)};
        const bool hc[2] = {h0, h1};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const uint32_t ref = child[c];
            if (!(ref & TRT_LEAF_BIT)) {
                reach |= hc[c] ? (1u << ref) : 0u;
                continue;
            }
            const unsigned long long m_hc = ballotb(hc[c]);
            if (m_hc == 0ull) continue;
            const uint32_t first = TRT_LEAF_FIRST(ref), count = TRT_LEAF_COUNT(ref);
            for (uint32_t k = 0; k < count; ++k) {  // interactBVHNode (bvh.cpp:211-229): index order
                const TriIsect T = sc.tri_isect[first + k];  // wave-uniform address
                if (COUNT && hc[c]) n_tri++;
                float tn, un, vn, det;
                bool ok_det, ok_in;
                triCandidateParts(T, o, d, tn, un, vn, det, ok_det, ok_in);
                const bool cand = ok_det && ok_in && hc[c];
                // a full queue among the candidates: empty all of them first (votes on the single compares: their own lane masks)
                if ((ballotb(ok_det) & ballotb(ok_in) & m_hc & ballotb(n_pend == (uint32_t)TRT_PEND_SLOTS)) != 0ull) flush();
                if (cand) {
                    my_pend[n_pend * STRIDE] = mk4(tn, det, u2f((first + k) | (first << 8) | (count << 16)), c == 0 ? f0 : f1);  // + the floor of the leaf's box (leaf-box rule)
                    n_pend++;
                }
            }
        }
    }
    flush();
}
// The walk for k_tail (a few thousand paths, latency-bound; not the queue kernels, which park such rays: traceQueueUniform): one vote per ray, and the
// instantiation with the literal test only in a wave that holds a ray with a zero direction component.
template <bool COUNT, int STRIDE = TRT_TRACE_BLOCK>
__device__ __forceinline__ void uniformWalk(const SceneDev& sc, f3 o, f3 d, bool valid, f4* __restrict__ my_pend, float& best_t, int32_t& best_tri,
                                            uint32_t& best_flags, uint32_t& n_inner, uint32_t& n_tri)
{
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const bool special = valid && raySpecial(inv);
    if (ballotb(special) != 0ull) uniformWalkImpl<COUNT, STRIDE, true>(sc, o, d, inv, valid, special, my_pend, best_t, best_tri, best_flags, n_inner, n_tri);
    else uniformWalkImpl<COUNT, STRIDE, false>(sc, o, d, inv, valid, false, my_pend, best_t, best_tri, best_flags, n_inner, n_tri);
}

template <bool SHADOW, bool COUNT, bool PRIMARY>
__device__ __forceinline__ void traceQueueUniform(const SceneDev& sc, const RaySource& src, uint32_t n, f4* __restrict__ hit,
                                                  const f4* __restrict__ sw, uint32_t light_mat, f4* __restrict__ Lacc, DeviceStats* stats, bool any_flag,
                                                  f4* __restrict__ pend)
{
    const bool any = SHADOW && any_flag;  // occlusion test (TRT_FLAG_FIXED_NEE): the ray carries its own t_max in rb.w
    uint32_t n_inner = 0, n_tri = 0;
    const uint32_t lb = xcdSwizzle(blockIdx.x, gridDim.x);
    const uint32_t stride = gridDim.x * TRT_TRACE_BLOCK;
    f4* my_pend = pend + threadIdx.x;  // slot s of this lane: my_pend[s * TRT_TRACE_BLOCK]
    // Rays with a zero direction component (raySpecial, trt_path.h: the literal slab test may see 0 * inf where the clean one does not) are not stored by the
    // walk below but parked — queue index only, per wave, in LDS — and walked again with the literal test by their own wave when it runs out of rays (up to 128
    // of them; a wave that meets more goes over its share a second time).  About one ray in 10^5 on the Cornell box.  Nothing of this sits inside the walk or
    // calls it from inside the loop (a wave-uniform branch in the walk cost 4 %, a call of the second walk from the loop 50 %: the compiler's doing), and no other
    // wave waits for it (handed to the last block of the launch, that block walked alone, latency-bound: +60 %).  What remains is +4 % on k_trace_closest of the
    // Cornell box for three compares, a vote and a predicated store (profiles/r04_hardening.txt).
    __shared__ uint32_t s_parked[2 * TRT_TRACE_BLOCK];
    uint32_t* my_parked = s_parked + (threadIdx.x >> 6) * 128u;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t n_parked = 0;  // wave-uniform
    auto rewalk = [&]() {
        for (uint32_t base = 0; base < n_parked; base += 64u) {
            const bool valid = base + lane < n_parked;
            const uint32_t i = my_parked[valid ? base + lane : 0u];
            f4 a, b;
            fetchRay<PRIMARY>(sc, src, i, a, b);
            const f3 o = mk3(a.x, a.y, a.z), d = mk3(a.w, b.x, b.y);
            float best_t = any ? b.w : TRT_INF;
            int32_t best_tri = -1;
            uint32_t best_flags = 0u;
            uint32_t ni = 0, nt = 0;
            uniformWalkImpl<false, TRT_TRACE_BLOCK, true>(sc, o, d, mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z), valid, valid, my_pend, best_t, best_tri, best_flags, ni, nt);
            if (valid) storeResult<SHADOW>(sc, o, d, best_t, best_tri, best_flags, i, SHADOW ? f2u(b.z) : 0u, hit, sw, light_mat, Lacc, any);
        }
        if (lane == 0 && n_parked) atomicAdd(&stats->redo_rays, n_parked);
        n_parked = 0;
    };
    for (uint32_t base = lb * TRT_TRACE_BLOCK; base < n; base += stride) {
        const uint32_t i = base + threadIdx.x;
        const bool valid = i < n;
        const uint32_t ii = valid ? i : n - 1;
        f4 a, b;
        fetchRay<PRIMARY>(sc, src, ii, a, b);
        const f3 o = mk3(a.x, a.y, a.z), d = mk3(a.w, b.x, b.y);
        const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        float best_t = any ? b.w : TRT_INF;
        int32_t best_tri = -1;
        uint32_t best_flags = 0u;
        uniformWalkImpl<COUNT, TRT_TRACE_BLOCK, false>(sc, o, d, inv, valid, false, my_pend, best_t, best_tri, best_flags, n_inner, n_tri);
        const bool special = valid && raySpecial(inv);
        if (valid && !special) storeResult<SHADOW>(sc, o, d, best_t, best_tri, best_flags, i, SHADOW ? f2u(b.z) : 0u, hit, sw, light_mat, Lacc, any);
        const unsigned long long m_sp = ballotb(special);
        if (m_sp != 0ull) {  // (wave-uniform, rare)
            const uint32_t at = n_parked + (uint32_t)__popcll(m_sp & ((1ull << lane) - 1ull));
            if (special && at < 128u) my_parked[at] = i;
            n_parked += (uint32_t)__popcll(m_sp);  // beyond 128: counted, not kept (see below)
        }
    }
    if (n_parked != 0u && n_parked <= 128u) rewalk();
    else if (n_parked > 128u) {
        // more than the list holds — a scene that aims whole rows of rays along an axis: this wave goes over its share of the queue once more and walks the
        // batches that hold such rays with the literal test (their results were not stored above)
        for (uint32_t base = lb * TRT_TRACE_BLOCK; base < n; base += stride) {
            const uint32_t i = base + threadIdx.x;
            const bool valid = i < n;
            f4 a, b;
            fetchRay<PRIMARY>(sc, src, valid ? i : n - 1, a, b);
            const f3 o = mk3(a.x, a.y, a.z), d = mk3(a.w, b.x, b.y);
            const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
            const bool special = valid && raySpecial(inv);
            if (ballotb(special) == 0ull) continue;
            float best_t = any ? b.w : TRT_INF;
            int32_t best_tri = -1;
            uint32_t best_flags = 0u;
            uint32_t ni = 0, nt = 0;
            uniformWalkImpl<false, TRT_TRACE_BLOCK, true>(sc, o, d, inv, special, special, my_pend, best_t, best_tri, best_flags, ni, nt);
            if (special) storeResult<SHADOW>(sc, o, d, best_t, best_tri, best_flags, i, SHADOW ? f2u(b.z) : 0u, hit, sw, light_mat, Lacc, any);
        }
        if (lane == 0) atomicAdd(&stats->redo_rays, n_parked);
    }
    if (COUNT) {
        const unsigned long long si = waveSum(n_inner), st = waveSum(n_tri);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&stats->inner_visits[SHADOW ? 1 : 0], si);
            atomicAdd(&stats->tri_tests[SHADOW ? 1 : 0], st);
        }
    }
}


template <bool SHADOW, bool COUNT, int DEPTH, bool SPILL, int IMPL, bool PRIMARY, int NK>
__device__ __forceinline__ void traceQueuePersistent(const SceneDev& sc, const RaySource& src, uint32_t n, f4* __restrict__ hit,
                                           const f4* __restrict__ sw, uint32_t light_mat, f4* __restrict__ Lacc, uint32_t* __restrict__ spill,
                                           uint32_t spill_stride, DeviceStats* stats, uint32_t* smem, bool any_flag, RedoList redo)
{
    LdsStack<DEPTH, SPILL> stk;
    stk.lds = smem + threadIdx.x;
    stk.spill = spill + (size_t)blockIdx.x * TRT_TRACE_BLOCK + threadIdx.x;
    stk.spill_stride = spill_stride;
    const bool any = SHADOW && any_flag;  // occlusion test (TRT_FLAG_FIXED_NEE): the ray carries its own t_max in rb.w
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lower = (1ull << lane) - 1ull;
    // contiguous queue slice of this wave (XCD-aware: neighbouring slices share an L2)
    const uint32_t n_waves = gridDim.x * (TRT_TRACE_BLOCK / 64);
    const uint32_t wave = xcdSwizzle(blockIdx.x, gridDim.x) * (TRT_TRACE_BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t per = (n + n_waves - 1) / n_waves;
    const unsigned long long w0 = (unsigned long long)wave * per;
    uint32_t next = (uint32_t)__builtin_amdgcn_readfirstlane((int)(w0 < n ? w0 : n));
    const uint32_t end = (uint32_t)__builtin_amdgcn_readfirstlane((int)((w0 + per) < n ? (w0 + per) : n));

    uint32_t cur = TRT_REF_IDLE, idx = 0, pid = 0;
    int sp = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 0), inv = mk3(0, 0, 0);
    float best_t = TRT_INF;
    int32_t best_tri = -1;
    uint32_t best_flags = 0;
    // IMPL 3: fold state of the leaf the lane is in (interactBVHNode's local `res`, bvh.cpp:213)
    uint32_t lk = 0;  // next triangle of the leaf, relative to its first
    float lt = TRT_INF;
    int32_t li = -1;
    uint32_t lflags = 0;
    TraceProbe pr;

    for (;;) {
        // ---- write back finished rays and refill free lanes from the wave's slice, in batches
        const bool working = cur < TRT_REF_DONE;
        const unsigned long long m_work = ballotb(working);
        const unsigned long long m_done = ballotb(cur == TRT_REF_DONE);
        const unsigned long long m_free = ~m_work;  // finished or empty lanes
        const bool can_fill = next < end;
        if (m_work == 0ull || (m_done != 0ull && (uint32_t)__popcll(can_fill ? m_free : m_done) >= sc.refill_min)) {
            if (cur == TRT_REF_DONE) {
                // a hit in front of the box of its own leaf does not count (leafEntry(), trt_path.h): the result is checked once per ray,
                // and the (one in ~10^7) rays that end on such a hit go to k_trace_fix instead of being stored
                if (checkedStore<SHADOW>(sc, o, d, inv, best_t, best_tri, best_flags, idx, pid, hit, sw, light_mat, Lacc, any)) redo.idx[atomicAdd(redo.count, 1u)] = idx;
                cur = TRT_REF_IDLE;
            }
            if (can_fill) {
                const uint32_t rank = (uint32_t)__popcll(m_free & lower);
                if (!working && next + rank < end) {
                    idx = next + rank;
                    f4 a, b;
                    fetchRay<PRIMARY>(sc, src, idx, a, b);
                    o = mk3(a.x, a.y, a.z);
                    d = mk3(a.w, b.x, b.y);
                    if (SHADOW) pid = f2u(b.z);
                    inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                    best_t = SHADOW ? b.w : TRT_INF; best_tri = -1; best_flags = 0u;  // shadow rays carry a bound: t_max (any) or a search hint
                    sp = 0;
                    cur = 0u;  // nodes[0] is always an inner node
                    lk = 0; lt = TRT_INF; li = -1;
                }
                const uint32_t taken = (uint32_t)__popcll(m_free);
                next = (end - next) < taken ? end : next + taken;
            }
            if (ballotb(cur < TRT_REF_DONE) == 0ull) break;  // nothing left in the slice
        }

        {
        const bool is_inner = !(cur & TRT_LEAF_BIT);
        const bool is_leaf = cur < TRT_REF_DONE && (cur & TRT_LEAF_BIT);
        const unsigned long long m_in = ballotb(is_inner), m_lf = ballotb(is_leaf);
        bool adv = false;  // this lane is done with its node: take the next one off the stack, or finish the ray
        if (sc.sched_in_w * (uint32_t)__popcll(m_in) >= sc.sched_lf_w * (uint32_t)__popcll(m_lf)) {
            // ---- inner-node step
            if (is_inner) {
                if (COUNT) { pr.n_inner++; if (lane == (uint32_t)__ffsll((long long)m_in) - 1u) pr.wave_inner++; }
                if (innerStep(sc, cur, sp, stk, o, inv, trt_cull_bound(best_t, sc.cull_alpha))) { lk = 0; lt = TRT_INF; li = -1; }
                else adv = true;
            }
        } else {
            // ---- leaf step: one triangle of interactBVHNode's loop (bvh.cpp:211-229), index order
            if (is_leaf) {
                const uint32_t first = TRT_LEAF_FIRST(cur), count = TRT_LEAF_COUNT(cur);
                if (lk < count) {
                    const uint32_t i = first + lk;
                    const TriIsect T = sc.tri_isect[i];
                    if (COUNT) { pr.n_tri++; if (lane == (uint32_t)__ffsll((long long)m_lf) - 1u) pr.wave_tri++; }
                    float t, un, vn, det;
                    if (triTest(T, o, d, t, un, vn, det)) {
                        const uint32_t fl = f2u(T.c.z);
                        if ((t == lt && (fl & 1u)) || t < lt) { lt = t; li = (int32_t)i; lflags = fl; }
                    }
                    lk++;
                }
                if (lk >= count) {  // leaf done: merge its winner, then the next node
                    if (li >= 0) {
                        bool take = lt < best_t;
                        if (lt == best_t && best_tri >= 0) {  // equal distance across leaves (bvh.cpp:168-172, order independent form)
                            const bool lem = (lflags & 1u) != 0, bem = (best_flags & 1u) != 0;
                            take = lem ? (!bem || li < best_tri) : (!bem && li > best_tri);
                        }
                        if (take) { best_t = lt; best_tri = li; best_flags = lflags; }
                    }
                    adv = true;
                }
            }
        }
        if (adv) {
            if (any && best_tri >= 0) cur = TRT_REF_DONE;
            else if (sp != 0) cur = stk.pop(--sp);
            else if (SHADOW && !any && best_tri < 0 && best_t < TRT_INF) { best_t = TRT_INF; cur = 0u; }  // nothing in front of the hint: search again without it
            else cur = TRT_REF_DONE;
            lk = 0; lt = TRT_INF; li = -1;
        }
        }
    }
    if (COUNT) {
        const unsigned long long si = waveSum(pr.n_inner), st = waveSum(pr.n_tri), wi = waveSum(pr.wave_inner), wt = waveSum(pr.wave_tri);
        if (lane == 0) {
            atomicAdd(&stats->inner_visits[SHADOW ? 1 : 0], si);
            atomicAdd(&stats->tri_tests[SHADOW ? 1 : 0], st);
            atomicAdd(&stats->wave_inner_steps, wi);
            atomicAdd(&stats->wave_leaf_steps, wt);
        }
    }
}

// The same persistent-wave scheduler over the 8-wide compressed nodes (NK = 1, trt_oct.h).  Lane state: the node group it is
// descending (`ng`: first-child index, hit byte, imask), the triangles it still has to test (`tg`: first record, one bit each), an
// 8-byte stack entry per level.  A lane with triangle bits waits for a leaf step, one without them for a node step; a wave runs the
// kind more of its lanes wait for (same weights).  idx == ~0: the lane holds no ray; no work bits and idx != ~0: ray finished, not
// stored yet.  Results are checked when they are stored (checkedStore<.., OCT>): the ray must pass the exact box of its hit's leaf.
// (Round 3 also tried rays fetched from the launch's queue through atomic counters, in chunks, with finished rays parked in LDS and
// stored 64 at a time: lanes per step 0.55 / 0.45 -> 0.64 / 0.57, and 20-30 % SLOWER on every scene — 28 KiB of LDS and 84 VGPRs
// leave five waves per SIMD instead of seven, and the parked results cost a second fetch of their rays: more instructions per ray
// than the busier lanes save.  And, on static slices, NO refill batches at all — a free lane takes the next ray at the beginning of the
// next node step, finished rays parked as above —: lanes per step 0.57 / 0.88 -> 0.69 / 1.12 (two triangles per leaf step) and 14-40 %
// slower: the ray fetch and its three reciprocals then run in almost every node step for a handful of lanes.  These kernels are bound
// by VALU ISSUE (profiles/r03_roofs_stair_oct.txt): what counts is wave-level instructions per ray, and a refill batch of 48 amortises
// 200 of them over 48 rays.  profiles/r03_ab_oct.txt has all of it.  Static slices and batches stay.)
// sc.leaf_loop (TRT_OCT_LEAF_LOOP as a scene parameter since round 4): triangles a lane tests per leaf step.  Leaves hold two: with 2 the
// second one follows at once, without another round of votes, and the lane is back at a node with its neighbours (lanes per leaf step
// 0.45 -> 0.88; staircase, veach-mis +4 %, the meshes +2-3 %; 3: no better).  A caller's tree with larger leaves (the reference's: 8)
// brings a group of up to 24 triangles per node and gets a longer loop (trt_create; TRT_LEAF_LOOP in the environment overrides).
// (Also measured: the root and its children read from an LDS copy — a fifth of all node fetches on veach-mis —: +-0.3 %.  These kernels
// are bound by VALU issue, profiles/r03_roofs_stair.txt, not by the texture addresser.  Removed.)
template <bool SHADOW, bool COUNT, int DEPTH, bool SPILL, bool PRIMARY>
__device__ __forceinline__ void traceQueuePersistentOct(const SceneDev& sc, const RaySource& src, uint32_t n, f4* __restrict__ hit,
                                           const f4* __restrict__ sw, uint32_t light_mat, f4* __restrict__ Lacc, uint32_t* __restrict__ spill,
                                           uint32_t spill_stride, DeviceStats* stats, uint32_t* smem, bool any_flag, RedoList redo, const LightBox& lbox)
{
    OctLdsStack<DEPTH, SPILL> stk;
    stk.lds = smem + threadIdx.x;
    stk.spill = spill + (size_t)blockIdx.x * TRT_TRACE_BLOCK + threadIdx.x;
    stk.spill_stride = spill_stride;
    const bool any = SHADOW && any_flag;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lower = (1ull << lane) - 1ull;
    const uint32_t n_waves = gridDim.x * (TRT_TRACE_BLOCK / 64);
    const uint32_t wave = xcdSwizzle(blockIdx.x, gridDim.x) * (TRT_TRACE_BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t per = (n + n_waves - 1) / n_waves;
    const unsigned long long w0 = (unsigned long long)wave * per;
    uint32_t next = (uint32_t)__builtin_amdgcn_readfirstlane((int)(w0 < n ? w0 : n));
    const uint32_t end = (uint32_t)__builtin_amdgcn_readfirstlane((int)((w0 + per) < n ? (w0 + per) : n));

    constexpr uint32_t NO_RAY = 0xFFFFFFFFu;
    uint32_t idx = NO_RAY;
    float stop_t = -TRT_INF;  // parity-mode shadow rays: a hit nearer than this ends the search (LightBox)
    int sp = 0;
    OctGroup ng, tg;
    ng.x = 0u; ng.y = 0u; tg.x = 0u; tg.y = 0u;
    f3 d = mk3(0, 0, 0);
    OctRay R;
    R.o = mk3(0, 0, 0); R.inv = mk3(0, 0, 0); R.octinv4 = 0u;
    float best_t = TRT_INF;
    int32_t best_tri = -1;
    uint32_t best_flags = 0;
    TraceProbe pr;

    for (;;) {
        const bool working = (tg.y | (ng.y & 0xFF000000u)) != 0u;
        const bool done = !working && idx != NO_RAY;
        const unsigned long long m_work = ballotb(working);
        const unsigned long long m_done = ballotb(done);
        const unsigned long long m_free = ~m_work;
        const bool can_fill = next < end;
        if (m_work == 0ull || (m_done != 0ull && (uint32_t)__popcll(can_fill ? m_free : m_done) >= sc.refill_min)) {
            if (done) {
                const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);  // the exact reciprocals (R.inv stands in 2^40 for an infinite one)
                const uint32_t pid = SHADOW ? f2u(src.rb[idx].z) : 0u;    // read again here instead of carried through the traversal
                if (checkedStore<SHADOW, true>(sc, R.o, d, inv, best_t, best_tri, best_flags, idx, pid, hit, sw, light_mat, Lacc, any)) redo.idx[atomicAdd(redo.count, 1u)] = idx;
                idx = NO_RAY;
            }
            if (can_fill) {
                const uint32_t rank = (uint32_t)__popcll(m_free & lower);
                if (!working && next + rank < end) {
                    idx = next + rank;
                    f4 a, b;
                    fetchRay<PRIMARY>(sc, src, idx, a, b);
                    d = mk3(a.w, b.x, b.y);
                    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                    R = makeOctRay(mk3(a.x, a.y, a.z), d, inv);
                    best_t = SHADOW ? b.w : TRT_INF; best_tri = -1; best_flags = 0u;
                    sp = 0;
                    ng.x = 0u; ng.y = 0x80000000u;  // the root
                    tg.y = 0u;
                    if (SHADOW && !any) {
                        // where the light's triangles can begin (LightBox): a ray that misses the box altogether cannot end on the light —
                        // no hit there counts — and needs no search at all
                        float e;
                        const bool pass = boxTest(lbox.lo[0], lbox.lo[1], lbox.lo[2], lbox.hi[0], lbox.hi[1], lbox.hi[2], R.o, inv, e);
                        stop_t = trt_leaf_floor(e, sc.leaf_alpha);
                        if (!pass) ng.y = 0u;
                    }
                }
                const uint32_t taken = (uint32_t)__popcll(m_free);
                next = (end - next) < taken ? end : next + taken;
            }
            if (ballotb((tg.y | (ng.y & 0xFF000000u)) != 0u) == 0ull) break;  // nothing left in the slice
        }

        const bool is_leaf = tg.y != 0u;
        const bool is_inner = !is_leaf && (ng.y & 0xFF000000u) != 0u;
        const unsigned long long m_in = ballotb(is_inner), m_lf = ballotb(is_leaf);
        if (COUNT) {
            const unsigned long long m_dn = ballotb(!is_leaf && !is_inner && idx != NO_RAY);
            if (lane == 0) { pr.c_in += (uint32_t)__popcll(m_in); pr.c_lf += (uint32_t)__popcll(m_lf); pr.c_done += (uint32_t)__popcll(m_dn); pr.c_it++; }
        }
        bool adv = false;  // the lane has used up its groups: next group off the stack, or the ray is finished
        if (sc.sched_in_w * (uint32_t)__popcll(m_in) >= sc.sched_lf_w * (uint32_t)__popcll(m_lf)) {
            if (is_inner) {
                if (COUNT) { pr.n_inner++; if (lane == (uint32_t)__ffsll((long long)m_in) - 1u) pr.wave_inner++; }
                const uint32_t ni = octNextChild(ng, R);
                if (ng.y & 0xFF000000u) stk.push(sp++, ng);
                octVisit(sc.onodes, ni, R, trt_cull_bound(best_t, sc.leaf_alpha), ng, tg);
                adv = (tg.y | (ng.y & 0xFF000000u)) == 0u;
            }
        } else {
            if (is_leaf) {
                // up to sc.leaf_loop triangles of the lane's group per step
#pragma unroll 1
                for (uint32_t rep = 0; (rep == 0u || rep < sc.leaf_loop) && tg.y != 0u; ++rep) {  // at least one: a step always makes progress
                    const uint32_t b = (uint32_t)__ffs((int)tg.y) - 1u;
                    tg.y &= tg.y - 1u;
                    const TriIsect T = sc.tri_trav[tg.x + b];
                    if (COUNT) { pr.n_tri++; if (rep == 0 && lane == (uint32_t)__ffsll((long long)m_lf) - 1u) pr.wave_tri++; }
                    float t, un, vn, det;
                    if (triTest(T, R.o, d, t, un, vn, det)) octFold(t, f2u(T.c.w), f2u(T.c.z), best_t, best_tri, best_flags);
                }
                if (SHADOW && !any && best_tri >= 0 && best_t < stop_t) { tg.y = 0u; ng.y = 0u; sp = 0; }  // occluded for certain (LightBox)
                if (tg.y == 0u) {
                    if (any && best_tri >= 0) { ng.y = 0u; sp = 0; }  // occlusion test: the first leaf that yields a hit ends the ray
                    adv = (ng.y & 0xFF000000u) == 0u;
                }
            }
        }
        if (adv) {
            if (sp != 0 && !(any && best_tri >= 0)) ng = stk.pop(--sp);
            else if (SHADOW && !any && best_tri < 0 && best_t < TRT_INF) { best_t = TRT_INF; ng.x = 0u; ng.y = 0x80000000u; }  // nothing in front of the hint: search again without it
            else ng.y = 0u;  // finished
        }
    }
    if (COUNT) {
        const unsigned long long si = waveSum(pr.n_inner), st = waveSum(pr.n_tri), wi = waveSum(pr.wave_inner), wt = waveSum(pr.wave_tri);
        if (lane == 0) {
            atomicAdd(&stats->inner_visits[SHADOW ? 1 : 0], si);
            atomicAdd(&stats->tri_tests[SHADOW ? 1 : 0], st);
            atomicAdd(&stats->wave_inner_steps, wi);
            atomicAdd(&stats->wave_leaf_steps, wt);
            atomicAdd(&stats->census_inner, (unsigned long long)pr.c_in);
            atomicAdd(&stats->census_leaf, (unsigned long long)pr.c_lf);
            atomicAdd(&stats->census_done, (unsigned long long)pr.c_done);
            atomicAdd(&stats->census_iters, (unsigned long long)pr.c_it);
        }
    }
}

template <bool SHADOW, bool COUNT, int DEPTH, bool SPILL, int IMPL, bool PRIMARY, int NK>
__device__ __forceinline__ void traceQueue(const SceneDev& sc, const RaySource& src, uint32_t n, f4* __restrict__ hit,
                                           const f4* __restrict__ sw, uint32_t light_mat, f4* __restrict__ Lacc, uint32_t* __restrict__ spill,
                                           uint32_t spill_stride, DeviceStats* stats, uint32_t* smem, bool any_flag, RedoList redo, const LightBox& lbox)
{
    if constexpr (IMPL == 0) traceQueueUniform<SHADOW, COUNT, PRIMARY>(sc, src, n, hit, sw, light_mat, Lacc, stats, any_flag, reinterpret_cast<f4*>(smem));
    else if constexpr (NK == 1) traceQueuePersistentOct<SHADOW, COUNT, DEPTH, SPILL, PRIMARY>(sc, src, n, hit, sw, light_mat, Lacc, spill, spill_stride, stats, smem, any_flag, redo, lbox);
    else traceQueuePersistent<SHADOW, COUNT, DEPTH, SPILL, IMPL, PRIMARY, NK>(sc, src, n, hit, sw, light_mat, Lacc, spill, spill_stride, stats, smem, any_flag, redo);
}

// PRIMARY: bounce 0 — ray i is the camera ray of path i, generated in registers (K1 of SURVEY.md §7 fused
// into K2: no primary-ray queue is ever written or read).
template <bool COUNT, int DEPTH, bool SPILL, int IMPL, bool PRIMARY, int NK>
__global__ TRT_TRACE_BOUNDS void k_trace_closest(SceneDev sc, RaySource src, f4* __restrict__ hit, uint32_t n,
                                                 uint32_t* __restrict__ spill, uint32_t spill_stride, DeviceStats* stats, RedoList redo)
{
    __shared__ __attribute__((aligned(16))) uint32_t smem[IMPL == 0 ? TRT_PEND_SLOTS * 4 * TRT_TRACE_BLOCK : (NK == 1 ? 2 : 1) * DEPTH * TRT_TRACE_BLOCK];  // stack (8-byte entries on the oct tree), or (uniform walk) the candidate queue
    const LightBox nobox = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    traceQueue<false, COUNT, DEPTH, SPILL, IMPL, PRIMARY, NK>(sc, src, n, hit, nullptr, 0u, nullptr, spill, spill_stride, stats, smem, false, redo, nobox);
}

// Shadow test of shade() (pathTracing.cpp:51-58): CLOSEST hit, visible iff
// its material is the light's (Q5); then L += w.  One launch per light, in
// light order; each path has at most one ray per launch, so the read-modify-
// write of Lacc needs no atomic and the sum order is fixed.
template <bool COUNT, int DEPTH, bool SPILL, int IMPL, int NK>
__global__ TRT_TRACE_BOUNDS void k_trace_shadow(SceneDev sc, ShadowQueue sq, uint32_t n, uint32_t light_mat, f4* __restrict__ Lacc,
                                                uint32_t* __restrict__ spill, uint32_t spill_stride, DeviceStats* stats, uint32_t any, RedoList redo, LightBox lbox)
{
    __shared__ __attribute__((aligned(16))) uint32_t smem[IMPL == 0 ? TRT_PEND_SLOTS * 4 * TRT_TRACE_BLOCK : (NK == 1 ? 2 : 1) * DEPTH * TRT_TRACE_BLOCK];
    RaySource src;
    src.ra = sq.sa;
    src.rb = sq.sb;
    src.s0 = 0;
    traceQueue<true, COUNT, DEPTH, SPILL, IMPL, false, NK>(sc, src, n, nullptr, sq.sw, light_mat, Lacc, spill, spill_stride, stats, smem, any != 0u, redo, lbox);
}

// The exact form of the traversal for the rays a traversal launch put on its redo list (see RedoList): a few blocks, launched
// behind every launch of k_trace_closest / k_trace_shadow of a per-lane driver; it finds an empty list all but once in ~10^7 rays
// on padded trees.  redo.count[0] = length of the list, redo.count[1] = blocks of this launch that are through: the last one adds
// the length to DeviceStats::redo_rays (trt_stats.redo_rays: how often the slow path ran is visible to the caller) and empties the list.
constexpr uint32_t TRT_FIX_BLOCKS = 32;
template <bool SHADOW, bool PRIMARY, int NK>
__global__ __launch_bounds__(TRT_TRACE_BLOCK) void k_trace_fix(SceneDev sc, RaySource src, f4* __restrict__ hit, const f4* __restrict__ sw, uint32_t light_mat,
                                                               f4* __restrict__ Lacc, uint32_t* __restrict__ spill, uint32_t spill_stride, RedoList redo, uint32_t any_flag,
                                                               DeviceStats* stats)
{
    __shared__ __attribute__((aligned(16))) uint32_t smem[TRT_LDS_STACK_MAX * TRT_TRACE_BLOCK];
    const uint32_t n = *redo.count;  // complete: the traversal launch precedes this one on the stream; stable until the last block resets it
    if (n == 0u) return;             // every block sees the same n: all leave here, or none
    {
        LdsStack<TRT_LDS_STACK_MAX, true> stk;
        stk.lds = smem + threadIdx.x;
        stk.spill = spill + (size_t)blockIdx.x * TRT_TRACE_BLOCK + threadIdx.x;
        stk.spill_stride = spill_stride;
        const bool any = SHADOW && any_flag != 0u;
        for (uint32_t k = blockIdx.x * TRT_TRACE_BLOCK + threadIdx.x; k < n; k += gridDim.x * TRT_TRACE_BLOCK) {
            const uint32_t i = redo.idx[k];
            f4 a, b;
            fetchRay<PRIMARY>(sc, src, i, a, b);
            const f3 o = mk3(a.x, a.y, a.z), d = mk3(a.w, b.x, b.y);
            uint32_t ni = 0, nt = 0;
            const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
            const Hit h = (raySpecial(inv) && rayOnABoxPlane(sc, o, inv))
                              ? traceClosestBvh2Glm<LdsStack<TRT_LDS_STACK_MAX, true>, false>(sc, o, d, stk, ni, nt, SHADOW ? b.w : TRT_INF, any)
                              : traceClosestPass<LdsStack<TRT_LDS_STACK_MAX, true>, false, NK, true>(sc, o, d, stk, ni, nt, SHADOW ? b.w : TRT_INF, any, SHADOW && !any);
            storeResult<SHADOW>(sc, o, d, h.t, h.tri, h.flags, i, SHADOW ? f2u(b.z) : 0u, hit, sw, light_mat, Lacc, any);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(redo.count + 1, 1u) + 1u == gridDim.x) {  // every block has read n and finished its share
            atomicAdd(&stats->redo_rays, n);
            redo.count[1] = 0u;
            redo.count[0] = 0u;  // ready for the next launch
        }
    }
}


// Block barrier for data exchanged through LDS only.  __syncthreads() also fences global memory: the compiler puts
// s_waitcnt vmcnt(0) in front of it, i.e. every wave waits for the acknowledgement of the ray records it has just stored
// (a trip to the L2 and, with the queues streaming at terabytes per second, well beyond) although no thread of the block ever
// reads them.  The fences here name the LDS address space, so only lgkmcnt(0) is waited for and stores stay in flight.
#ifndef TRT_LDS_BARRIER
#define TRT_LDS_BARRIER 1
#endif
__device__ __forceinline__ void ldsBarrier()
{
#if TRT_LDS_BARRIER
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#else
    __syncthreads();
#endif
}

// Block-wide stream compaction in two halves.  blockStage: every thread calls it with its `flag`; wave
// ranks come from __ballot/popcount, wave totals go through LDS, and after ONE barrier thread 0 turns them
// into exclusive offsets and issues the block's single atomicAdd on the queue counter — without waiting for
// it.  The value it returns (`pend_base`, thread 0) is published in the NEXT call, before that call's
// barrier, into the previous stage's buffer `pend_s`; after that barrier every thread reads its slot of the
// previous stage as pend_s[NW] + pend_s[wave] + rank.  s_cnt holds three buffers of NW + 1 words, used in
// turn (`parity` 0,1,2): a buffer is rewritten two barriers after its last reader.
template <int BLOCK>
__device__ inline uint32_t* blockStage(bool flag, uint32_t* counter, uint32_t* s_cnt, int parity, uint32_t& rank, const uint32_t* pend_s, uint32_t& pend_base)
{
    constexpr int NW = BLOCK / 64;
    uint32_t* s = s_cnt + parity * (NW + 1);
    const unsigned long long ballot = ballotb(flag);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    rank = (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) s[wave] = (uint32_t)__popcll(ballot);
    if (threadIdx.x == 0 && pend_s) const_cast<uint32_t*>(pend_s)[NW] = pend_base;
    ldsBarrier();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < NW; ++w) { const uint32_t c = s[w]; s[w] = tot; tot += c; }
        pend_base = tot ? atomicAdd(counter, tot) : 0u;
    }
    return s;
}

// Two queues reserved by ONE 64-bit atomic (low word: queue a, high word: queue b): the counters of the last light's
// shadow queue and of the next bounce's queue share an 8-byte word, so a tile costs one atomic and one barrier less —
// the atomic unit takes the reservations of one address one after the other, and k_shade's blocks reach them in bursts.
// s2: NW offsets of a, base of a, NW offsets of b, base of b.  Same protocol as blockStage: the previous stage's base is
// published before the barrier, this stage's atomic is issued behind it and its result (`base`, thread 0) is consumed later.
template <int BLOCK>
__device__ inline void blockStage2(bool fa, bool fb, unsigned long long* counter, uint32_t* s2, uint32_t& rank_a, uint32_t& rank_b,
                                   const uint32_t* pend_s, uint32_t pend_base, unsigned long long& base)
{
    constexpr int NW = BLOCK / 64;
    const unsigned long long ba = ballotb(fa), bb = ballotb(fb);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    rank_a = (uint32_t)__popcll(ba & below);
    rank_b = (uint32_t)__popcll(bb & below);
    if (lane == 0) { s2[wave] = (uint32_t)__popcll(ba); s2[NW + 1 + wave] = (uint32_t)__popcll(bb); }
    if (threadIdx.x == 0 && pend_s) const_cast<uint32_t*>(pend_s)[NW] = pend_base;
    ldsBarrier();
    if (threadIdx.x == 0) {
        uint32_t ta = 0, tb = 0;
        for (int w = 0; w < NW; ++w) {
            const uint32_t ca = s2[w], cb = s2[NW + 1 + w];
            s2[w] = ta; ta += ca;
            s2[NW + 1 + w] = tb; tb += cb;
        }
        base = (ta | tb) ? atomicAdd(counter, ((unsigned long long)tb << 32) | ta) : 0ull;
    }
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
    unsigned long long* pair_count;  // low word: survivors -> qout, high word: shadow rays of the LAST light (one reservation for both)
    uint32_t* shadow_counts;  // light l < n_lights - 1: shadow_counts[l * shadow_count_stride]
    uint32_t shadow_count_stride;
    f4* Lacc;
    TileDesc td;
    uint32_t s0;
    int32_t max_depth;
    uint32_t primary;         // bounce 0: entry i is the camera ray of path i (not in HBM); Lacc is initialised here
    // Byte counts of the small scene tables that k_shade<TABS> stages in LDS (read only for the tables of TABS):
    // material / light / CDF / light-triangle look-ups then cost an LDS access instead of a dependent
    // global round trip each.
    uint32_t lds_mat_bytes, lds_light_bytes, lds_cum_bytes, lds_ltri_bytes, lds_tshade_bytes;
    const f4* lds_image;      // the tables of TABS, each padded to 16 B, packed in that order
    uint32_t lds_image_words;
    uint32_t rows_lds;        // number of rows of td.rows to keep in LDS as 16-bit values (0: the tile has more than shadeRowsLds(block) rows, or rows >= 65536)
    DeviceStats* stats;
};

constexpr uint32_t TRT_SHADE_LDS_TABLE_BYTES = 24 * 1024;
// rows of the tile's row table kept in LDS as 16-bit values: 8192 in 512-thread blocks (3 x 41 KB per CU), 2304 — a 4K image has 2160 — in
// 256-thread blocks (5 x 29 KB per CU); taller tiles read the table from global memory
constexpr uint32_t shadeRowsLds(int block) { return block >= 512 ? 8192u : 2304u; }

// The tile's row table as k_shade sees it: a 16-bit copy in LDS when it fits.  The key of a path's random stream hangs on the
// image row, i.e. on this look-up: from global memory it is a dependent load in every vertex, and — the memory counter being
// in order — one that also waits for the queue records requested ahead for the next tile.
struct RowsShade {
    const uint16_t* lds;
    bool in_lds;
    __device__ int operator()(const TileDesc& td, uint32_t r) const
    {
        int y = 0;
        if (in_lds) {
            y = lds[r];
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(y));  // keeps this a ds_read (merged with the global load below it would become a FLAT one)
#endif
        } else {
            y = td.rows[r];
        }
        return y;
    }
};

// TABS: bit k set = table k (materials, lights, light CDF, light triangles, shading triangles) is staged in LDS by this
// instantiation.  A compile-time choice so that every table access is a plain LDS (ds_read) or global load: behind a
// run-time choice the pointers are generic, the accesses FLAT, and each of them waits for vmcnt(0) — i.e. for the ray
// stores issued before it — as well as for the LDS.
// (Round 3's probe `short` — weight and throughput records stored as 8 bytes, wrong images — bought k_shade 3 %: profiles/r03_ab_oct.txt (8).)
// ONE_LIGHT: the scene has exactly one light (the loop over the lights' stages is empty at compile time).  BLOCK threads, WAVES per SIMD asked of the compiler.
template <uint32_t TABS, bool ONE_LIGHT, int BLOCK = (ONE_LIGHT ? TRT_SHADE1_BLOCK : TRT_SHADEN_BLOCK), int WAVES = (ONE_LIGHT ? TRT_SHADE1_WAVES : TRT_SHADEN_WAVES)>
__global__ __launch_bounds__(BLOCK, WAVES) void k_shade(SceneDev sc, ShadeArgs A)
{
    constexpr int TRT_SHADE_BLOCK = BLOCK;
    constexpr uint32_t TRT_SHADE_ROWS_LDS = shadeRowsLds(BLOCK);
    __shared__ uint32_t s_cnt[3 * (TRT_SHADE_BLOCK / 64 + 1)];
    __shared__ uint32_t s_cnt2[2 * (2 * (TRT_SHADE_BLOCK / 64) + 2)];  // blockStage2, two buffers used in turn
    __shared__ uint32_t s_shaded, s_anyhit;
    __shared__ __attribute__((aligned(16))) uint32_t s_tab[TABS ? TRT_SHADE_LDS_TABLE_BYTES / 4 : 4];
    __shared__ uint16_t s_rows[TRT_SHADE_ROWS_LDS];
    if (threadIdx.x == 0) { s_shaded = 0; s_anyhit = 0; }
    for (uint32_t r = threadIdx.x; r < A.rows_lds; r += TRT_SHADE_BLOCK) s_rows[r] = (uint16_t)A.td.rows[r];
    RowsShade rows;
    rows.lds = s_rows;
    rows.in_lds = A.rows_lds != 0u;
    bool rows_visible = A.rows_lds == 0u;  // the copy above is visible to the block (a barrier lies in between)
    {   // this block's view of the staged tables: the copy itself happens in the first tile, next to that tile's own loads
        uint32_t off = 0;
        auto at = [&](uint32_t bytes) -> const void* { const void* q = s_tab + off / 4; off += (bytes + 15u) & ~15u; return q; };
        if (TABS & 1u) sc.materials = static_cast<const MaterialDev*>(at(A.lds_mat_bytes));
        if (TABS & 2u) sc.lights = static_cast<const LightDev*>(at(A.lds_light_bytes));
        if (TABS & 4u) sc.light_cum = static_cast<const float*>(at(A.lds_cum_bytes));
        if (TABS & 8u) sc.light_tris = static_cast<const LightTriDev*>(at(A.lds_ltri_bytes));
        if (TABS & 16u) sc.tri_shade = static_cast<const TriShade*>(at(A.lds_tshade_bytes));
    }
    bool staged = TABS == 0u;
    int parity = 0, parity2 = 0;
    uint32_t bounce_depth = 0;
    const uint32_t per_grid = gridDim.x * TRT_SHADE_BLOCK;
    // uniform trip count per block: every thread reaches every barrier
    // Everything a vertex reads from the queues is requested at once (one memory round trip instead of hit -> ray -> ...
    // in a chain) and one tile AHEAD (TRT_SHADE_PIPE): the records of the block's next tile are in flight while this tile
    // is computed, so a tile starts on data that has already arrived.
    auto loadTile = [&](uint32_t i, f4& hit4, f4& ra, f4& rb, f4& bt) {
        hit4 = mk4(TRT_INF, u2f(0xFFFFFFFFu), 0.0f, 0.0f); ra = mk4(0, 0, 0, 0); rb = ra; bt = mk4(1.0f, 1.0f, 1.0f, 0.0f);
        if (i < A.n) {
            hit4 = TRT_LDQ(1, A.hit + i);
            if (!A.primary) { ra = TRT_LDQ(1, A.qin.ra + i); rb = TRT_LDQ(1, A.qin.rb + i); bt = TRT_LDQ(1, A.qin.bt + i); }
        }
    };
#if TRT_SHADE_PIPE
    f4 n_hit, n_ra, n_rb, n_bt;
    loadTile(blockIdx.x * TRT_SHADE_BLOCK + threadIdx.x, n_hit, n_ra, n_rb, n_bt);
#endif
    for (uint32_t base = blockIdx.x * TRT_SHADE_BLOCK; base < A.n; base += per_grid) {
        const uint32_t i = base + threadIdx.x;
        f4 hit4, ra, rb, bt;
#if TRT_SHADE_PIPE
        hit4 = n_hit; ra = n_ra; rb = n_rb; bt = n_bt;
#else
        loadTile(i, hit4, ra, rb, bt);
#endif
        if (!staged || !rows_visible) {
            // the staged tables lie packed, in LDS layout, in one device buffer (trt_create): 16-byte words, coalesced
            if (!staged) {
                f4* l = reinterpret_cast<f4*>(s_tab);
                for (uint32_t w = threadIdx.x; w < A.lds_image_words; w += TRT_SHADE_BLOCK) l[w] = A.lds_image[w];
            }
            __syncthreads();
            staged = true;
            rows_visible = true;
        }
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(hit4.x), "+v"(ra.x), "+v"(rb.x), "+v"(bt.x));  // keeps the four loads up here (the compiler would sink them behind the miss test)
#endif
#if TRT_SHADE_PIPE
        loadTile(i + per_grid, n_hit, n_ra, n_rb, n_bt);  // n <= 0x7FFF0000 and per_grid <= 2^25: no wrap-around
#endif
        ShadeCtx c;
        c.had_hit = c.shade_ok = c.add_L = false;
        if (i < A.n) {
            if (A.primary) {
                // every path passes here exactly once, hit or miss: L starts at 0 (+ the radiance of a directly
                // visible light, pathTracing.cpp:9-12 through main.cpp:101)
                primaryRay(sc, A.td, A.s0, i, ra, rb, rows);
                shadeBegin(sc, A.td, A.s0, ra, rb, bt, hit4, c, rows);
                A.Lacc[i] = c.add_L ? mk4(0.0f + c.addL.x, 0.0f + c.addL.y, 0.0f + c.addL.z, 0.0f) : mk4(0.0f, 0.0f, 0.0f, 0.0f);
            } else {
                shadeBegin(sc, A.td, A.s0, ra, rb, bt, hit4, c, rows);
                if (c.add_L) {
                    f4 L = A.Lacc[c.pid];
                    L.x = L.x + c.addL.x; L.y = L.y + c.addL.y; L.z = L.z + c.addL.z;
                    A.Lacc[c.pid] = L;
                }
            }
            if (c.had_hit) bounce_depth = c.depth;
        }
        const unsigned long long ok_ballot = ballotb(c.shade_ok);
        if ((threadIdx.x & 63u) == 0 && ok_ballot) atomicAdd(&s_shaded, (uint32_t)__popcll(ok_ballot));
        if (ballotb(c.had_hit) && (threadIdx.x & 63u) == 0) s_anyhit = 1;

        // Queue slots come from one atomicAdd per block and queue (blockStage), and that atomic's round trip
        // is taken off the critical path: the rays of a stage are written one stage later, after the NEXT
        // stage's compute and barrier, by when the base offset has long arrived.
        //   stage l (one per light): NEE sample (pathTracing.cpp:34-74) -> shadow queue l
        //   last stage: RR(0.8) then nextRay (pathTracing.cpp:78-99) -> next bounce's queue
        uint32_t pend_base = 0;   // thread 0: value returned by the previous stage's atomic
        const uint32_t* pend_s = nullptr;
        bool pend_emit = false;
        uint32_t pend_rank = 0, pend_li = 0;
        f3 pend_wo = mk3(0, 0, 0), pend_w = mk3(0, 0, 0);
        float pend_tmax = TRT_INF;  // TRT_FLAG_FIXED_NEE: how far the occlusion test of the shadow ray reaches
        const uint32_t nl = ONE_LIGHT ? 1u : sc.n_lights;
        for (uint32_t li = 0; li + 1 < nl; ++li) {  // every light but the last: a stage of its own
            bool emit = false;
            f3 wo = mk3(0, 0, 0), contrib = mk3(0, 0, 0);
            float t_max = TRT_INF;
            if (c.shade_ok) emit = lightSample(sc, c.vx, *c.m, li, c.rng, wo, contrib, A.td.fixed_nee != 0u, t_max);
            uint32_t rank;
            uint32_t* s = blockStage<TRT_SHADE_BLOCK>(emit, A.shadow_counts + (size_t)li * A.shadow_count_stride, s_cnt, parity, rank, pend_s, pend_base);
            parity = parity == 2 ? 0 : parity + 1;
            if (pend_emit) {
                const uint32_t slot = pend_s[TRT_SHADE_BLOCK / 64] + pend_s[threadIdx.x >> 6] + pend_rank;
                const f3 so = rayOrigin(c, pend_wo);  // Q6: the hit point itself unless TRT_FLAG_RAY_OFFSET
                TRT_STQ(2, A.sq[pend_li].sa + slot, mk4(so.x, so.y, so.z, pend_wo.x));
                TRT_STQ(2, A.sq[pend_li].sb + slot, mk4(pend_wo.y, pend_wo.z, u2f(c.pid), pend_tmax));
                TRT_STQ(2, A.sq[pend_li].sw + slot, mk4(pend_w.x, pend_w.y, pend_w.z, 0.0f));
            }
            pend_s = s; pend_emit = emit; pend_rank = rank; pend_li = li; pend_wo = wo; pend_w = c.beta * contrib; pend_tmax = t_max;
        }
        // Last stage: the last light's NEE sample AND the extension ray, reserved together (blockStage2).  The extension
        // ray comes in two halves around the reservation: the decision (RR, lobe draws) before the barrier, the direction
        // (Sample / refract: the long part) behind the atomic, whose round trip it covers.  The rays of this stage are
        // stored after the last barrier: a wave that has stores in flight when it needs an atomic's result waits for
        // vmcnt(0), i.e. for the stores as well, and the block waits for that wave.
        bool emit_s = false;
        f3 wo_s = mk3(0, 0, 0), w_s = mk3(0, 0, 0);
        float tmax_s = TRT_INF;
        if (nl && c.shade_ok) {
            f3 contrib = mk3(0, 0, 0);
            emit_s = lightSample(sc, c.vx, *c.m, nl - 1u, c.rng, wo_s, contrib, A.td.fixed_nee != 0u, tmax_s);
            w_s = c.beta * contrib;
        }
        NextPlan plan;
        const bool emit_next = shadeNextDecide(c, A.max_depth, plan);
        uint32_t rank_next, rank_s;
        unsigned long long pair_base = 0;  // thread 0
        uint32_t* s2 = s_cnt2 + parity2 * (2 * (TRT_SHADE_BLOCK / 64) + 2);
        parity2 ^= 1;
        blockStage2<TRT_SHADE_BLOCK>(emit_next, emit_s, A.pair_count, s2, rank_next, rank_s, pend_s, pend_base, pair_base);
        if (pend_emit) {  // the last light but one (its base was published before the barrier above)
            const uint32_t slot = pend_s[TRT_SHADE_BLOCK / 64] + pend_s[threadIdx.x >> 6] + pend_rank;
            const f3 so = rayOrigin(c, pend_wo);
            TRT_STQ(2, A.sq[pend_li].sa + slot, mk4(so.x, so.y, so.z, pend_wo.x));
            TRT_STQ(2, A.sq[pend_li].sb + slot, mk4(pend_wo.y, pend_wo.z, u2f(c.pid), pend_tmax));
            TRT_STQ(2, A.sq[pend_li].sw + slot, mk4(pend_w.x, pend_w.y, pend_w.z, 0.0f));
        }
        f4 nra = mk4(0, 0, 0, 0), nrb = nra, nbt = nra;
        if (emit_next) shadeNextFinish(c, plan, nra, nrb, nbt);
        if (threadIdx.x == 0) {  // publish both bases
            s2[TRT_SHADE_BLOCK / 64] = (uint32_t)pair_base;
            s2[2 * (TRT_SHADE_BLOCK / 64) + 1] = (uint32_t)(pair_base >> 32);
        }
        ldsBarrier();
        if (emit_s) {
            const uint32_t slot = s2[2 * (TRT_SHADE_BLOCK / 64) + 1] + s2[TRT_SHADE_BLOCK / 64 + 1 + (threadIdx.x >> 6)] + rank_s;
            const f3 so = rayOrigin(c, wo_s);
            TRT_STQ(2, A.sq[nl - 1u].sa + slot, mk4(so.x, so.y, so.z, wo_s.x));
            TRT_STQ(2, A.sq[nl - 1u].sb + slot, mk4(wo_s.y, wo_s.z, u2f(c.pid), tmax_s));
            TRT_STQ(2, A.sq[nl - 1u].sw + slot, mk4(w_s.x, w_s.y, w_s.z, 0.0f));
        }
        if (emit_next) {
            const uint32_t slot = s2[TRT_SHADE_BLOCK / 64] + s2[threadIdx.x >> 6] + rank_next;
            TRT_STQ(2, A.qout.ra + slot, nra);
            TRT_STQ(2, A.qout.rb + slot, nrb);
            TRT_STQ(2, A.qout.bt + slot, nbt);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_shaded) atomicAdd(&A.stats->shaded_hits, (unsigned long long)s_shaded);
        if (s_anyhit) atomicMax(&A.stats->max_depth_hit, bounce_depth);
    }
}

// ------------------------------------------------------------- tail ----
// The geometric tail of a pass: once few paths are left (a small fraction of a
// percent of the work, but dozens of bounces), every remaining path is finished
// by one lane in ONE launch instead of three launches and a host round trip per
// bounce.  Same device functions, same order of operations per path as the
// wavefront kernels (trace, shadeBegin, per light: sample + shadow trace + add,
// shadeNext), so the result is unchanged.
struct TailArgs {
    RayQueue q;
    uint32_t n;
    f4* Lacc;
    TileDesc td;
    uint32_t s0;
    int32_t max_depth;
    uint32_t* spill;
    uint32_t spill_stride;
    uint32_t uniform;  // tiny scene: traverse with the wave-uniform walk (scalar loads) instead of the per-lane stack
    DeviceStats* stats;
};

template <bool COUNT, int NK>
__global__ __launch_bounds__(TRT_TRACE_BLOCK) void k_tail(SceneDev sc, TailArgs A)
{
    __shared__ __attribute__((aligned(16))) uint32_t smem[TRT_LDS_STACK_MAX * TRT_TRACE_BLOCK];  // the stacks, or (uniform) the candidate queues
    static_assert(TRT_LDS_STACK_MAX >= TRT_PEND_SLOTS * 4, "candidate queue must fit the stack area");
    LdsStack<TRT_LDS_STACK_MAX, true> stk;
    stk.lds = smem + threadIdx.x;
    stk.spill = A.spill + (size_t)blockIdx.x * TRT_TRACE_BLOCK + threadIdx.x;
    stk.spill_stride = A.spill_stride;
    uint32_t ni[2] = {0, 0}, nt[2] = {0, 0};
    unsigned long long n_shadow = 0, n_indirect = 0, n_shaded = 0;
    uint32_t deepest = 0;
    bool any = false;
    const uint32_t stride = gridDim.x * TRT_TRACE_BLOCK;
    for (uint32_t i = blockIdx.x * TRT_TRACE_BLOCK + threadIdx.x; i < A.n; i += stride) {
        f4 ra = A.q.ra[i], rb = A.q.rb[i], bt = A.q.bt[i];
        const uint32_t pid = f2u(rb.z);
        f4 L = A.Lacc[pid];
        for (;;) {
            Hit h;
            if (A.uniform) {  // the few rays still alive are bound by the latency of a bounce: no per-lane gathers, no stack
                const f3 o = mk3(ra.x, ra.y, ra.z), d = mk3(ra.w, rb.x, rb.y);
                h.t = TRT_INF; h.tri = -1; h.flags = 0u; h.u = 0.f; h.v = 0.f;
                uniformWalk<COUNT>(sc, o, d, true, reinterpret_cast<f4*>(smem) + threadIdx.x, h.t, h.tri, h.flags, ni[0], nt[0]);
                if (h.tri >= 0) {
                    float t, un, vn, det;
                    if (triTest(sc.tri_isect[h.tri], o, d, t, un, vn, det)) { h.u = un / det; h.v = vn / det; }
                }
            } else {
                h = traceClosest<LdsStack<TRT_LDS_STACK_MAX, true>, COUNT, NK>(sc, mk3(ra.x, ra.y, ra.z), mk3(ra.w, rb.x, rb.y), stk, ni[0], nt[0]);
            }
            ShadeCtx c;
            shadeBegin(sc, A.td, A.s0, ra, rb, bt, mk4(h.t, u2f((uint32_t)h.tri), h.u, h.v), c);
            if (c.had_hit) { any = true; deepest = c.depth > deepest ? c.depth : deepest; }
            if (c.add_L) { L.x = L.x + c.addL.x; L.y = L.y + c.addL.y; L.z = L.z + c.addL.z; }
            if (c.shade_ok) n_shaded++;
            for (uint32_t li = 0; li < sc.n_lights; ++li) {
                f3 wo, contrib;
                const bool fixed = A.td.fixed_nee != 0u;
                float t_max = TRT_INF;
                if (!c.shade_ok || !lightSample(sc, c.vx, *c.m, li, c.rng, wo, contrib, fixed, t_max)) continue;
                const f3 w = c.beta * contrib;
                n_shadow++;
                Hit sh;
                if (A.uniform) {
                    sh.t = fixed ? t_max : TRT_INF; sh.tri = -1; sh.flags = 0u; sh.u = 0.f; sh.v = 0.f;
                    uniformWalk<COUNT>(sc, rayOrigin(c, wo), wo, true, reinterpret_cast<f4*>(smem) + threadIdx.x, sh.t, sh.tri, sh.flags, ni[1], nt[1]);
                } else {
                    sh = traceClosest<LdsStack<TRT_LDS_STACK_MAX, true>, COUNT, NK>(sc, rayOrigin(c, wo), wo, stk, ni[1], nt[1], t_max, fixed, !fixed);
                }
                if (fixed ? sh.tri < 0 : (sh.tri >= 0 && (sh.flags >> 8) == (uint32_t)sc.lights[li].mat)) { L.x = L.x + w.x; L.y = L.y + w.y; L.z = L.z + w.z; }
            }
            f4 nra, nrb, nbt;
            if (!shadeNext(c, A.max_depth, nra, nrb, nbt)) break;
            ra = nra; rb = nrb; bt = nbt;
            n_indirect++;
        }
        A.Lacc[pid] = L;
    }
    const unsigned long long s_sh = waveSum(n_shadow), s_in = waveSum(n_indirect), s_sd = waveSum(n_shaded);
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (COUNT) { c0 = waveSum(ni[0]); c1 = waveSum(nt[0]); c2 = waveSum(ni[1]); c3 = waveSum(nt[1]); }
    if ((threadIdx.x & 63) == 0) {
        if (s_sh) atomicAdd(&A.stats->tail_rays_shadow, s_sh);
        if (s_in) atomicAdd(&A.stats->tail_rays_indirect, s_in);
        if (s_sd) atomicAdd(&A.stats->shaded_hits, s_sd);
        if (COUNT) {
            atomicAdd(&A.stats->inner_visits[0], c0); atomicAdd(&A.stats->tri_tests[0], c1);
            atomicAdd(&A.stats->inner_visits[1], c2); atomicAdd(&A.stats->tri_tests[1], c3);
        }
    }
    uint32_t dm = any ? deepest + 1u : 0u;  // wave maximum, then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(dm, off); dm = o > dm ? o : dm; }
    if ((threadIdx.x & 63) == 0 && dm) atomicMax(&A.stats->max_depth_hit, dm - 1u);
}

// (Round 4 built a second form of this kernel — a GROUP of G >= 1 + lights lanes per path, every lane carrying the same path state, lane 0
// tracing the extension ray into vertex k + 1 while lane 1 + l traces the shadow ray of light l from vertex k, results exchanged by wave
// shuffles, the radiance sum in k_tail's order: bit-identical, the whole -m gpu suite green — and measured it: back 0.59 -> 0.55 ms,
// veach-mis 0.81 -> 0.77, staircase 4.57 -> 4.55, soup 8.9 -> 10.4, the 10 M mesh 2.5 -> 3.3.  The tail is not the latency of its longest
// path alone: with G times the lanes its early, populous bounces become throughput-bound.  Deleted (commit history has it);
// profiles/r04_ab_shade_tail.txt has the table, with the sweep of the hand-over point TRT_TAIL_N: 131 072 is best on every scene.)

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

// The queue lengths of a bounce, pushed to the host: counters (b, c) and (b + 1, c), c = 0..n-1, into a pinned,
// device-visible host buffer (out[2c], out[2c+1]), then a sequence number — the host spins on that word instead of
// paying a DMA copy plus a stream-synchronise wake-up per bounce (one wave, launched behind k_shade).
__global__ __launch_bounds__(64) void k_publish_counts(const uint32_t* __restrict__ counts, uint32_t stride, uint32_t b, uint32_t n,
                                                        const unsigned long long* __restrict__ pair, volatile uint32_t* out, uint32_t seq)
{
    const uint32_t t = threadIdx.x;
    if (t < 2u * n) {
        const uint32_t c = t >> 1, which = t & 1u;
        uint32_t v = counts[(size_t)c * stride + b + which];
        // the two counters k_shade keeps in one word (ShadeArgs::pair_count): queue length of bounce b + 1, shadow rays of the last light
        if (c == 0u && which == 1u) v = (uint32_t)pair[0];
        if (n > 1u && c == n - 1u && which == 0u) v = (uint32_t)(pair[0] >> 32);
        out[t] = v;
    }
    __threadfence_system();
    __syncthreads();
    if (t == 0) {
        __hip_atomic_store(const_cast<uint32_t*>(out) + 2u * 16u, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
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
