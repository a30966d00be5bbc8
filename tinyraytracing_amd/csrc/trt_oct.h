// trt_oct.h — the 8-wide compressed BVH node of the per-lane traversal (node kind 1) and its slab test in the node's
// quantised frame.  Layout and traversal bookkeeping follow the published compressed-wide-BVH scheme (Ylitie, Karras, Laine:
// "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", HPG 2017): 80 bytes = five 16-B loads for EIGHT
// children (the exact 4-wide node: seven loads for four), children and triangles addressed implicitly (base + popcount), one
// 8-byte stack entry per node instead of one per child, octant-ordered descent instead of a sort.  What is specific to this
// repository is the exactness argument below: the reference's hit (bvh.cpp:146-245) must come out bit for bit.
//
//   q0 = (p.x, p.y, p.z, bits: ex | ey << 8 | ez << 16 | imask << 24)     frame origin, per-axis scale 2^(e-127), inner-child slots
//   q1 = (bits child_base, bits tri_base, meta[0..3], meta[4..7])          first inner child (onodes), first triangle (tri_trav)
//   q2 = (qlo.x[0..3], qlo.x[4..7], qlo.y[0..3], qlo.y[4..7])              one byte per child and bound:
//   q3 = (qlo.z[0..3], qlo.z[4..7], qhi.x[0..3], qhi.x[4..7])                  bound = p + q * 2^(e-127)
//   q4 = (qhi.y[0..3], qhi.y[4..7], qhi.z[0..3], qhi.z[4..7])
//   meta[i]: 0 = empty slot; inner child: 0b001sssss with sssss = 24 + i; leaf slot: (unary triangle count, 1..3 bits) << 5 | offset of
//   its first triangle from tri_base (0..23).  The low five bits are the child's position in the 32-bit hit mask of a visit: bits
//   24..31 inner children (xor-ed with the ray's octant so that the highest set bit is the child to enter first), bits 0..23 one
//   bit per triangle.
//
// Exactness.  The reference enters a leaf iff the ray passes the box test of every node on the leaf's root path (bvh.cpp:156-166),
// each on the caller's exact box with the arithmetic of boxTest() (trt_path.h).  Here a visit tests the QUANTISED boxes with
// DIFFERENT arithmetic (one fma per bound in the node's frame), so two things are needed:
//  (1) No false negatives: if the reference passes a box, octVisit() passes the quantised box around it, and the entry distance it
//      uses for culling is not larger than the reference's.  Stored boxes contain the exact ones (builder, checked in binary64:
//      p + qlo s <= lo, p + qhi s >= hi); per axis the near plane is computed as fma(q, s inv, (p - o) inv - m) and the far plane
//      with + m, where m = 2^-21 |(p - o) inv| + 2^-13 |s inv| + 2^-100 exceeds the rounding error of BOTH computations (ours: two
//      roundings of the origin term, one of the fma; the reference's: (lo - o) rounded, times inv rounded; |values| <=
//      |(p - o) inv| + 255 |s inv|; u = 2^-24: 6 u |org| + 765 u |s inv| needed, 8 u |org| + 2048 u |s inv| taken).  Where a product
//      overflows the planes come out as NaN or as the infinity on their own side, and fmaxf / fminf drop NaNs: the axis then
//      constrains nothing, which only enlarges the visited set.
//      A direction component that is exactly zero (1 / d = +-inf: about one camera ray in 40 000 of an axis-aligned camera, where
//      d.x is a difference of numbers near 278 and comes out as a multiple of 3e-5) gets the reference's own meaning: its slab test
//      yields (-inf, +inf) when the origin lies between the planes (NaN, dropped, when it lies ON one) and two infinities of one sign
//      — a miss, unless every axis is like that, and then no triangle test can hit either — when it lies outside.  Here the
//      reciprocal of such an axis is replaced by K = 2^40 and both margins grow by 2^30 (beyond every culling bound, 2^24 at most
//      for coordinates below 2^40): an origin between the stored planes, on them or less than 2^-10 outside gives near < 0 and
//      far > every bound — no constraint — and one further outside |near| or |far| beyond 2^30, a miss, as in the reference.
//      (Leaving such an axis out altogether, the first version, was exact too, but a ray along an axis then visited every node
//      of a slab of the scene: 1 500 nodes and 4 600 triangles for the centre column of the 2 M-triangle mesh, one lane for
//      milliseconds.)  Scenes whose coordinates reach 2^40 are not given this node kind.  So the traversal reaches a SUPERSET of
//      the reference's leaves and never culls a node that holds a hit which counts (trt_cull_bound, trt_prims.h).
//  (2) The extra leaves must not contribute: a hit counts only if the ray also passes the reference's test of the exact box of the
//      triangle's own leaf — for a nested tree (the only kind that gets this node kind) equivalent to passing every box above it —
//      and the leaf-box rule holds.  As with that rule (trt_path.h, traceClosest) the RESULT of a ray is checked once, when it is
//      written back: a hit that should not count can only matter if the traversal ends with it, and then the ray is traced again in
//      the exact form on the exact 4-wide tree (k_trace_fix / traceClosestPass<RULE>).  A triangle hit whose leaf box the ray
//      misses needs the Moller-Trumbore test and the slab test to contradict each other by rounding: about as rare as the hits
//      the leaf-box rule exists for.
#pragma once
#include "trt_path.h"

namespace trtd {

// Packed at a stride of 80 B: a node straddles two 128-B lines 3 times in 8.  One node per line (stride 128 B, measured:
// profiles/r03_ab_oct.txt (6)) fetches fewer lines but takes 1.6x the cache: soup +2.6 %, the meshes -0.6 / -1.2 %, staircase -4.4 %, veach-mis -3.5 %.
struct OctNode {
    f4 q[5];
};
static_assert(sizeof(OctNode) == 80, "OctNode is five 16-byte words");

#define TRT_OCT_MAX_LEAF_TRIS 3u

// Per-ray constants of the quantised-frame test.
struct OctRay {
    f3 o, inv;         // inv: 1 / d, an infinite one (d = +-0) replaced by TRT_OCT_DEAD_K
    uint32_t octinv4;  // (7 - octant) in each of the four bytes; octant bit 2 = d.x < 0, bit 1 = d.y < 0, bit 0 = d.z < 0.  A CLEAR bit
                       // of (7 - octant) therefore says: that component of d is negative (the near plane of the axis is the box's upper one)
};
#define TRT_OCT_DEAD_K 1.099511627776e12f   /* 2^40: stands in for 1 / 0 (header, (1)) */
#define TRT_OCT_DEAD_M 1.073741824e9f      /* 2^30: what the margins of such an axis grow by */
// +-inf -> +K: `d < 0` is false for -0 as well, so the axis counts as a positive one and its near plane is the lower one
TRT_HD inline float octSafeInv(float inv) { return fabsf(inv) <= 3.4028234e38f ? inv : (inv != inv ? inv : TRT_OCT_DEAD_K); }
TRT_HD inline OctRay makeOctRay(f3 o, f3 d, f3 inv)
{
    OctRay r;
    r.o = o;
    r.inv = mk3(octSafeInv(inv.x), octSafeInv(inv.y), octSafeInv(inv.z));
    const uint32_t oct = (d.x < 0.0f ? 4u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 1u : 0u);
    r.octinv4 = (7u - oct) * 0x01010101u;
    return r;
}

// A traversal "group": nodes — x = index of the first inner child of the node the entries came from, y = hit bits 24..31 (by
// octant-permuted slot) | imask in bits 0..7; triangles — x = first triangle (tri_trav), y = one bit per triangle (bits 0..23).
struct OctGroup {
    uint32_t x, y;
};

TRT_HD inline float octByte(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xFFu); }

// One visit of node `ni`: the hit mask of its eight children (see the header): inner children in bits 24..31 at position
// 24 + (slot ^ octinv), triangles of leaf children in bits 0..23.  `cull` = trt_cull_bound(best hit so far).
TRT_HD inline void octVisit(const OctNode* __restrict__ nodes, uint32_t ni, const OctRay& R, float cull, OctGroup& ng, OctGroup& tg)
{
    const f4* q = nodes[ni].q;
    const f4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
    const uint32_t ew = f2u(q0.w);
    // ray in the node's frame: t(bound byte b) = b * idir + org, widened by m (header, (1))
    const float idx = u2f((ew & 0xFFu) << 23) * R.inv.x, idy = u2f(((ew >> 8) & 0xFFu) << 23) * R.inv.y, idz = u2f(((ew >> 16) & 0xFFu) << 23) * R.inv.z;
    const float ogx = (q0.x - R.o.x) * R.inv.x, ogy = (q0.y - R.o.y) * R.inv.y, ogz = (q0.z - R.o.z) * R.inv.z;
    const float cx = fabsf(R.inv.x) == TRT_OCT_DEAD_K ? TRT_OCT_DEAD_M : 7.8886090522101181e-31f;  // 2^-100, or the margin of an axis with d = 0
    const float cy = fabsf(R.inv.y) == TRT_OCT_DEAD_K ? TRT_OCT_DEAD_M : 7.8886090522101181e-31f;
    const float cz = fabsf(R.inv.z) == TRT_OCT_DEAD_K ? TRT_OCT_DEAD_M : 7.8886090522101181e-31f;
    const float mx = fmaf(fabsf(idx), 1.220703125e-4f, fmaf(fabsf(ogx), 4.76837158203125e-7f, cx));
    const float my = fmaf(fabsf(idy), 1.220703125e-4f, fmaf(fabsf(ogy), 4.76837158203125e-7f, cy));
    const float mz = fmaf(fabsf(idz), 1.220703125e-4f, fmaf(fabsf(ogz), 4.76837158203125e-7f, cz));
    const float onx = ogx - mx, ony = ogy - my, onz = ogz - mz, ofx = ogx + mx, ofy = ogy + my, ofz = ogz + mz;
    // near / far plane bytes by the sign of the direction, four children per word
    const bool nx = (R.octinv4 & 4u) == 0, ny = (R.octinv4 & 2u) == 0, nz = (R.octinv4 & 1u) == 0;
    const uint32_t lox0 = f2u(q2.x), lox1 = f2u(q2.y), loy0 = f2u(q2.z), loy1 = f2u(q2.w), loz0 = f2u(q3.x), loz1 = f2u(q3.y);
    const uint32_t hix0 = f2u(q3.z), hix1 = f2u(q3.w), hiy0 = f2u(q4.x), hiy1 = f2u(q4.y), hiz0 = f2u(q4.z), hiz1 = f2u(q4.w);
    const uint32_t nrx[2] = {nx ? hix0 : lox0, nx ? hix1 : lox1}, frx[2] = {nx ? lox0 : hix0, nx ? lox1 : hix1};
    const uint32_t nry[2] = {ny ? hiy0 : loy0, ny ? hiy1 : loy1}, fry[2] = {ny ? loy0 : hiy0, ny ? loy1 : hiy1};
    const uint32_t nrz[2] = {nz ? hiz0 : loz0, nz ? hiz1 : loz1}, frz[2] = {nz ? loz0 : hiz0, nz ? loz1 : hiz1};
    const uint32_t meta[2] = {f2u(q1.z), f2u(q1.w)};
    uint32_t hits = 0u;
    TRT_UNROLL
    for (int h = 0; h < 2; ++h) {
        // four children at once on the packed meta bytes: inner children (0b001xxxxx) get their position xor-ed with the octant
        const uint32_t m4 = meta[h];
        const uint32_t is_inner4 = (m4 & (m4 << 1)) & 0x10101010u;            // bit 4 of a byte: bits 3 and 4 both set <=> 24 <= low5 (inner)
        const uint32_t inner_mask4 = (is_inner4 >> 4) * 0xFFu;                  // 0xFF in the bytes of inner children
        const uint32_t pos4 = (m4 ^ (R.octinv4 & inner_mask4)) & 0x1F1F1F1Fu;   // position in the hit mask
        const uint32_t bits4 = (m4 >> 5) & 0x07070707u;                         // what to set there: 1 (inner), unary count (leaf), 0 (empty)
        // (two children per v_pk_fma_f32 — 5.1 clocks per wave against 2 x 4.2, tools/valu_probe.hip — measured slower: the pairs have to
        // be assembled first; profiles/r03_ab_oct.txt.  Removed.)
        TRT_UNROLL
        for (int k = 0; k < 4; ++k) {
            const float tnx = fmaf(octByte(nrx[h], k), idx, onx), tny = fmaf(octByte(nry[h], k), idy, ony), tnz = fmaf(octByte(nrz[h], k), idz, onz);
            const float tfx = fmaf(octByte(frx[h], k), idx, ofx), tfy = fmaf(octByte(fry[h], k), idy, ofy), tfz = fmaf(octByte(frz[h], k), idz, ofz);
            const float tmin = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
            const float tmax = fminf(fminf(tfx, tfy), fminf(tfz, cull));
            const bool hit = !(tmin > tmax);
            const uint32_t b = (bits4 >> (8 * k)) & 0xFFu, p = (pos4 >> (8 * k)) & 0xFFu;
            hits |= hit ? (b << p) : 0u;
        }
    }
    ng.x = f2u(q1.x);
    ng.y = (hits & 0xFF000000u) | (ew >> 24);
    tg.x = f2u(q1.y);
    tg.y = hits & 0x00FFFFFFu;
}

// The inner child to enter next: highest set bit of the hit byte (nearest octant first); clears it in `ng`.
TRT_HD inline uint32_t octNextChild(OctGroup& ng, const OctRay& R)
{
    const uint32_t bit = 31u - trt_clz32(ng.y);
    ng.y &= ~(1u << bit);
    const uint32_t slot = (bit - 24u) ^ (R.octinv4 & 7u);
    const uint32_t imask = ng.y & 0xFFu;
    const uint32_t rel = trt_popc32(imask & ((1u << slot) - 1u));
    return ng.x + rel;
}

// Triangle record of the oct traversal: TriIsect with c.w = original (post-BVH) index | position in the CALLER's leaf << 27 (0..14) and
// the size of that leaf (1..15) in bits 1..4 of the flags word c.z (bits the other users of the flags — emissive bit 0, material
// bits 8.. — do not read).  tri_trav holds every node's leaf triangles next to each other, each slot's in index order.  A caller's leaf
// of 4..15 triangles (the reference builds with 8, main.cpp:76) is laid out as several slots of <= 3 triangles (trt_oct_build.h), so
// position and size speak of the caller's leaf, not of the slot: the tie rule below needs to know the reference's leaf.
TRT_HD inline uint32_t octTriOrig(uint32_t w) { return w & 0x07FFFFFFu; }
TRT_HD inline uint32_t octTriPos(uint32_t w) { return (w >> 27) & 15u; }
TRT_HD inline uint32_t octTriCount(uint32_t fl) { return (fl >> 1) & 15u; }

// Fold of one triangle candidate into the best hit: interactBVHNode's scan (bvh.cpp:219) and traverseBVH's merge
// (bvh.cpp:168-172) in one step per candidate, in a form that does not depend on the order the candidates arrive in (the slots of
// one caller's leaf are entered in octant order, not index order).  Nearer wins.  At equal distance the reference's scan of a leaf
// keeps the FIRST candidate unless a later one is emissive, and then the LAST emissive one; its merge of siblings keeps the
// leftmost leaf whose result is emissive, else the rightmost leaf's.  As one total order of preference: emissive before not;
// among emissive ones the leftmost leaf, inside a leaf the highest index; among the others the rightmost leaf, inside a leaf the
// lowest index.  Leaves cover disjoint index ranges in tree order, so "left of" is "index below" (the same fold as
// uniformWalk()'s, trt_kernels.h, which sees whole leaves).
TRT_HD inline void octFold(float t, uint32_t w, uint32_t fl, float& best_t, int32_t& best_tri, uint32_t& best_flags)
{
    const int32_t j = (int32_t)octTriOrig(w);
    bool take = t < best_t;
    if (t == best_t && best_tri >= 0) {
        const bool em = (fl & 1u) != 0, bem = (best_flags & 1u) != 0;
        const int32_t first = j - (int32_t)octTriPos(w);
        const bool same_leaf = best_tri >= first && best_tri < first + (int32_t)octTriCount(fl);
        const bool below = j < best_tri;
        take = em ? (!bem || (below != same_leaf)) : (!bem && (below == same_leaf));
    }
    if (take) { best_t = t; best_tri = j; best_flags = fl; }
}

// Where a light's triangles can begin.  Union of the (caller's, exact) boxes of the leaves that hold the triangles of one light's
// material, per shadow launch.  In parity mode a shadow ray is "visible" iff its CLOSEST counting hit carries that material (Q5).
// Every hit on such a triangle that counts lies at or behind floor(entry of its leaf's box) (the leaf-box rule, trt_path.h), and the
// entry of a box that contains the leaf's is no later (subtraction, multiplication, min and max are monotone; where a NaN of 0 * inf
// makes the two tests differ, the leaf's fails and none of its hits count), nor is its floor (trt_leaf_floor is monotone).  So once the
// search holds ANY hit nearer than stop = floor(entry of this box), the closest hit is not on the light: the ray is occluded and the
// search ends — provided that hit counts, which the store checks as for every result (octResultCounts; otherwise the exact form runs).
// A ray that does not pass this box at all has no counting hit on the light and needs no search.
struct LightBox {
    float lo[3], hi[3];
};

// Stack of the oct traversal: one 8-byte group per level.  push(sp, g) / pop(sp).
// Per-lane closest-hit search on the oct tree (the order of operations of the wave driver in trt_kernels.h, one lane): returns
// the hit WITHOUT the final check (see traceClosestOct).  Same `t_init` / `any` / `redo` meaning as traceClosestPass.
template <class Stack, bool COUNT>
TRT_HD inline Hit traceOctPass(const SceneDev& sc, f3 o, f3 d, Stack& stk, uint32_t& n_inner, uint32_t& n_tri, float t_init, bool any, bool redo,
                               const LightBox* lbox = nullptr)
{
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const OctRay R = makeOctRay(o, d, inv);
    float best_t = t_init;
    int32_t best_tri = -1;
    uint32_t best_flags = 0u;
    float stop_t = -TRT_INF;
    if (lbox && !any) {
        float e;
        if (!boxTest(lbox->lo[0], lbox->lo[1], lbox->lo[2], lbox->hi[0], lbox->hi[1], lbox->hi[2], o, inv, e)) {
            Hit h;
            h.t = best_t; h.tri = -1; h.u = 0.f; h.v = 0.f; h.flags = 0u;
            return h;
        }
        stop_t = trt_leaf_floor(e, sc.leaf_alpha);
    }
    for (;;) {  // once; twice when `redo` and nothing lies in front of t_init
        int sp = 0;
        OctGroup ng, tg;
        ng.x = 0u; ng.y = 0x80000000u;  // the root: "child 31" of a group whose first child is node 0
        tg.x = 0u; tg.y = 0u;
        bool stop = false;
        for (;;) {
            if (ng.y & 0xFF000000u) {
                const uint32_t ni = octNextChild(ng, R);
                if (ng.y & 0xFF000000u) stk.push(sp++, ng);
                if (COUNT) n_inner++;
                octVisit(sc.onodes, ni, R, trt_cull_bound(best_t, sc.leaf_alpha), ng, tg);
            }
            while (tg.y) {
                const uint32_t b = trt_ctz32(tg.y);
                tg.y &= tg.y - 1u;
                const TriIsect T = sc.tri_trav[tg.x + b];
                if (COUNT) n_tri++;
                float t, un, vn, det;
                if (triTest(T, o, d, t, un, vn, det)) octFold(t, f2u(T.c.w), f2u(T.c.z), best_t, best_tri, best_flags);
            }
            if (best_tri >= 0 && (any || best_t < stop_t)) { stop = true; break; }
            if (!(ng.y & 0xFF000000u)) {
                if (sp == 0) break;
                ng = stk.pop(--sp);
            }
        }
        if (stop || !redo || best_tri >= 0 || !(best_t < TRT_INF)) break;
        best_t = TRT_INF;
    }
    Hit h;
    h.t = best_t; h.tri = best_tri; h.u = 0.f; h.v = 0.f; h.flags = best_flags;
    if (best_tri >= 0) {
        float t, un, vn, det;
        if (triTest(sc.tri_isect[best_tri], o, d, t, un, vn, det)) { h.u = un / det; h.v = vn / det; }
    }
    return h;
}

// does the result (t, tri) of an oct traversal count?  The ray must pass the exact box of tri's leaf (the reference's own test of
// that leaf: for a nested tree the same as passing every box above it) and the hit must not lie in front of it (leafFloor()).
TRT_HD inline bool octResultCounts(const SceneDev& sc, float t, int32_t tri, f3 o, f3 inv)
{
    if (tri < 0) return true;
    const f4 a = sc.leaf_box[2 * (size_t)tri], b = sc.leaf_box[2 * (size_t)tri + 1];
    float e;
    const bool pass = boxTest(a.x, a.y, a.z, a.w, b.x, b.y, o, inv, e);
    return pass && !(t < trt_leaf_floor(e, sc.leaf_alpha));
}

// Closest hit through the oct tree, with the check of the result and the exact form (on the exact 4-wide tree) behind it.
// `StackW` is the 32-bit reference stack of traceClosestPass.
template <class StackO, class StackW, bool COUNT>
TRT_HD inline Hit traceClosestOct(const SceneDev& sc, f3 o, f3 d, StackO& stk, StackW& stkw, uint32_t& n_inner, uint32_t& n_tri, float t_init = TRT_INF,
                                  bool any = false, bool redo = false, const LightBox* lbox = nullptr)
{
    {   // a zero direction component: not on the quantised nodes (their slab arithmetic meets 0 * inf at planes of its own) — the exact nodes, or, where the origin
        // may lie on a box plane, the literal walk (trt_path.h); what k_trace_fix does with such a ray
        const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        if (raySpecial(inv)) {
            if (rayOnABoxPlane(sc, o, inv)) return traceClosestBvh2Glm<StackW, COUNT>(sc, o, d, stkw, n_inner, n_tri, t_init, any);
            return traceClosestPass<StackW, COUNT, 0, true>(sc, o, d, stkw, n_inner, n_tri, t_init, any, redo);
        }
    }
    const Hit h = traceOctPass<StackO, COUNT>(sc, o, d, stk, n_inner, n_tri, t_init, any, redo, lbox);
    if (octResultCounts(sc, h.t, h.tri, o, mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z))) return h;
    return traceClosestPass<StackW, COUNT, 0, true>(sc, o, d, stkw, n_inner, n_tri, t_init, any, redo);
}

}  // namespace trtd
