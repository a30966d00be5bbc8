// trt_oct_build.h — host side: collapses the caller's BVH2 (include/trt.h) into the 8-wide compressed nodes of trt_oct.h and lays
// the triangles out the way those nodes address them.  Used by trt_create (trt_api.hip) and tests/hostsim, so both walk the same tree.
//
// Collapse: the dynamic programme of trt_wide.h for eight children — which intermediate boxes to drop so that the sum of the
// half-areas of all wide nodes (the expected number of node visits) is minimal.  Dropping boxes cannot change a hit (trt_wide.h);
// this node kind is only built for NESTED trees with finite boxes (every other tree keeps the exact 4-wide nodes).
// Leaves: a slot holds <= 3 triangles.  A caller's leaf of 4..15 triangles — the reference's own trees have up to 8 (main.cpp:76,
// bvh.cpp:16-144) — is laid out as ceil(count / 3) slots over consecutive pieces of its range, EVERY ONE WITH THE LEAF'S OWN BOX
// (splitLargeLeaves): the reference tests every triangle of a leaf whose box the ray passes (bvh.cpp:151-154, 211-229), and a
// tighter box around a piece could skip a triangle whose Moller-Trumbore test would have accepted the ray — for a ray nearly in
// the triangle's plane the computed distance tn / det can lie anywhere along the ray, so no box smaller than the
// leaf's is safe.  With the leaf's box on every piece the visited set of triangles is exactly that of the caller's tree; the tie
// rule (octFold) and the check of the result (leaf_box) speak of the caller's leaf throughout.
// Slots: a child's slot index says on which side of the node's centre it lies (bit 2 = +x, bit 1 = +y, bit 0 = +z), assigned
// greedily by the projection of its centre on the slot's diagonal, so that "slot xor (7 - ray octant), highest first" enters the
// children roughly front to back without sorting (Ylitie et al. 2017).  The order is free: only the visited SET matters.
// Quantisation: frame origin p = the per-axis minimum of the children's boxes, scale 2^(e-127) the smallest power of two that fits
// the extent into 0..255; every byte is moved outward until p + q * scale contains the exact bound IN BINARY64 (exact arithmetic:
// a 24-bit and an 8-bit significand), which is the premise of trt_oct.h's no-false-negative argument.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <limits>
#include <memory>
#include <vector>

#include "trt_oct.h"
#include "trt_wide.h"

namespace trtd {

struct OctTree {
    std::vector<OctNode> nodes;     // nodes[0] is the root
    std::vector<TriIsect> tri_trav; // the triangles in node order: c.w = original index | position in the caller's leaf << 27, c.z |= that leaf's size << 1
    uint32_t levels = 0;            // nodes on the longest root path: the traversal stack needs levels - 1 entries
    bool ok = false;                // false: this tree keeps the exact 4-wide nodes (not nested / not finite / coordinates >= 2^40)
    uint32_t split_leaves = 0;      // caller's leaves of more than 3 triangles that were laid out as several slots
    const char* why = "";
};

// The caller's tree with every leaf of more than TRT_OCT_MAX_LEAF_TRIS triangles replaced by a balanced binary subtree over
// consecutive pieces of <= TRT_OCT_MAX_LEAF_TRIS triangles of its range, every box in that subtree the leaf's own (header).  New
// nodes are appended, so parents still precede children.  Empty result: nothing to split.
inline std::vector<trt_bvh_node> splitLargeLeaves(const trt_bvh_node* nodes2, uint32_t n_nodes2, uint32_t* n_split)
{
    std::vector<trt_bvh_node> out;
    *n_split = 0;
    bool any = false;
    for (uint32_t n = 0; n < n_nodes2 && !any; ++n)
        for (uint32_t ref : {nodes2[n].child0, nodes2[n].child1})
            if ((ref & TRT_LEAF_BIT) && TRT_LEAF_COUNT(ref) > TRT_OCT_MAX_LEAF_TRIS) any = true;
    if (!any) return out;
    out.assign(nodes2, nodes2 + n_nodes2);
    struct Piece { uint32_t first, count; };
    for (uint32_t n = 0; n < n_nodes2; ++n) {
        for (int k = 0; k < 2; ++k) {
            const uint32_t ref = k ? nodes2[n].child1 : nodes2[n].child0;
            if (!(ref & TRT_LEAF_BIT) || TRT_LEAF_COUNT(ref) <= TRT_OCT_MAX_LEAF_TRIS) continue;
            ++*n_split;
            const float* lo = k ? nodes2[n].lo1 : nodes2[n].lo0;
            const float* hi = k ? nodes2[n].hi1 : nodes2[n].hi0;
            const uint32_t first = TRT_LEAF_FIRST(ref), count = TRT_LEAF_COUNT(ref);
            const uint32_t m = (count + TRT_OCT_MAX_LEAF_TRIS - 1u) / TRT_OCT_MAX_LEAF_TRIS;  // 2..5 pieces, sizes as even as possible
            Piece pc[5];
            for (uint32_t i = 0, at = first; i < m; ++i) { pc[i].first = at; pc[i].count = count / m + (i < count % m ? 1u : 0u); at += pc[i].count; }
            // balanced subtree over pieces [a, b): returns its reference
            struct Rec {
                std::vector<trt_bvh_node>& out; const Piece* pc; const float* lo; const float* hi;
                uint32_t make(uint32_t a, uint32_t b)
                {
                    if (b - a == 1) return TRT_MAKE_LEAF(pc[a].first, pc[a].count);
                    const uint32_t me = (uint32_t)out.size();
                    out.emplace_back();
                    const uint32_t mid = a + (b - a + 1) / 2;
                    const uint32_t c0 = make(a, mid), c1 = make(mid, b);
                    trt_bvh_node& nd = out[me];
                    for (int x = 0; x < 3; ++x) { nd.lo0[x] = nd.lo1[x] = lo[x]; nd.hi0[x] = nd.hi1[x] = hi[x]; }
                    nd.child0 = c0; nd.child1 = c1;
                    nd.reserved[0] = nd.reserved[1] = 0u;
                    return me;
                }
            } rec{out, pc, lo, hi};
            const uint32_t sub = rec.make(0, m);
            if (k) out[n].child1 = sub; else out[n].child0 = sub;
        }
    }
    return out;
}

// `threads`: host threads to use (trt_wide.h, par); the tree does not depend on it.
inline OctTree buildOct(const trt_bvh_node* caller_nodes2, uint32_t caller_n_nodes2, uint32_t n_tris, const TriIsect* tri_isect, unsigned threads = 1)
{
    using namespace wide_detail;
    constexpr int W = 8;
    OctTree t;
    if (caller_n_nodes2 == 0) { t.why = "no nodes"; return t; }
    // position and size of every triangle's leaf IN THE CALLER'S TREE (the reference's leaf: tie rule, octFold)
    std::vector<uint8_t> leaf_info((size_t)std::max<uint32_t>(n_tris, 1u), 0);  // pos | count << 4
    par::forRange(caller_n_nodes2, threads, 65536, [&](size_t n0, size_t n1) {
        for (size_t n = n0; n < n1; ++n)
            for (uint32_t ref : {caller_nodes2[n].child0, caller_nodes2[n].child1}) {
                if (!(ref & TRT_LEAF_BIT)) continue;
                const uint32_t first = TRT_LEAF_FIRST(ref), count = TRT_LEAF_COUNT(ref);
                for (uint32_t i = 0; i < count && first + i < n_tris; ++i) leaf_info[first + i] = (uint8_t)(i | (count << 4));
            }
    });
    const std::vector<trt_bvh_node> split = splitLargeLeaves(caller_nodes2, caller_n_nodes2, &t.split_leaves);
    const trt_bvh_node* nodes2 = split.empty() ? caller_nodes2 : split.data();
    const uint32_t n_nodes2 = split.empty() ? caller_n_nodes2 : (uint32_t)split.size();
    // ---- premises: finite boxes with lo <= hi below 2^40, nested (leaves hold <= 3 triangles after splitLargeLeaves)
    {
        struct Bad { uint32_t node; const char* why; };
        const unsigned T = threads ? threads : 1;
        std::vector<Bad> bad(T, Bad{0xFFFFFFFFu, ""});  // per chunk: the first node that fails (the lowest index decides the message)
        std::atomic<unsigned> chunk{0};
        par::forRange(n_nodes2, T, 65536, [&](size_t n0, size_t n1) {
            Bad& mine = bad[chunk.fetch_add(1) % T];
            for (size_t n = n0; n < n1 && mine.node == 0xFFFFFFFFu; ++n) {
                Entry ch[2];
                children(nodes2[n], ch);
                const char* why = nullptr;
                for (int k = 0; k < 2 && !why; ++k) {
                    for (int a = 0; a < 3 && !why; ++a) {
                        const float lo = ch[k].b.lo[a], hi = ch[k].b.hi[a];
                        if (!(std::isfinite(lo) && std::isfinite(hi) && lo <= hi)) why = "a box is not finite or has lo > hi";
                        else if (!(std::fabs(lo) < 1.0995116e12f && std::fabs(hi) < 1.0995116e12f)) why = "coordinates of 2^40 or more";
                    }
                    if (why) break;
                    if (ch[k].ref & TRT_LEAF_BIT) {
                        if (TRT_LEAF_COUNT(ch[k].ref) > TRT_OCT_MAX_LEAF_TRIS) why = "a leaf of more than 3 triangles";
                        continue;
                    }
                    if (ch[k].ref >= n_nodes2) { why = "a child index out of range"; break; }  // (a node no path reaches: validateBvh does not see it)
                    Entry g[2];
                    children(nodes2[ch[k].ref], g);
                    if (!contains(ch[k].b, g[0].b) || !contains(ch[k].b, g[1].b)) why = "boxes are not nested";
                }
                if (why) mine = Bad{(uint32_t)n, why};
            }
        });
        const Bad* first = nullptr;
        for (const Bad& x : bad)
            if (x.node != 0xFFFFFFFFu && (!first || x.node < first->node)) first = &x;
        if (first) { t.why = first->why; return t; }
    }
    // ---- dynamic programme (trt_wide.h collapseBvh, eight children); tables uninitialised, first touched by the task that owns them
    std::unique_ptr<double[]> area(new double[n_nodes2]);
    std::unique_ptr<double[]> rootc(new double[n_nodes2]);
    std::unique_ptr<double[]> best(new double[(size_t)n_nodes2 * (W - 1)]);  // best[n * 7 + (k - 1)], k = 1..7: n's subtree as <= k children of a wide node
    std::unique_ptr<uint8_t[]> split_root(new uint8_t[n_nodes2]);
    std::unique_ptr<uint8_t[]> split_k(new uint8_t[(size_t)n_nodes2 * (W - 1)]);  // 0: n stays one child; else i of the (i, k - i) split
    auto bestOf = [&](uint32_t ref, int k) -> double { return (ref & TRT_LEAF_BIT) ? 0.0 : best[(size_t)ref * (W - 1) + (k - 1)]; };
    auto visitDown = [&](uint32_t n) {
        Entry c[2];
        children(nodes2[n], c);
        for (int k = 0; k < 2; ++k)
            if (!(c[k].ref & TRT_LEAF_BIT)) area[c[k].ref] = halfAreaD(c[k].b);
    };
    auto solve = [&](uint32_t n) {
        const uint32_t l = nodes2[n].child0, r = nodes2[n].child1;
        double br = 1.0e300; int bi = 1;
        for (int i = 1; i <= W - 1; ++i) { const double c = bestOf(l, i) + bestOf(r, W - i); if (c < br) { br = c; bi = i; } }
        rootc[n] = area[n] + br;
        split_root[n] = (uint8_t)bi;
        for (int k = 1; k <= W - 1; ++k) {
            double b = rootc[n]; int s = 0;
            for (int i = 1; i < k; ++i) { const double c = bestOf(l, i) + bestOf(r, k - i); if (c < b) { b = c; s = i; } }
            best[(size_t)n * (W - 1) + (k - 1)] = b;
            split_k[(size_t)n * (W - 1) + (k - 1)] = (uint8_t)s;
        }
    };
    {
        Entry c[2];
        children(nodes2[0], c);
        Box rb;
        for (int a = 0; a < 3; ++a) { rb.lo[a] = std::fmin(c[0].b.lo[a], c[1].b.lo[a]); rb.hi[a] = std::fmax(c[0].b.hi[a], c[1].b.hi[a]); }
        area[0] = halfAreaD(rb);
    }
    {
        const TreeCut cut = cutTree(nodes2, threads);
        for (uint32_t n : cut.top) visitDown(n);
        par::forTasks(cut.roots.size(), threads, [&](size_t ti) {
            std::vector<uint32_t> order, st{cut.roots[ti]};
            while (!st.empty()) {
                const uint32_t n = st.back(); st.pop_back();
                order.push_back(n);
                visitDown(n);
                if (!(nodes2[n].child0 & TRT_LEAF_BIT)) st.push_back(nodes2[n].child0);
                if (!(nodes2[n].child1 & TRT_LEAF_BIT)) st.push_back(nodes2[n].child1);
            }
            for (size_t idx = order.size(); idx-- > 0;) solve(order[idx]);
        });
        for (size_t idx = cut.top.size(); idx-- > 0;) solve(cut.top[idx]);
    }
    // ---- emit level by level (breadth first: the upper levels end up together at the front of the array), parents before children, a
    // node's inner children next to each other in slot order, its leaf triangles likewise.  Per level: every node is laid out on its own
    // (phase A), a prefix sum over the level hands out the child and triangle indices, and the nodes are written (phase B).
    struct Job { uint32_t bvh2, oct; };
    struct Draft {
        OctNode on;          // complete but for child_base / tri_base
        uint32_t ref[W];     // per slot: the BVH2 reference of the child in it (TRT_WIDE_EMPTY: none)
        uint32_t n_inner, n_tri;
        uint32_t child_base, tri_base;
        bool ok;
    };
    auto draft = [&](const Job& j, Draft& d) {
        Entry e[W];
        int n = 0;
        {
            struct Item { Entry en; int k; };
            Item stack[2 * W];  // at most W entries wait at a time (every item stands for >= 1 of the <= W children)
            int top = 0;
            Entry c[2];
            children(nodes2[j.bvh2], c);
            const int i = split_root[j.bvh2];
            stack[top++] = {c[1], W - i};
            stack[top++] = {c[0], i};
            while (top > 0) {
                const Item it = stack[--top];
                const uint32_t ref = it.en.ref;
                const int s = (ref & TRT_LEAF_BIT) ? 0 : split_k[(size_t)ref * (W - 1) + (it.k - 1)];
                if (s == 0) {
                    if (!((ref & TRT_LEAF_BIT) && TRT_LEAF_COUNT(ref) == 0)) e[n++] = it.en;  // an empty leaf needs no slot
                    continue;
                }
                Entry g[2];
                children(nodes2[ref], g);
                stack[top++] = {g[1], it.k - s};
                stack[top++] = {g[0], s};
            }
        }
        // frame
        double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        for (int a = 0; a < 3; ++a) {
            for (int k = 0; k < n; ++k) {
                if (k == 0 || e[k].b.lo[a] < lo[a]) lo[a] = e[k].b.lo[a];
                if (k == 0 || e[k].b.hi[a] > hi[a]) hi[a] = e[k].b.hi[a];
            }
        }
        // slots by octant: greedy on the projection of the child's centre on the slot's diagonal
        int slot_of[W], child_in[W];
        for (int k = 0; k < W; ++k) { slot_of[k] = -1; child_in[k] = -1; }
        {
            double cost[W][W];
            for (int k = 0; k < n; ++k) {
                double cc[3];
                for (int a = 0; a < 3; ++a) cc[a] = 0.5 * ((double)e[k].b.lo[a] + e[k].b.hi[a]) - 0.5 * (lo[a] + hi[a]);
                for (int s = 0; s < W; ++s) cost[k][s] = ((s & 4) ? cc[0] : -cc[0]) + ((s & 2) ? cc[1] : -cc[1]) + ((s & 1) ? cc[2] : -cc[2]);
            }
            for (int round = 0; round < n; ++round) {
                int bk = -1, bs = -1;
                for (int k = 0; k < n; ++k) {
                    if (slot_of[k] >= 0) continue;
                    for (int s = 0; s < W; ++s)
                        if (child_in[s] < 0 && (bk < 0 || cost[k][s] > cost[bk][bs])) { bk = k; bs = s; }
                }
                slot_of[bk] = bs;
                child_in[bs] = bk;
            }
        }
        // quantisation
        uint32_t ebits[3];
        uint32_t qlo[3][W], qhi[3][W];
        d.ok = true;
        for (int a = 0; a < 3; ++a) {
            const double p = lo[a], ext = hi[a] - lo[a];
            int eb = 1;
            if (ext > 0.0) {
                int ex = 0;
                (void)std::frexp(ext / 255.0, &ex);  // ext / 255 = m * 2^ex, m in [0.5, 1): 2^ex >= ext / 255
                eb = std::max(ex + 127, 1);
            }
            for (;; ++eb) {
                if (eb > 254) { d.ok = false; return; }
                const double s = std::ldexp(1.0, eb - 127);
                bool fits = true;
                for (int sl = 0; sl < W && fits; ++sl) {
                    const int k = child_in[sl];
                    if (k < 0) { qlo[a][sl] = 255u; qhi[a][sl] = 0u; continue; }
                    long ql = (long)std::floor(((double)e[k].b.lo[a] - p) / s);
                    ql = std::min(std::max(ql, 0L), 255L);
                    while (ql > 0 && p + (double)ql * s > (double)e[k].b.lo[a]) --ql;
                    long qh = (long)std::ceil(((double)e[k].b.hi[a] - p) / s);
                    qh = std::max(qh, 0L);
                    while (qh < 256 && p + (double)qh * s < (double)e[k].b.hi[a]) ++qh;
                    if (qh > 255 || p + (double)ql * s > (double)e[k].b.lo[a]) { fits = false; break; }
                    qlo[a][sl] = (uint32_t)ql;
                    qhi[a][sl] = (uint32_t)qh;
                }
                if (fits) break;
            }
            ebits[a] = (uint32_t)eb;
        }
        // children: inner ones get consecutive node indices in slot order, leaf triangles consecutive records in slot order
        uint32_t imask = 0u, tri_off = 0u, n_inner = 0u;
        uint8_t meta[W];
        for (int sl = 0; sl < W; ++sl) {
            meta[sl] = 0;
            d.ref[sl] = TRT_WIDE_EMPTY;
            const int k = child_in[sl];
            if (k < 0) continue;
            const uint32_t ref = e[k].ref;
            d.ref[sl] = ref;
            if (ref & TRT_LEAF_BIT) {
                const uint32_t count = TRT_LEAF_COUNT(ref);
                meta[sl] = (uint8_t)((((1u << count) - 1u) << 5) | tri_off);
                tri_off += count;
            } else {
                imask |= 1u << sl;
                meta[sl] = (uint8_t)(0x20u | (24u + (uint32_t)sl));
                ++n_inner;
            }
        }
        d.n_inner = n_inner;
        d.n_tri = tri_off;
        auto pack4 = [](const uint32_t* v) { return v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24); };
        auto packm = [](const uint8_t* v) { return (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24); };
        d.on.q[0] = mk4((float)lo[0], (float)lo[1], (float)lo[2], u2f(ebits[0] | (ebits[1] << 8) | (ebits[2] << 16) | (imask << 24)));
        d.on.q[1] = mk4(0.f, 0.f, u2f(packm(meta)), u2f(packm(meta + 4)));
        d.on.q[2] = mk4(u2f(pack4(qlo[0])), u2f(pack4(qlo[0] + 4)), u2f(pack4(qlo[1])), u2f(pack4(qlo[1] + 4)));
        d.on.q[3] = mk4(u2f(pack4(qlo[2])), u2f(pack4(qlo[2] + 4)), u2f(pack4(qhi[0])), u2f(pack4(qhi[0] + 4)));
        d.on.q[4] = mk4(u2f(pack4(qhi[1])), u2f(pack4(qhi[1] + 4)), u2f(pack4(qhi[2])), u2f(pack4(qhi[2] + 4)));
    };
    std::vector<Job> cur{{0u, 0u}}, next;
    std::vector<Draft> drafts;
    t.nodes.reserve(n_nodes2 / 4 + 1);
    t.tri_trav.reserve(n_tris);
    t.nodes.emplace_back();
    while (!cur.empty()) {
        ++t.levels;
        drafts.resize(cur.size());
        par::forRange(cur.size(), threads, 256, [&](size_t j0, size_t j1) {
            for (size_t j = j0; j < j1; ++j) draft(cur[j], drafts[j]);
        });
        size_t n_nodes = t.nodes.size(), n_tt = t.tri_trav.size();
        for (Draft& d : drafts) {
            if (!d.ok) { t.nodes.clear(); t.tri_trav.clear(); t.levels = 0; t.why = "extent not representable"; return t; }
            d.child_base = (uint32_t)n_nodes;
            d.tri_base = (uint32_t)n_tt;
            n_nodes += d.n_inner;
            n_tt += d.n_tri;
        }
        const size_t first_child = t.nodes.size();
        t.nodes.resize(n_nodes);
        t.tri_trav.resize(n_tt);
        next.resize(n_nodes - first_child);
        par::forRange(cur.size(), threads, 256, [&](size_t j0, size_t j1) {
            for (size_t j = j0; j < j1; ++j) {
                Draft& d = drafts[j];
                d.on.q[1].x = u2f(d.child_base);
                d.on.q[1].y = u2f(d.tri_base);
                t.nodes[cur[j].oct] = d.on;
                uint32_t ci = d.child_base, ti = d.tri_base;
                for (int sl = 0; sl < W; ++sl) {
                    const uint32_t ref = d.ref[sl];
                    if (ref == TRT_WIDE_EMPTY) continue;
                    if (ref & TRT_LEAF_BIT) {
                        const uint32_t first = TRT_LEAF_FIRST(ref), count = TRT_LEAF_COUNT(ref);
                        for (uint32_t i = 0; i < count; ++i) {
                            TriIsect T = tri_isect[first + i];
                            const uint32_t li = leaf_info[first + i];  // the caller's leaf, not this slot
                            T.c.w = u2f((first + i) | ((li & 15u) << 27));
                            T.c.z = u2f(f2u(T.c.z) | ((li >> 4) << 1));
                            t.tri_trav[ti++] = T;
                        }
                    } else {
                        next[ci - first_child] = Job{ref, ci};
                        ++ci;
                    }
                }
            }
        });
        cur.swap(next);
    }
    if (t.tri_trav.empty()) t.tri_trav.push_back(TriIsect{mk4(0, 0, 0, 0), mk4(0, 0, 0, 0), mk4(0, 0, 0, 0)});
    t.ok = true;
    return t;
}

// Per light: the union of the boxes of the leaves that hold the triangles of its material (LightBox, trt_oct.h); a light without
// triangles gets an inverted box, which no ray passes.  `lb` = leafBoxesOf().
inline std::vector<LightBox> lightBoxesOf(const std::vector<f4>& lb, const int32_t* tri_mat, uint32_t n_tris, const trt_light* lights, uint32_t n_lights)
{
    std::vector<LightBox> out(n_lights, LightBox{{3.0e38f, 3.0e38f, 3.0e38f}, {-3.0e38f, -3.0e38f, -3.0e38f}});
    if (n_lights == 0) return out;
    for (uint32_t i = 0; i < n_tris; ++i) {
        const int32_t m = tri_mat[i];
        for (uint32_t l = 0; l < n_lights; ++l) {
            if (m != lights[l].mat) continue;
            LightBox& B = out[l];
            const f4 a = lb[2 * (size_t)i], b = lb[2 * (size_t)i + 1];
            B.lo[0] = std::fmin(B.lo[0], a.x); B.lo[1] = std::fmin(B.lo[1], a.y); B.lo[2] = std::fmin(B.lo[2], a.z);
            B.hi[0] = std::fmax(B.hi[0], a.w); B.hi[1] = std::fmax(B.hi[1], b.x); B.hi[2] = std::fmax(B.hi[2], b.y);
        }
    }
    return out;
}

}  // namespace trtd
