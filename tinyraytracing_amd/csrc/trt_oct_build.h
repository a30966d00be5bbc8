// trt_oct_build.h — host side: collapses the caller's BVH2 (include/trt.h) into the 8-wide compressed nodes of trt_oct.h and lays
// the triangles out the way those nodes address them.  Used by trt_create (trt_api.hip) and tests/hostsim, so both walk the same tree.
//
// Collapse: the dynamic programme of trt_wide.h for eight children — which intermediate boxes to drop so that the sum of the
// half-areas of all wide nodes (the expected number of node visits) is minimal.  Dropping boxes cannot change a hit (trt_wide.h);
// this node kind is only built for NESTED trees with finite boxes (every other tree keeps the exact 4-wide nodes).
// Slots: a child's slot index says on which side of the node's centre it lies (bit 2 = +x, bit 1 = +y, bit 0 = +z), assigned
// greedily by the projection of its centre on the slot's diagonal, so that "slot xor (7 - ray octant), highest first" enters the
// children roughly front to back without sorting (Ylitie et al. 2017).  The order is free: only the visited SET matters.
// Quantisation: frame origin p = the per-axis minimum of the children's boxes, scale 2^(e-127) the smallest power of two that fits
// the extent into 0..255; every byte is moved outward until p + q * scale contains the exact bound IN BINARY64 (exact arithmetic:
// a 24-bit and an 8-bit significand), which is the premise of trt_oct.h's no-false-negative argument.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "trt_oct.h"
#include "trt_wide.h"

namespace trtd {

struct OctTree {
    std::vector<OctNode> nodes;     // nodes[0] is the root
    std::vector<TriIsect> tri_trav; // the triangles in node order: c.w = original index | position in leaf << 27 | leaf size << 29
    uint32_t levels = 0;            // nodes on the longest root path: the traversal stack needs levels - 1 entries
    bool ok = false;                // false: this tree keeps the exact 4-wide nodes (not nested / not finite / a leaf of more than 3 triangles / coordinates >= 2^40)
    const char* why = "";
};

inline OctTree buildOct(const trt_bvh_node* nodes2, uint32_t n_nodes2, uint32_t n_tris, const TriIsect* tri_isect)
{
    using namespace wide_detail;
    constexpr int W = 8;
    OctTree t;
    if (n_nodes2 == 0) { t.why = "no nodes"; return t; }
    // ---- premises: finite boxes with lo <= hi below 2^40, nested, leaves of <= 3 triangles
    for (uint32_t n = 0; n < n_nodes2; ++n) {
        Entry ch[2];
        children(nodes2[n], ch);
        for (int k = 0; k < 2; ++k) {
            for (int a = 0; a < 3; ++a) {
                const float lo = ch[k].b.lo[a], hi = ch[k].b.hi[a];
                if (!(std::isfinite(lo) && std::isfinite(hi) && lo <= hi)) { t.why = "a box is not finite or has lo > hi"; return t; }
                if (!(std::fabs(lo) < 1.0995116e12f && std::fabs(hi) < 1.0995116e12f)) { t.why = "coordinates of 2^40 or more"; return t; }
            }
            if (ch[k].ref & TRT_LEAF_BIT) {
                if (TRT_LEAF_COUNT(ch[k].ref) > TRT_OCT_MAX_LEAF_TRIS) { t.why = "a leaf of more than 3 triangles"; return t; }
                continue;
            }
            Entry g[2];
            children(nodes2[ch[k].ref], g);
            if (!contains(ch[k].b, g[0].b) || !contains(ch[k].b, g[1].b)) { t.why = "boxes are not nested"; return t; }
        }
    }
    // ---- dynamic programme (trt_wide.h collapseBvh, eight children)
    std::vector<double> area(n_nodes2, 0.0);
    std::vector<uint32_t> order;
    order.reserve(n_nodes2);
    {
        std::vector<uint32_t> st{0u};
        while (!st.empty()) {
            const uint32_t n = st.back(); st.pop_back();
            order.push_back(n);
            Entry c[2];
            children(nodes2[n], c);
            for (int k = 0; k < 2; ++k)
                if (!(c[k].ref & TRT_LEAF_BIT)) { area[c[k].ref] = halfAreaD(c[k].b); st.push_back(c[k].ref); }
        }
        Entry c[2];
        children(nodes2[0], c);
        Box rb;
        for (int a = 0; a < 3; ++a) { rb.lo[a] = std::fmin(c[0].b.lo[a], c[1].b.lo[a]); rb.hi[a] = std::fmax(c[0].b.hi[a], c[1].b.hi[a]); }
        area[0] = halfAreaD(rb);
    }
    std::vector<double> rootc(n_nodes2, 0.0);
    std::vector<double> best((size_t)n_nodes2 * (W - 1), 0.0);  // best[n * 7 + (k - 1)], k = 1..7: n's subtree as <= k children of a wide node
    std::vector<uint8_t> split_root(n_nodes2, 1);
    std::vector<uint8_t> split_k((size_t)n_nodes2 * (W - 1), 0);  // 0: n stays one child; else i of the (i, k - i) split
    auto bestOf = [&](uint32_t ref, int k) -> double { return (ref & TRT_LEAF_BIT) ? 0.0 : best[(size_t)ref * (W - 1) + (k - 1)]; };
    for (size_t idx = order.size(); idx-- > 0;) {
        const uint32_t n = order[idx];
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
    }
    // ---- emit, parents before children, a node's inner children next to each other
    struct Job { uint32_t bvh2, oct, level; };
    std::vector<Job> jobs;
    t.nodes.reserve(n_nodes2 / 4 + 1);
    t.tri_trav.reserve(n_tris);
    t.nodes.emplace_back();
    jobs.push_back({0u, 0u, 1u});
    size_t head = 0;
    while (head < jobs.size()) {  // FIFO: breadth first (the upper levels end up together at the front of the array)
        const Job j = jobs[head++];
        t.levels = std::max(t.levels, j.level);
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
        for (int a = 0; a < 3; ++a) {
            const double p = lo[a], ext = hi[a] - lo[a];
            int eb = 1;
            if (ext > 0.0) {
                int ex = 0;
                (void)std::frexp(ext / 255.0, &ex);  // ext / 255 = m * 2^ex, m in [0.5, 1): 2^ex >= ext / 255
                eb = std::max(ex + 127, 1);
            }
            for (;; ++eb) {
                if (eb > 254) { t.nodes.clear(); t.tri_trav.clear(); t.why = "extent not representable"; return t; }
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
        const uint32_t child_base = (uint32_t)t.nodes.size(), tri_base = (uint32_t)t.tri_trav.size();
        uint32_t imask = 0u, tri_off = 0u;
        uint8_t meta[W];
        for (int sl = 0; sl < W; ++sl) {
            meta[sl] = 0;
            const int k = child_in[sl];
            if (k < 0) continue;
            const uint32_t ref = e[k].ref;
            if (ref & TRT_LEAF_BIT) {
                const uint32_t first = TRT_LEAF_FIRST(ref), count = TRT_LEAF_COUNT(ref);
                meta[sl] = (uint8_t)((((1u << count) - 1u) << 5) | tri_off);
                for (uint32_t i = 0; i < count; ++i) {
                    TriIsect T = tri_isect[first + i];
                    T.c.w = u2f((first + i) | (i << 27) | (count << 29));
                    t.tri_trav.push_back(T);
                }
                tri_off += count;
            } else {
                imask |= 1u << sl;
                meta[sl] = (uint8_t)(0x20u | (24u + (uint32_t)sl));
                const uint32_t idx = (uint32_t)t.nodes.size();
                t.nodes.emplace_back();
                jobs.push_back({ref, idx, j.level + 1});
            }
        }
        auto pack4 = [](const uint32_t* v) { return v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24); };
        auto packm = [](const uint8_t* v) { return (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24); };
        OctNode on;
        on.q[0] = mk4((float)lo[0], (float)lo[1], (float)lo[2], u2f(ebits[0] | (ebits[1] << 8) | (ebits[2] << 16) | (imask << 24)));
        on.q[1] = mk4(u2f(child_base), u2f(tri_base), u2f(packm(meta)), u2f(packm(meta + 4)));
        on.q[2] = mk4(u2f(pack4(qlo[0])), u2f(pack4(qlo[0] + 4)), u2f(pack4(qlo[1])), u2f(pack4(qlo[1] + 4)));
        on.q[3] = mk4(u2f(pack4(qlo[2])), u2f(pack4(qlo[2] + 4)), u2f(pack4(qhi[0])), u2f(pack4(qhi[0] + 4)));
        on.q[4] = mk4(u2f(pack4(qhi[1])), u2f(pack4(qhi[1] + 4)), u2f(pack4(qhi[2])), u2f(pack4(qhi[2] + 4)));
        t.nodes[j.oct] = on;
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
    for (uint32_t l = 0; l < n_lights; ++l) {
        LightBox& B = out[l];
        for (uint32_t i = 0; i < n_tris; ++i) {
            if (tri_mat[i] != lights[l].mat) continue;
            const f4 a = lb[2 * (size_t)i], b = lb[2 * (size_t)i + 1];
            B.lo[0] = std::fmin(B.lo[0], a.x); B.lo[1] = std::fmin(B.lo[1], a.y); B.lo[2] = std::fmin(B.lo[2], a.z);
            B.hi[0] = std::fmax(B.hi[0], a.w); B.hi[1] = std::fmax(B.hi[1], b.x); B.hi[2] = std::fmax(B.hi[2], b.y);
        }
    }
    return out;
}

}  // namespace trtd
