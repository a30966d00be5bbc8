// trt_wide.h — host side: collapses the caller's BVH2 (include/trt.h, trt_bvh_node) into the 4-wide
// nodes the per-lane traversal walks (WideNode, trt_path.h).  Used by trt_create (trt_api.hip) and by
// tests/hostsim, so both sides walk the very same tree.
//
// Why the result of a ray cannot change: a wide node keeps the leaves and the leaf order of the binary
// tree and only drops some intermediate boxes.  The reference descends into a child iff the ray passes
// the child's box test (bvh.cpp:156-166), so a leaf is reached iff the ray passes every box on its
// root path.  The slab test (boxTest, trt_path.h) is monotone in the box: when box P contains box C,
// every per-axis bound of C lies inside P's ((x - o) * inv is monotone in x under rounding, fminf/fmaxf
// drop NaNs the same way for both), so "passes C" implies "passes P" and entry(P) <= entry(C).  Dropping
// P's test therefore never changes which leaves are reached, nor what is culled by the best hit.  An
// intermediate node whose box does NOT contain both of its children's boxes (possible only in a tree
// handed over by a foreign builder) is kept as a node of its own and never dropped.  Among the collapses that
// respect this, the one with the fewest expected wide-node visits is taken (dynamic programme below).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <limits>
#include <vector>

#include "trt_path.h"

namespace trtd {

struct WideTree {
    std::vector<WideNode> nodes;  // nodes[0] is the root
    uint32_t stack_need = 0;      // upper bound of the traversal stack: max over root paths of sum(children - 1)
    uint64_t dropped = 0;         // intermediate boxes dropped
};

namespace wide_detail {
struct Box { float lo[3], hi[3]; };
struct Entry { Box b; uint32_t ref; };  // ref: BVH2 child reference (leaf bit or BVH2 node index)

inline float halfArea(const Box& b)
{
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}
inline double halfAreaD(const Box& b)
{
    const double dx = (double)b.hi[0] - b.lo[0], dy = (double)b.hi[1] - b.lo[1], dz = (double)b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}
inline bool contains(const Box& p, const Box& c)
{
    for (int a = 0; a < 3; ++a)
        if (!(p.lo[a] <= c.lo[a] && p.hi[a] >= c.hi[a])) return false;
    return true;
}
inline void children(const trt_bvh_node& n, Entry out[2])
{
    for (int a = 0; a < 3; ++a) {
        out[0].b.lo[a] = n.lo0[a]; out[0].b.hi[a] = n.hi0[a];
        out[1].b.lo[a] = n.lo1[a]; out[1].b.hi[a] = n.hi1[a];
    }
    out[0].ref = n.child0;
    out[1].ref = n.child1;
}
}  // namespace wide_detail

// `nodes` must have passed validateBvh (every inner node reachable exactly once, indices in range).
// Which intermediate nodes to drop is chosen by dynamic programming over the binary tree so that the expected
// number of wide-node visits (sum of the half-areas of the boxes of all wide nodes) is minimal:
//   root(n)    = area(n) + min_{i=1..3} best(left, i) + best(right, 4 - i)      n becomes a wide node
//   best(n, k) = min(root(n), min_{i<k} best(left, i) + best(right, k - i))     n's subtree as <= k children of a wide node
//   best(leaf, k) = 0
inline WideTree collapseBvh(const trt_bvh_node* nodes, uint32_t n_nodes)
{
    using namespace wide_detail;
    WideTree w;
    if (n_nodes == 0) return w;
    // boxes of the inner nodes (union of the two stored child boxes) and whether a node may be opened
    std::vector<double> area(n_nodes, 0.0);  // double: the sums below span leaf boxes to the scene box
    std::vector<uint8_t> openable(n_nodes, 1);   // as a child: its stored box contains its children's boxes
    std::vector<uint32_t> order;                 // pre-order
    order.reserve(n_nodes);
    {
        std::vector<uint32_t> st{0u};
        while (!st.empty()) {
            const uint32_t n = st.back(); st.pop_back();
            order.push_back(n);
            Entry c[2];
            children(nodes[n], c);
            for (int k = 0; k < 2; ++k) {
                if (c[k].ref & TRT_LEAF_BIT) continue;
                Entry g[2];
                children(nodes[c[k].ref], g);
                area[c[k].ref] = halfAreaD(c[k].b);
                openable[c[k].ref] = contains(c[k].b, g[0].b) && contains(c[k].b, g[1].b);
                st.push_back(c[k].ref);
            }
        }
        Entry c[2];
        children(nodes[0], c);
        Box rb;
        for (int a = 0; a < 3; ++a) { rb.lo[a] = std::fmin(c[0].b.lo[a], c[1].b.lo[a]); rb.hi[a] = std::fmax(c[0].b.hi[a], c[1].b.hi[a]); }
        area[0] = halfAreaD(rb);
    }
    // bottom-up
    std::vector<double> rootc(n_nodes, 0.0);
    std::vector<double> best((size_t)n_nodes * 3, 0.0);  // best[n*3 + (k-1)], k = 1..3
    std::vector<uint8_t> split_root(n_nodes, 1);         // i of the best (i, 4-i) split when n is a wide node
    std::vector<uint8_t> split_k((size_t)n_nodes * 3, 0);  // 0: keep n as one child; else i of the (i, k-i) split
    auto bestOf = [&](uint32_t ref, int k) -> double { return (ref & TRT_LEAF_BIT) ? 0.0 : best[(size_t)ref * 3 + (k - 1)]; };
    for (size_t idx = order.size(); idx-- > 0;) {
        const uint32_t n = order[idx];
        const uint32_t l = nodes[n].child0, r = nodes[n].child1;
        double br = 1.0e300; int bi = 1;
        for (int i = 1; i <= 3; ++i) { const double c = bestOf(l, i) + bestOf(r, 4 - i); if (c < br) { br = c; bi = i; } }
        rootc[n] = area[n] + br;
        split_root[n] = (uint8_t)bi;
        for (int k = 1; k <= 3; ++k) {
            double b = rootc[n]; int s = 0;
            if (openable[n])
                for (int i = 1; i < k; ++i) { const double c = bestOf(l, i) + bestOf(r, k - i); if (c < b) { b = c; s = i; } }
            best[(size_t)n * 3 + (k - 1)] = b;
            split_k[(size_t)n * 3 + (k - 1)] = (uint8_t)s;
        }
    }
    // top-down: emit the wide nodes
    struct Job { uint32_t bvh2, wide; uint32_t need; };
    std::vector<Job> jobs;
    w.nodes.reserve(n_nodes / 2 + 1);
    w.nodes.emplace_back();
    jobs.push_back({0u, 0u, 0u});
    const float qnan = std::numeric_limits<float>::quiet_NaN();
    while (!jobs.empty()) {
        const Job j = jobs.back();
        jobs.pop_back();
        Entry e[TRT_WIDE];
        int n = 0;
        // expand (ref, box, k): the subtree as at most k children, left to right
        struct Item { Entry en; int k; };
        std::vector<Item> stack;
        {
            Entry c[2];
            children(nodes[j.bvh2], c);
            const int i = split_root[j.bvh2];
            stack.push_back({c[1], 4 - i});
            stack.push_back({c[0], i});
        }
        while (!stack.empty()) {
            const Item it = stack.back(); stack.pop_back();
            const uint32_t ref = it.en.ref;
            const int s = (ref & TRT_LEAF_BIT) ? 0 : split_k[(size_t)ref * 3 + (it.k - 1)];
            if (s == 0) { e[n++] = it.en; continue; }
            Entry c[2];
            children(nodes[ref], c);
            stack.push_back({c[1], it.k - s});
            stack.push_back({c[0], s});
            ++w.dropped;
        }
        const uint32_t need = j.need + (uint32_t)(n - 1);
        if (need > w.stack_need) w.stack_need = need;
        WideNode wn;
        for (int k = 0; k < TRT_WIDE; ++k) {
            float* q = reinterpret_cast<float*>(wn.q);
            uint32_t ref = TRT_WIDE_EMPTY;
            if (k < n) {
                for (int a = 0; a < 3; ++a) { q[a * 4 + k] = e[k].b.lo[a]; q[(3 + a) * 4 + k] = e[k].b.hi[a]; }
                if (e[k].ref & TRT_LEAF_BIT) {
                    ref = e[k].ref;
                } else {
                    ref = (uint32_t)w.nodes.size();
                    w.nodes.emplace_back();
                    jobs.push_back({e[k].ref, ref, need});
                }
            } else {
                for (int a = 0; a < 6; ++a) q[a * 4 + k] = qnan;  // an all-NaN box fails every slab test
            }
            uint32_t* qu = reinterpret_cast<uint32_t*>(wn.q);  // integer view: child references are not floats
            qu[6 * 4 + k] = ref;
            qu[7 * 4 + k] = 0u;
        }
        w.nodes[j.wide] = wn;
    }
    return w;
}

// The simple collapse (open the child with the largest box until the node is full).  trt_create uses it above 4 M
// triangles, where it measured better (blob-10M at 4K: 10.95 against 11.34 visits per ray, +3 % rays/s; the area model
// of the dynamic programme fits a finely tessellated closed surface, whose rays start on it, less well), and
// TRT_WIDE_GREEDY=0/1 forces either one for A/B runs.
inline WideTree collapseBvhGreedy(const trt_bvh_node* nodes, uint32_t n_nodes)
{
    using namespace wide_detail;
    WideTree w;
    if (n_nodes == 0) return w;
    w.nodes.reserve(n_nodes / 2 + 1);
    struct Job { uint32_t bvh2, wide; };
    std::vector<Job> jobs;
    std::vector<uint32_t> parent_need;  // per wide node: sum(children-1) over the path from the root to it, inclusive
    w.nodes.emplace_back();
    parent_need.push_back(0);
    jobs.push_back({0u, 0u});
    const float qnan = std::numeric_limits<float>::quiet_NaN();
    while (!jobs.empty()) {
        const Job j = jobs.back();
        jobs.pop_back();
        Entry e[TRT_WIDE];
        int n = 2;
        children(nodes[j.bvh2], e);
        while (n < TRT_WIDE) {
            // open the inner child with the largest box whose own box contains both of its children's
            int pick = -1;
            float best = -1.0f;
            for (int k = 0; k < n; ++k) {
                if (e[k].ref & TRT_LEAF_BIT) continue;
                Entry c[2];
                children(nodes[e[k].ref], c);
                if (!contains(e[k].b, c[0].b) || !contains(e[k].b, c[1].b)) continue;
                const float a = halfArea(e[k].b);
                if (a > best || pick < 0) { best = a; pick = k; }
            }
            if (pick < 0) break;
            Entry c[2];
            children(nodes[e[pick].ref], c);
            for (int k = n; k > pick + 1; --k) e[k] = e[k - 1];  // keep the left-to-right (leaf index) order
            e[pick] = c[0];
            e[pick + 1] = c[1];
            ++n;
            ++w.dropped;
        }
        const uint32_t need = parent_need[j.wide] + (uint32_t)(n - 1);
        if (need > w.stack_need) w.stack_need = need;
        WideNode wn;
        for (int k = 0; k < TRT_WIDE; ++k) {
            float* q = reinterpret_cast<float*>(wn.q);
            uint32_t ref = TRT_WIDE_EMPTY;
            if (k < n) {
                for (int a = 0; a < 3; ++a) { q[a * 4 + k] = e[k].b.lo[a]; q[(3 + a) * 4 + k] = e[k].b.hi[a]; }
                if (e[k].ref & TRT_LEAF_BIT) {
                    ref = e[k].ref;
                } else {
                    ref = (uint32_t)w.nodes.size();
                    w.nodes.emplace_back();
                    parent_need.push_back(need);
                    jobs.push_back({e[k].ref, ref});
                }
            } else {
                for (int a = 0; a < 6; ++a) q[a * 4 + k] = qnan;  // an all-NaN box fails every slab test
            }
            uint32_t* qu = reinterpret_cast<uint32_t*>(wn.q);  // integer view: child references are not floats
            qu[6 * 4 + k] = ref;
            qu[7 * 4 + k] = 0u;
        }
        w.nodes[j.wide] = wn;
    }
    return w;
}

// For every triangle i the caller's box of the leaf it lies in ([2 i] = (lo.xyz, hi.x), [2 i + 1] = (hi.y, hi.z, 0, 0)):
// leafEntry() of trt_path.h tests a triangle hit against the entry distance of the box of its own leaf.
inline std::vector<f4> leafBoxesOf(const trt_bvh_node* nodes2, uint32_t n_nodes2, uint32_t n_tris)
{
    std::vector<f4> lb((size_t)std::max<uint32_t>(n_tris, 1u) * 2, mk4(0.f, 0.f, 0.f, 0.f));
    for (uint32_t n = 0; n < n_nodes2; ++n) {
        const trt_bvh_node& nd = nodes2[n];
        const uint32_t ref[2] = {nd.child0, nd.child1};
        const float* lo[2] = {nd.lo0, nd.lo1};
        const float* hi[2] = {nd.hi0, nd.hi1};
        for (int k = 0; k < 2; ++k) {
            if (!(ref[k] & TRT_LEAF_BIT) || TRT_LEAF_COUNT(ref[k]) == 0) continue;
            const size_t first = TRT_LEAF_FIRST(ref[k]), count = TRT_LEAF_COUNT(ref[k]);
            for (size_t i = first; i < first + count && i < n_tris; ++i) {
                lb[2 * i] = mk4(lo[k][0], lo[k][1], lo[k][2], hi[k][0]);
                lb[2 * i + 1] = mk4(hi[k][1], hi[k][2], 0.f, 0.f);
            }
        }
    }
    return lb;
}

}  // namespace trtd
