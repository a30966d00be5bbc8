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
#include <atomic>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

#include "trt_path.h"

namespace trtd {

struct WideTree {
    std::vector<WideNode> nodes;  // nodes[0] is the root
    uint32_t stack_need = 0;      // upper bound of the traversal stack: max over root paths of sum(children - 1)
    uint64_t dropped = 0;         // intermediate boxes dropped
};


// ---- host threads for the collapses (10 M triangles: seconds on one core).  `threads` is passed in by the caller (trt_create reads
// TRT_HOST_THREADS, default min(16, hardware threads)); every result is independent of it — the tests build with 1 and with several.
namespace par {
inline unsigned defaultThreads()
{
    unsigned h = std::thread::hardware_concurrency();
    if (h == 0) h = 1;
    return h < 16u ? h : 16u;
}
// f(begin, end) over [0, n) in contiguous chunks, one per thread
template <class F>
inline void forRange(size_t n, unsigned threads, size_t grain, F f)
{
    size_t T = threads ? threads : 1;
    if (grain && n / grain < T) T = n / grain;
    if (T <= 1) { if (n) f((size_t)0, n); return; }
    std::vector<std::thread> th;
    th.reserve(T - 1);
    const size_t per = (n + T - 1) / T;
    for (size_t t = 1; t < T; ++t) {
        const size_t b = t * per, e = std::min(n, b + per);
        if (b < e) th.emplace_back([=] { f(b, e); });
    }
    f((size_t)0, std::min(n, per));
    for (std::thread& x : th) x.join();
}
// f(task) for task in [0, n), handed out one at a time (tasks of very different sizes)
template <class F>
inline void forTasks(size_t n, unsigned threads, F f)
{
    size_t T = threads ? threads : 1;
    if (T > n) T = n;
    if (T <= 1) { for (size_t i = 0; i < n; ++i) f(i); return; }
    std::atomic<size_t> next{0};
    auto body = [&] { for (size_t i; (i = next.fetch_add(1, std::memory_order_relaxed)) < n;) f(i); };
    std::vector<std::thread> th;
    th.reserve(T - 1);
    for (size_t t = 1; t < T; ++t) th.emplace_back(body);
    body();
    for (std::thread& x : th) x.join();
}
}  // namespace par

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
// Containment as the slab test sees the boxes: interactAABB (bvh.cpp:231-245) takes min / max of the two planes of an axis, so a box stored with lo > hi
// IS the box [hi, lo] to every ray; compared raw, an inverted parent could "contain" a child that reaches outside it.
inline bool contains(const Box& p, const Box& c)
{
    for (int a = 0; a < 3; ++a)
        if (!(std::fmin(p.lo[a], p.hi[a]) <= std::fmin(c.lo[a], c.hi[a]) && std::fmax(p.lo[a], p.hi[a]) >= std::fmax(c.lo[a], c.hi[a]))) return false;
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
// Do the boxes of the tree nest — does every stored box of an inner child contain the two boxes stored in that child?  The premise of culling by distance
// (trt_path.h, "the rule is what makes culling exact": floor(entry of a leaf's box) >= floor(entry of every box above it)).  A foreign tree may break it
// (include/trt.h accepts any boxes); such a tree is walked WITHOUT distance culling (SceneDev::cull_alpha = +inf), i.e. exactly as bvh.cpp:146-175 walks it.
// Nodes no path reaches are looked at too: a "no" because of one of them costs speed, never correctness.
inline bool boxesNested(const trt_bvh_node* nodes, uint32_t n_nodes, unsigned threads = 1);

// The binary tree cut for parallel work: `top` = the inner nodes above the cut, parents before children; `roots` = the inner nodes
// on the cut, each the root of a subtree one task walks on its own.  One thread: the cut is the root itself.
struct TreeCut {
    std::vector<uint32_t> top, roots;
};
inline TreeCut cutTree(const trt_bvh_node* nodes, unsigned threads)
{
    TreeCut c;
    std::vector<uint32_t> q{0u};
    size_t head = 0;
    const size_t want = threads > 1 ? (size_t)threads * 16 : 1;
    while (head < q.size() && q.size() - head < want) {
        const uint32_t n = q[head++];
        c.top.push_back(n);
        if (!(nodes[n].child0 & TRT_LEAF_BIT)) q.push_back(nodes[n].child0);
        if (!(nodes[n].child1 & TRT_LEAF_BIT)) q.push_back(nodes[n].child1);
    }
    c.roots.assign(q.begin() + (long)head, q.end());
    return c;
}

// Emission of the 4-wide nodes, parents first, in the order of a depth-first walk with a LIFO of pending nodes (a node's inner
// children get consecutive indices when the node is written; all descendants of a child follow before those of the child to its
// left).  `expand(bvh2 node, e[TRT_WIDE], dropped)` returns the children of the wide node rooted there, left to right.
// Parallel form with the same layout: the wide nodes down to a fixed wide depth are expanded first (few), the subtree below every node
// of that depth is emitted by a task into a block of its own with block-relative indices, and the blocks are then placed where
// the sequential walk would have put them (the descendants of a node form one contiguous run in that order).
template <class Expand>
inline void emitWide(const trt_bvh_node* nodes, unsigned threads, Expand expand, WideTree& w)
{
    const float qnan = std::numeric_limits<float>::quiet_NaN();
    struct Job { uint32_t bvh2, wide, need; };
    auto writeNode = [&](WideNode& wn, const Entry* e, int n, const uint32_t* refs) {
        for (int k = 0; k < TRT_WIDE; ++k) {
            float* q = reinterpret_cast<float*>(wn.q);
            if (k < n) {
                for (int a = 0; a < 3; ++a) { q[a * 4 + k] = e[k].b.lo[a]; q[(3 + a) * 4 + k] = e[k].b.hi[a]; }
            } else {
                for (int a = 0; a < 6; ++a) q[a * 4 + k] = qnan;  // an all-NaN box fails every slab test
            }
            uint32_t* qu = reinterpret_cast<uint32_t*>(wn.q);  // integer view: child references are not floats
            qu[6 * 4 + k] = k < n ? refs[k] : TRT_WIDE_EMPTY;
            qu[7 * 4 + k] = 0u;
        }
    };
    // the walk of one subtree: nodes appended to `out` (out[first] = the subtree's root, already allocated by the caller)
    auto walk = [&](std::vector<WideNode>& out, Job root, uint32_t& stack_need, uint64_t& dropped) {
        std::vector<Job> jobs{root};
        while (!jobs.empty()) {
            const Job j = jobs.back();
            jobs.pop_back();
            Entry e[TRT_WIDE];
            const int n = expand(j.bvh2, e, dropped);
            const uint32_t need = j.need + (uint32_t)(n - 1);
            if (need > stack_need) stack_need = need;
            uint32_t refs[TRT_WIDE];
            for (int k = 0; k < n; ++k) {
                if (e[k].ref & TRT_LEAF_BIT) { refs[k] = e[k].ref; continue; }
                refs[k] = (uint32_t)out.size();
                out.emplace_back();
                jobs.push_back({e[k].ref, refs[k], need});
            }
            writeNode(out[j.wide], e, n, refs);
        }
    };
    (void)nodes;
    if (threads <= 1) {
        w.nodes.emplace_back();
        walk(w.nodes, {0u, 0u, 0u}, w.stack_need, w.dropped);
        return;
    }
    // ---- the top: wide nodes of depth < CUT, expanded once and kept
    constexpr uint32_t CUT = 4;  // up to 4^4 = 256 tasks
    struct TopNode { uint32_t bvh2, need, depth; Entry e[TRT_WIDE]; int n; int32_t kid[TRT_WIDE]; int32_t task; };
    std::vector<TopNode> top;
    struct Task { uint32_t bvh2, need; std::vector<WideNode> block; uint32_t stack_need = 0; uint64_t dropped = 0; uint32_t base = 0; };
    std::vector<Task> tasks;
    {
        top.push_back({0u, 0u, 0u, {}, 0, {-1, -1, -1, -1}, -1});
        for (size_t i = 0; i < top.size(); ++i) {
            if (top[i].depth >= CUT) {  // a task's root: its node is written by the task
                top[i].task = (int32_t)tasks.size();
                tasks.emplace_back();
                tasks.back().bvh2 = top[i].bvh2;
                tasks.back().need = top[i].need;
                continue;
            }
            Entry e[TRT_WIDE];
            const int n = expand(top[i].bvh2, e, w.dropped);
            const uint32_t need = top[i].need + (uint32_t)(n - 1);
            if (need > w.stack_need) w.stack_need = need;
            top[i].n = n;
            for (int k = 0; k < n; ++k) {
                top[i].e[k] = e[k];
                if (e[k].ref & TRT_LEAF_BIT) continue;
                top[i].kid[k] = (int32_t)top.size();
                top.push_back({e[k].ref, need, top[i].depth + 1, {}, 0, {-1, -1, -1, -1}, -1});  // (invalidates no index: kids are addressed by index)
            }
        }
    }
    // ---- the blocks: block[0] = the task's root
    par::forTasks(tasks.size(), threads, [&](size_t ti) {
        Task& t = tasks[ti];
        t.block.emplace_back();
        walk(t.block, {t.bvh2, 0u, t.need}, t.stack_need, t.dropped);
    });
    // ---- placement: replay of the sequential walk over the top (a task's root is one node there, its descendants one run)
    struct Pending { int32_t top_index; uint32_t wide; };
    std::vector<uint32_t> top_wide(top.size(), 0u);
    size_t total = 1;
    {
        std::vector<Pending> jobs{{0, 0u}};
        while (!jobs.empty()) {
            const Pending j = jobs.back();
            jobs.pop_back();
            const TopNode& tn = top[(size_t)j.top_index];
            top_wide[(size_t)j.top_index] = j.wide;
            if (tn.task >= 0) {  // descendants of the task's root: block[1..]
                tasks[(size_t)tn.task].base = (uint32_t)total;
                total += tasks[(size_t)tn.task].block.size() - 1;
                continue;
            }
            for (int k = 0; k < tn.n; ++k)
                if (tn.kid[k] >= 0) jobs.push_back({tn.kid[k], (uint32_t)total++});
        }
    }
    w.nodes.resize(total);
    for (size_t i = 0; i < top.size(); ++i) {
        const TopNode& tn = top[i];
        if (tn.task >= 0) continue;
        uint32_t refs[TRT_WIDE];
        for (int k = 0; k < tn.n; ++k) refs[k] = tn.kid[k] >= 0 ? top_wide[(size_t)tn.kid[k]] : tn.e[k].ref;
        writeNode(w.nodes[top_wide[i]], tn.e, tn.n, refs);
    }
    par::forTasks(tasks.size(), threads, [&](size_t ti) {
        Task& t = tasks[ti];
        uint32_t root_wide = 0;
        for (size_t i = 0; i < top.size(); ++i)
            if (top[i].task == (int32_t)ti) root_wide = top_wide[i];
        // block-relative index r > 0 -> base + r - 1
        for (size_t r = 0; r < t.block.size(); ++r) {
            WideNode wn = t.block[r];
            uint32_t* qu = reinterpret_cast<uint32_t*>(wn.q);
            for (int k = 0; k < TRT_WIDE; ++k) {
                const uint32_t ref = qu[6 * 4 + k];
                if (ref != TRT_WIDE_EMPTY && !(ref & TRT_LEAF_BIT)) qu[6 * 4 + k] = t.base + ref - 1u;
            }
            w.nodes[r == 0 ? root_wide : t.base + r - 1u] = wn;
        }
        t.block = std::vector<WideNode>();
    });
    for (const Task& t : tasks) {
        if (t.stack_need > w.stack_need) w.stack_need = t.stack_need;
        w.dropped += t.dropped;
    }
}
inline bool boxesNested(const trt_bvh_node* nodes, uint32_t n_nodes, unsigned threads)
{
    std::atomic<bool> ok{true};
    par::forRange(n_nodes, threads ? threads : 1, 65536, [&](size_t n0, size_t n1) {
        for (size_t n = n0; n < n1 && ok.load(std::memory_order_relaxed); ++n) {
            Entry c[2];
            children(nodes[n], c);
            for (int k = 0; k < 2; ++k) {
                if ((c[k].ref & TRT_LEAF_BIT) || c[k].ref >= n_nodes) continue;
                Entry g[2];
                children(nodes[c[k].ref], g);
                if (!contains(c[k].b, g[0].b) || !contains(c[k].b, g[1].b)) { ok.store(false, std::memory_order_relaxed); break; }
            }
        }
    });
    return ok.load();
}

// The filter of planeMaybe() (trt_path.h): one bit per (axis, coordinate) of every stored box plane, 4-8 bits of table per entry (a false "maybe" costs one slow
// walk of a ray that is rare to begin with; a false "no" cannot happen).  Returns log2 of the number of bits.
inline uint32_t planeFilterBuild(const trt_bvh_node* nodes, uint32_t n_nodes, std::vector<uint32_t>& bits, unsigned threads = 1)
{
    uint32_t lg = 10;
    while (lg < 32 && (1ull << lg) < 48ull * n_nodes) ++lg;  // 12 entries per node, >= 4 bits each
    bits.assign((size_t)((1ull << lg) / 32), 0u);
    const uint32_t shift = 32u - lg;
    std::atomic<uint32_t>* words = reinterpret_cast<std::atomic<uint32_t>*>(bits.data());
    par::forRange(n_nodes, threads ? threads : 1, 65536, [&](size_t n0, size_t n1) {
        for (size_t n = n0; n < n1; ++n) {
            const trt_bvh_node& nd = nodes[n];
            for (int a = 0; a < 3; ++a)
                for (float x : {nd.lo0[a], nd.hi0[a], nd.lo1[a], nd.hi1[a]}) {
                    const uint32_t h = (planeKey(a, x) * 2246822519u) >> shift;
                    words[h >> 5].fetch_or(1u << (h & 31u), std::memory_order_relaxed);
                }
        }
    });
    return lg;
}
}  // namespace wide_detail

// `nodes` must have passed validateBvh (every inner node reachable exactly once, indices in range).
// Which intermediate nodes to drop is chosen by dynamic programming over the binary tree so that the expected
// number of wide-node visits (sum of the half-areas of the boxes of all wide nodes) is minimal:
//   root(n)    = area(n) + min_{i=1..3} best(left, i) + best(right, 4 - i)      n becomes a wide node
//   best(n, k) = min(root(n), min_{i<k} best(left, i) + best(right, k - i))     n's subtree as <= k children of a wide node
//   best(leaf, k) = 0
inline WideTree collapseBvh(const trt_bvh_node* nodes, uint32_t n_nodes, unsigned threads = 1)
{
    using namespace wide_detail;
    WideTree w;
    if (n_nodes == 0) return w;
    // Tables of the dynamic programme; every entry of a reachable node is written before it is read (children before parents),
    // so nothing is initialised here: the pages are first touched by the tasks that own them.
    std::unique_ptr<double[]> area(new double[n_nodes]);   // box of the inner node (double: the sums span leaf boxes to the scene box)
    std::unique_ptr<uint8_t[]> openable(new uint8_t[n_nodes]);  // as a child: its stored box contains its children's boxes
    std::unique_ptr<double[]> rootc(new double[n_nodes]);
    std::unique_ptr<double[]> best(new double[(size_t)n_nodes * 3]);       // best[n*3 + (k-1)], k = 1..3
    std::unique_ptr<uint8_t[]> split_root(new uint8_t[n_nodes]);            // i of the best (i, 4-i) split when n is a wide node
    std::unique_ptr<uint8_t[]> split_k(new uint8_t[(size_t)n_nodes * 3]);   // 0: keep n as one child; else i of the (i, k-i) split
    auto bestOf = [&](uint32_t ref, int k) -> double { return (ref & TRT_LEAF_BIT) ? 0.0 : best[(size_t)ref * 3 + (k - 1)]; };
    auto visitDown = [&](uint32_t n) {  // what a node tells about its inner children
        Entry c[2];
        children(nodes[n], c);
        for (int k = 0; k < 2; ++k) {
            if (c[k].ref & TRT_LEAF_BIT) continue;
            Entry g[2];
            children(nodes[c[k].ref], g);
            area[c[k].ref] = halfAreaD(c[k].b);
            openable[c[k].ref] = contains(c[k].b, g[0].b) && contains(c[k].b, g[1].b);
        }
    };
    auto solve = [&](uint32_t n) {
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
    };
    {
        Entry c[2];
        children(nodes[0], c);
        Box rb;
        for (int a = 0; a < 3; ++a) { rb.lo[a] = std::fmin(c[0].b.lo[a], c[1].b.lo[a]); rb.hi[a] = std::fmax(c[0].b.hi[a], c[1].b.hi[a]); }
        area[0] = halfAreaD(rb);
        openable[0] = 1;
    }
    const TreeCut cut = cutTree(nodes, threads);
    for (uint32_t n : cut.top) visitDown(n);
    par::forTasks(cut.roots.size(), threads, [&](size_t ti) {  // bottom-up inside every subtree of the cut
        std::vector<uint32_t> order, st{cut.roots[ti]};
        while (!st.empty()) {
            const uint32_t n = st.back(); st.pop_back();
            order.push_back(n);
            visitDown(n);
            if (!(nodes[n].child0 & TRT_LEAF_BIT)) st.push_back(nodes[n].child0);
            if (!(nodes[n].child1 & TRT_LEAF_BIT)) st.push_back(nodes[n].child1);
        }
        for (size_t idx = order.size(); idx-- > 0;) solve(order[idx]);
    });
    for (size_t idx = cut.top.size(); idx-- > 0;) solve(cut.top[idx]);
    // top-down: emit the wide nodes
    auto expand = [&](uint32_t bvh2, Entry* e, uint64_t& dropped) -> int {
        int n = 0;
        struct Item { Entry en; int k; };
        Item stack[2 * TRT_WIDE];  // at most TRT_WIDE entries wait at a time (every item stands for >= 1 of the <= TRT_WIDE children)
        int top = 0;
        Entry c[2];
        children(nodes[bvh2], c);
        const int i = split_root[bvh2];
        stack[top++] = {c[1], 4 - i};
        stack[top++] = {c[0], i};
        while (top > 0) {
            const Item it = stack[--top];
            const uint32_t ref = it.en.ref;
            const int s = (ref & TRT_LEAF_BIT) ? 0 : split_k[(size_t)ref * 3 + (it.k - 1)];
            if (s == 0) { e[n++] = it.en; continue; }
            Entry g[2];
            children(nodes[ref], g);
            stack[top++] = {g[1], it.k - s};
            stack[top++] = {g[0], s};
            ++dropped;
        }
        return n;
    };
    emitWide(nodes, threads, expand, w);
    return w;
}

// The simple collapse (open the child with the largest box until the node is full).  trt_create uses it above 4 M
// triangles, where it measured better (blob-10M at 4K: 10.95 against 11.34 visits per ray, +3 % rays/s; the area model
// of the dynamic programme fits a finely tessellated closed surface, whose rays start on it, less well), and
// TRT_WIDE_GREEDY=0/1 forces either one for A/B runs.
inline WideTree collapseBvhGreedy(const trt_bvh_node* nodes, uint32_t n_nodes, unsigned threads = 1)
{
    using namespace wide_detail;
    WideTree w;
    if (n_nodes == 0) return w;
    auto expand = [&](uint32_t bvh2, Entry* e, uint64_t& dropped) -> int {
        int n = 2;
        children(nodes[bvh2], e);
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
            ++dropped;
        }
        return n;
    };
    emitWide(nodes, threads, expand, w);
    return w;
}

// For every triangle i the caller's box of the leaf it lies in ([2 i] = (lo.xyz, hi.x), [2 i + 1] = (hi.y, hi.z, 0, 0)):
// leafEntry() of trt_path.h tests a triangle hit against the entry distance of the box of its own leaf.
inline std::vector<f4> leafBoxesOf(const trt_bvh_node* nodes2, uint32_t n_nodes2, uint32_t n_tris, unsigned threads = 1)
{
    std::vector<f4> lb((size_t)std::max<uint32_t>(n_tris, 1u) * 2, mk4(0.f, 0.f, 0.f, 0.f));
    // (a triangle lies in one leaf: no two nodes write the same entry)
    par::forRange(n_nodes2, threads, 65536, [&](size_t n0, size_t n1) {
    for (size_t n = n0; n < n1; ++n) {
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
    });
    return lb;
}

}  // namespace trtd
