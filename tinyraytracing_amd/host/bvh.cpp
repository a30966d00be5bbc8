// bvh.cpp — host BVH builders producing the flat 64-B node array of trt.h.
// Role of buildBVH (bvh.cpp:16-144 of the reference, called at main.cpp:76):
// top-down SAH over triangle centroids, leaves of <= leaf_num triangles, node
// boxes padded by +-0.001 (bvh.cpp:31-40), triangles reordered into leaf
// order.  Written from scratch on an index permutation (the reference sorts
// whole 168-byte Triangle objects three times per node); topology is free to
// differ — only nearest-hit and the tie rules matter.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <stdexcept>

#include <omp.h>

#include "scene.h"

namespace trt {

int hostThreads()
{
    int t = omp_get_max_threads();
    if (t > 16) t = 16;
    if (const char* e = std::getenv("TRT_HOST_THREADS")) t = std::max(1, std::atoi(e));
    return t;
}

namespace {

// Threads of the builder: TRT_HOST_THREADS, else min(16, what OpenMP would take).  A GPU box hands a process a share of a big host
// (16 CPUs of 256 here): a team of 256 threads on that share took 4.7 s for 10 M triangles where 16 take a third of it
// (profiles/r03_create_cost_10m.txt) — every task wait then waits for threads that are not running.
int builderThreads() { return hostThreads(); }

struct Box {
    vec3 lo = vec3(std::numeric_limits<float>::max());
    vec3 hi = vec3(-std::numeric_limits<float>::max());
    void grow(const Box& b) { lo = vmin(lo, b.lo); hi = vmax(hi, b.hi); }
    void grow(vec3 p) { lo = vmin(lo, p); hi = vmax(hi, p); }
    float halfArea() const
    {
        const float x = hi.x - lo.x, y = hi.y - lo.y, z = hi.z - lo.z;
        return x * y + x * z + y * z;
    }
};

struct Prim {
    Box box;   // unpadded triangle bounds
    vec3 c;    // centroid (Triangle::center, scene.cpp:197)
};

struct Builder {
    const std::vector<Prim>& prims;
    std::vector<uint32_t>& idx;
    int leaf_num;
    BvhBuilder kind;
    std::vector<float> scratch_area;  // suffix areas for the sweep

    Builder(const std::vector<Prim>& p, std::vector<uint32_t>& i, int leaf, BvhBuilder k) : prims(p), idx(i), leaf_num(leaf), kind(k) {}

    Box bounds(size_t lo, size_t hi) const
    {
        Box b;
        for (size_t i = lo; i < hi; ++i) b.grow(prims[idx[i]].box);
        return b;
    }

    static void storeBox(float* lo3, float* hi3, const Box& b)
    {
        // bvh.cpp:31-40: every node's box is the triangle bounds -/+ 0.001f
        lo3[0] = b.lo.x - 0.001f; lo3[1] = b.lo.y - 0.001f; lo3[2] = b.lo.z - 0.001f;
        hi3[0] = b.hi.x + 0.001f; hi3[1] = b.hi.y + 0.001f; hi3[2] = b.hi.z + 0.001f;
    }

    // Exact SAH over all n-1 split positions on each axis (the reference's
    // strategy).  Returns false when no split beats making a leaf impossible
    // (n > leaf_num always splits; ties fall back to the median).
    bool sweepSplit(size_t lo, size_t hi, int& axis_out, size_t& mid_out)
    {
        const size_t n = hi - lo;
        float best = std::numeric_limits<float>::max();
        int best_axis = -1;
        size_t best_mid = lo + n / 2;
        if (scratch_area.size() < n) scratch_area.resize(n);
        for (int axis = 0; axis < 3; ++axis) {
            std::sort(idx.begin() + lo, idx.begin() + hi, [&](uint32_t a, uint32_t b) {
                const float ca = prims[a].c[axis], cb = prims[b].c[axis];
                return ca < cb || (ca == cb && a < b);
            });
            Box acc;
            for (size_t i = n; i-- > 1;) {
                acc.grow(prims[idx[lo + i]].box);
                scratch_area[i] = acc.halfArea();
            }
            acc = Box();
            for (size_t i = 1; i < n; ++i) {
                acc.grow(prims[idx[lo + i - 1]].box);
                const float cost = acc.halfArea() * (float)i + scratch_area[i] * (float)(n - i);
                if (cost < best) { best = cost; best_axis = axis; best_mid = lo + i; }
            }
        }
        if (best_axis < 0) { best_axis = 0; best_mid = lo + n / 2; }
        if (best_axis != 2)
            std::sort(idx.begin() + lo, idx.begin() + hi, [&](uint32_t a, uint32_t b) {
                const float ca = prims[a].c[best_axis], cb = prims[b].c[best_axis];
                return ca < cb || (ca == cb && a < b);
            });
        axis_out = best_axis;
        mid_out = best_mid;
        return true;
    }

    // 32-bin SAH on the centroid bounds; O(n) per node: one pass for the centroid bounds, ONE for the bins of all three axes, one
    // for the partition; the boxes of the two sides fall out of the bins (unions of the same triangle boxes).  A big node (the top of
    // a 10 M-triangle tree: the root alone is 10 M gathers per pass) splits every pass into chunks run as OpenMP tasks; the
    // partition is then a stable one through a scratch array (chunk counts, prefix, scatter), so the result does not depend on the
    // number of chunks or threads.
    static constexpr int NB = 32;
    static constexpr size_t CHUNK = 131072;  // a node of >= 2 chunks is split across tasks
    struct Bins {
        Box bb[3][NB];
        uint32_t cnt[3][NB];
        Bins() { std::memset(cnt, 0, sizeof(cnt)); }
        void merge(const Bins& o)
        {
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < NB; ++b) { bb[a][b].grow(o.bb[a][b]); cnt[a][b] += o.cnt[a][b]; }
        }
    };
    static int binOf(float c, float base, float scale)
    {
        int b = (int)((c - base) * scale);
        return b < 0 ? 0 : (b >= NB ? NB - 1 : b);
    }
    template <class F>
    static void chunked(size_t lo, size_t hi, F f)  // f(chunk index, begin, end); chunks of CHUNK elements as tasks
    {
        const size_t n = hi - lo, nc = (n + CHUNK - 1) / CHUNK;
        if (nc <= 1) { f((size_t)0, lo, hi); return; }
        for (size_t c = 0; c < nc; ++c) {
            const size_t b = lo + c * CHUNK, e = std::min(hi, b + CHUNK);
#pragma omp task firstprivate(c, b, e) shared(f)
            f(c, b, e);
        }
#pragma omp taskwait
    }
    bool binnedSplit(size_t lo, size_t hi, size_t& mid_out, Box& b0, Box& b1)
    {
        const size_t n = hi - lo, nc = (n + CHUNK - 1) / CHUNK;
        Box cb;
        {
            std::vector<Box> part(nc);
            chunked(lo, hi, [&](size_t c, size_t b, size_t e) {
                Box x;
                for (size_t i = b; i < e; ++i) x.grow(prims[idx[i]].c);
                part[c] = x;
            });
            for (const Box& x : part) cb.grow(x);
        }
        float scale[3];
        for (int axis = 0; axis < 3; ++axis) {
            const float ext = cb.hi[axis] - cb.lo[axis];
            scale[axis] = ext > 0.f ? (float)NB / ext : 0.f;
        }
        Bins bins;
        {
            std::vector<Bins> part(nc);
            chunked(lo, hi, [&](size_t c, size_t b, size_t e) {
                Bins& x = part[c];
                for (size_t i = b; i < e; ++i) {
                    const Prim& p = prims[idx[i]];
                    for (int axis = 0; axis < 3; ++axis) {
                        if (!(scale[axis] > 0.f)) continue;
                        const int k = binOf(p.c[axis], cb.lo[axis], scale[axis]);
                        x.bb[axis][k].grow(p.box);
                        x.cnt[axis][k]++;
                    }
                }
            });
            for (const Bins& x : part) bins.merge(x);
        }
        float best = std::numeric_limits<float>::max();
        int best_axis = -1, best_bin = 0;
        for (int axis = 0; axis < 3; ++axis) {
            if (!(scale[axis] > 0.f)) continue;
            const Box* bb = bins.bb[axis];
            const uint32_t* cnt = bins.cnt[axis];
            float ra[NB];
            uint32_t rc[NB];
            Box acc;
            uint32_t c = 0;
            for (int b = NB - 1; b >= 1; --b) {
                acc.grow(bb[b]);
                c += cnt[b];
                ra[b] = acc.halfArea();
                rc[b] = c;
            }
            acc = Box();
            c = 0;
            for (int b = 1; b < NB; ++b) {
                acc.grow(bb[b - 1]);
                c += cnt[b - 1];
                if (c == 0 || rc[b] == 0) continue;
                const float cost = acc.halfArea() * (float)c + ra[b] * (float)rc[b];
                if (cost < best) { best = cost; best_axis = axis; best_bin = b; }
            }
        }
        if (best_axis < 0) return false;
        const float sc = scale[best_axis], base = cb.lo[best_axis];
        auto left = [&](uint32_t a) { return binOf(prims[a].c[best_axis], base, sc) < best_bin; };
        if (nc <= 1) {
            auto it = std::partition(idx.begin() + lo, idx.begin() + hi, left);
            mid_out = (size_t)(it - idx.begin());
        } else {
            std::vector<size_t> nleft(nc, 0);
            chunked(lo, hi, [&](size_t c, size_t b, size_t e) {
                size_t k = 0;
                for (size_t i = b; i < e; ++i) k += left(idx[i]) ? 1 : 0;
                nleft[c] = k;
            });
            std::vector<size_t> lpos(nc), rpos(nc);
            size_t total_left = 0;
            for (size_t c = 0; c < nc; ++c) { lpos[c] = total_left; total_left += nleft[c]; }
            size_t r = total_left;
            for (size_t c = 0; c < nc; ++c) { rpos[c] = r; r += std::min(hi, lo + (c + 1) * CHUNK) - (lo + c * CHUNK) - nleft[c]; }
            std::vector<uint32_t> tmp(n);
            chunked(lo, hi, [&](size_t c, size_t b, size_t e) {
                size_t l = lpos[c], q = rpos[c];
                for (size_t i = b; i < e; ++i) {
                    const uint32_t a = idx[i];
                    if (left(a)) tmp[l++] = a; else tmp[q++] = a;
                }
            });
            chunked(lo, hi, [&](size_t, size_t b, size_t e) { std::memcpy(&idx[b], &tmp[b - lo], (e - b) * sizeof(uint32_t)); });
            mid_out = lo + total_left;
        }
        if (!(mid_out > lo && mid_out < hi)) return false;
        b0 = Box();
        b1 = Box();
        for (int b = 0; b < NB; ++b) (b < best_bin ? b0 : b1).grow(bins.bb[best_axis][b]);
        return true;
    }

    // Chooses the split of idx[lo,hi) (reordering that range) and returns its position.
    // `have_boxes`: b0 / b1 are the bounds of the two sides already (the binned split knows them).
    size_t split(size_t lo, size_t hi, uint32_t depth, bool& have_boxes, Box& b0, Box& b1)
    {
        const size_t n = hi - lo;
        size_t mid = lo + n / 2;
        const bool use_sweep = kind == BVH_SWEEP_SAH || (kind == BVH_AUTO && n <= 65536);
        bool ok = false;
        have_boxes = false;
        if (depth < 56) {
            if (use_sweep) { int axis; ok = sweepSplit(lo, hi, axis, mid); }
            else have_boxes = ok = binnedSplit(lo, hi, mid, b0, b1);
        }
        if (!ok) mid = lo + n / 2;  // degenerate input (coincident centroids) or a runaway depth: median by index
        return mid;
    }

    // ---- exact SAH below SWEEP_MAX triangles, the three centroid orders kept instead of re-made ----
    // sweepSplit() sorts a node's range three to four times; over the 16 levels below 65 536 triangles that was 60 % of the
    // build of a 2 M-triangle scene.  Here the subtree's range is sorted ONCE per axis (same comparator: a total order), a node
    // is evaluated by scanning the three orders — the same boxes grown in the same sequence, so the same costs and the same
    // choice as sweepSplit() —, and the two orders of the axes not chosen are split by a stable partition, which leaves each side
    // in exactly the order a fresh sort would give it.  A leaf takes its triangles in the order of its parent's split axis, which
    // is where sweepSplit() leaves them.  Same tree, same triangle order, O(n) per node.
    struct Sorted {
        size_t base = 0;              // order[a][i - base] for position i of idx
        std::vector<uint32_t> order[3];
        std::vector<uint32_t> tmp;
    };
    uint8_t* side = nullptr;          // per triangle: which side of the current split (shared by all tasks: disjoint triangles)

    uint32_t buildSorted(Sorted& S, size_t lo, size_t hi, uint32_t depth, int parent_axis, std::vector<trt_bvh_node>& out, uint32_t& deepest)
    {
        const size_t n = hi - lo;
        auto at = [&](int a, size_t i) -> uint32_t& { return S.order[a][i - S.base]; };
        if (n <= (size_t)leaf_num || depth >= 56) {
            if (parent_axis >= 0)
                for (size_t i = lo; i < hi; ++i) idx[i] = at(parent_axis, i);
            // (a runaway depth: build() goes on with medians by index and never sorts again)
            return build(lo, hi, depth, out, deepest, false);
        }
        float best = std::numeric_limits<float>::max();
        int best_axis = -1;
        size_t best_mid = lo + n / 2;
        if (scratch_area.size() < n) scratch_area.resize(n);
        for (int axis = 0; axis < 3; ++axis) {
            Box acc;
            for (size_t i = n; i-- > 1;) {
                acc.grow(prims[at(axis, lo + i)].box);
                scratch_area[i] = acc.halfArea();
            }
            acc = Box();
            for (size_t i = 1; i < n; ++i) {
                acc.grow(prims[at(axis, lo + i - 1)].box);
                const float cost = acc.halfArea() * (float)i + scratch_area[i] * (float)(n - i);
                if (cost < best) { best = cost; best_axis = axis; best_mid = lo + i; }
            }
        }
        if (best_axis < 0) { best_axis = 0; best_mid = lo + n / 2; }
        const size_t mid = best_mid;
        Box b0, b1;
        for (size_t i = lo; i < mid; ++i) { const uint32_t a = at(best_axis, i); side[a] = 0; b0.grow(prims[a].box); }
        for (size_t i = mid; i < hi; ++i) { const uint32_t a = at(best_axis, i); side[a] = 1; b1.grow(prims[a].box); }
        for (int axis = 0; axis < 3; ++axis) {
            if (axis == best_axis) continue;
            size_t l = lo, r = 0;
            if (S.tmp.size() < n) S.tmp.resize(n);
            for (size_t i = lo; i < hi; ++i) {
                const uint32_t a = at(axis, i);
                if (side[a]) S.tmp[r++] = a; else at(axis, l++) = a;  // (l <= i: nothing unread is overwritten)
            }
            for (size_t k = 0; k < r; ++k) at(axis, l + k) = S.tmp[k];
        }
        const uint32_t me = (uint32_t)out.size();
        out.emplace_back();
        const uint32_t c0 = buildSorted(S, lo, mid, depth + 1, best_axis, out, deepest);
        const uint32_t c1 = buildSorted(S, mid, hi, depth + 1, best_axis, out, deepest);
        trt_bvh_node& nd = out[me];
        storeBox(nd.lo0, nd.hi0, b0);
        storeBox(nd.lo1, nd.hi1, b1);
        nd.child0 = c0;
        nd.child1 = c1;
        nd.reserved[0] = nd.reserved[1] = 0;
        return me;
    }

    // Builds the subtree over idx[lo,hi) into `out` (appending, parents before children) and returns its
    // child reference; inner references are absolute indices into `out`.
    uint32_t build(size_t lo, size_t hi, uint32_t depth, std::vector<trt_bvh_node>& out, uint32_t& deepest, bool sorted_ok = true)
    {
        const size_t n = hi - lo;
        if (n <= (size_t)leaf_num) {
            if (depth > deepest) deepest = depth;
            return TRT_MAKE_LEAF(lo, n);
        }
        if (sorted_ok && side && depth < 56 && n < PARALLEL_MIN && (kind == BVH_SWEEP_SAH || (kind == BVH_AUTO && n <= 65536))) {
            Sorted S;
            S.base = lo;
            for (int axis = 0; axis < 3; ++axis) {
                S.order[axis].assign(idx.begin() + (long)lo, idx.begin() + (long)hi);
                std::sort(S.order[axis].begin(), S.order[axis].end(), [&](uint32_t a, uint32_t b) {
                    const float ca = prims[a].c[axis], cb = prims[b].c[axis];
                    return ca < cb || (ca == cb && a < b);
                });
            }
            return buildSorted(S, lo, hi, depth, -1, out, deepest);
        }
        bool have_boxes = false;
        Box b0, b1;
        const size_t mid = split(lo, hi, depth, have_boxes, b0, b1);
        const uint32_t me = (uint32_t)out.size();
        out.emplace_back();
        if (!have_boxes) { b0 = bounds(lo, mid); b1 = bounds(mid, hi); }
        uint32_t c0, c1;
        if (n >= PARALLEL_MIN) {
            // big subtrees: the two halves are independent -> OpenMP tasks, each into its own node vector
            // (own scratch too), spliced back in pre-order with their inner references shifted
            std::vector<trt_bvh_node> left, right;
            uint32_t dl = 0, dr = 0, rl = 0, rr = 0;
#pragma omp task shared(left, dl, rl) if (n >= PARALLEL_MIN)
            {
                Builder sub(prims, idx, leaf_num, kind);
                sub.side = side;
                rl = sub.build(lo, mid, depth + 1, left, dl, sorted_ok);
            }
#pragma omp task shared(right, dr, rr) if (n >= PARALLEL_MIN)
            {
                Builder sub(prims, idx, leaf_num, kind);
                sub.side = side;
                rr = sub.build(mid, hi, depth + 1, right, dr, sorted_ok);
            }
#pragma omp taskwait
            auto splice = [&](std::vector<trt_bvh_node>& sub, uint32_t ref) -> uint32_t {
                const uint32_t off = (uint32_t)out.size();
                for (trt_bvh_node& nd : sub) {
                    if (!(nd.child0 & TRT_LEAF_BIT)) nd.child0 += off;
                    if (!(nd.child1 & TRT_LEAF_BIT)) nd.child1 += off;
                }
                out.insert(out.end(), sub.begin(), sub.end());
                return (ref & TRT_LEAF_BIT) ? ref : ref + off;
            };
            c0 = splice(left, rl);
            c1 = splice(right, rr);
            deepest = std::max(deepest, std::max(dl, dr));
        } else {
            c0 = build(lo, mid, depth + 1, out, deepest, sorted_ok);
            c1 = build(mid, hi, depth + 1, out, deepest, sorted_ok);
        }
        trt_bvh_node& nd = out[me];
        storeBox(nd.lo0, nd.hi0, b0);
        storeBox(nd.lo1, nd.hi1, b1);
        nd.child0 = c0;
        nd.child1 = c1;
        nd.reserved[0] = nd.reserved[1] = 0;
        return me;
    }
    static constexpr size_t PARALLEL_MIN = 200000;
};

}  // namespace

FlatBVH buildBVH(std::vector<Triangle>& triangles, int leaf_num, BvhBuilder builder)
{
    if (leaf_num < 1 || leaf_num > (int)TRT_MAX_LEAF_TRIS) throw std::runtime_error("buildBVH: leaf_num must be in 1..15");
    if (triangles.size() > TRT_MAX_TRIS) throw std::runtime_error("buildBVH: too many triangles");
    const size_t n = triangles.size();
    std::vector<Prim> prims(n);
    const int threads = builderThreads();
#pragma omp parallel for schedule(static) num_threads(threads) if (n >= 100000)
    for (size_t i = 0; i < n; ++i) {
        const Triangle& t = triangles[i];
        prims[i].box.grow(t.v[0]);
        prims[i].box.grow(t.v[1]);
        prims[i].box.grow(t.v[2]);
        prims[i].c = t.center;
    }
    std::vector<uint32_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0u);

    Builder b(prims, idx, leaf_num, builder);
    std::vector<uint8_t> side(n, 0);
    b.side = side.data();
    FlatBVH out;
    if (n <= (size_t)leaf_num) {
        // A scene that fits one leaf still gets a root node: child0 = all
        // triangles, child1 = empty leaf.
        trt_bvh_node root;
        std::memset(&root, 0, sizeof(root));
        Box bx = n ? b.bounds(0, n) : Box();
        if (!n) { bx.lo = vec3(0.f); bx.hi = vec3(0.f); }
        Builder::storeBox(root.lo0, root.hi0, bx);
        Builder::storeBox(root.lo1, root.hi1, bx);
        root.child0 = TRT_MAKE_LEAF(0, n);
        root.child1 = TRT_MAKE_LEAF(0, 0);
        out.nodes.push_back(root);
        out.depth = 1;
        return out;
    }
    uint32_t root = 0, deepest = 0;
    out.nodes.reserve(n / (size_t)std::max(1, leaf_num / 2) + 4);
#pragma omp parallel num_threads(threads)
#pragma omp single
    root = b.build(0, n, 0, out.nodes, deepest);
    if (root != 0) throw std::runtime_error("buildBVH: internal error (root index)");
    out.depth = deepest;

    // reorder the triangles into leaf order (the reference's in-place sorts)
    std::vector<Triangle> sorted(n);
#pragma omp parallel for schedule(static) num_threads(threads) if (n >= 100000)
    for (size_t i = 0; i < n; ++i) sorted[i] = std::move(triangles[idx[i]]);  // idx is a permutation: every source moved once
    triangles.swap(sorted);
    return out;
}

void FlatScene::build(const Scene& scene, const FlatBVH& bvh, const uint32_t* order)
{
    const size_t n = scene.triangles.size();
    tri_v.resize(n * 9);
    tri_vn.resize(n * 9);
    tri_vt.resize(n * 6);
    tri_mat.resize(n);
    int bad_material = 0;
    const int threads = builderThreads();
#pragma omp parallel for schedule(static) num_threads(threads) if (n >= 100000)
    for (size_t i = 0; i < n; ++i) {
        const Triangle& t = scene.triangles[order ? order[i] : i];
        for (int k = 0; k < 3; ++k) {
            tri_v[i * 9 + k * 3 + 0] = t.v[k].x; tri_v[i * 9 + k * 3 + 1] = t.v[k].y; tri_v[i * 9 + k * 3 + 2] = t.v[k].z;
            tri_vn[i * 9 + k * 3 + 0] = t.vn[k].x; tri_vn[i * 9 + k * 3 + 1] = t.vn[k].y; tri_vn[i * 9 + k * 3 + 2] = t.vn[k].z;
            tri_vt[i * 6 + k * 2 + 0] = t.vt[k].x; tri_vt[i * 6 + k * 2 + 1] = t.vt[k].y;
        }
        if (t.mtl_id < 0 || t.mtl_id >= (int)scene.materials.size()) {
#pragma omp atomic write
            bad_material = 1;  // (no exception out of a parallel loop)
        }
        tri_mat[i] = t.mtl_id;
    }
    if (bad_material) throw std::runtime_error("flatten: triangle without material");
    nodes = bvh.nodes;

    materials.resize(scene.materials.size());
    textures.clear();
    texture_data.clear();
    for (size_t i = 0; i < scene.materials.size(); ++i) {
        const Material& m = scene.materials[i];
        trt_material& f = materials[i];
        f.Kd[0] = m.Kd.x; f.Kd[1] = m.Kd.y; f.Kd[2] = m.Kd.z;
        f.Ks[0] = m.Ks.x; f.Ks[1] = m.Ks.y; f.Ks[2] = m.Ks.z;
        f.Tr[0] = m.Tr.x; f.Tr[1] = m.Tr.y; f.Tr[2] = m.Tr.z;
        f.Ns = m.Ns;
        f.Ni = m.Ni;
        f.radiance[0] = m.radiance.x; f.radiance[1] = m.radiance.y; f.radiance[2] = m.radiance.z;
        f.is_emissive = m.is_emissive ? 1 : 0;
        f.tex = -1;
        // shade() takes the texture branch whenever map_Kd != "" (pathTracing.cpp:17);
        // an undecodable file would index an empty cv::Mat there, so it is
        // rejected here instead.
        if (!m.map_Kd.empty()) {
            if (m.img.empty()) throw std::runtime_error("flatten: texture not loaded: " + m.map_Kd);
            f.tex = (int32_t)texture_data.size();
            texture_data.push_back(m.img);
            trt_texture tx;
            tx.width = m.map_width;
            tx.height = m.map_height;
            tx.rgb = nullptr;
            textures.push_back(tx);
        }
    }
    for (size_t i = 0; i < textures.size(); ++i) textures[i].rgb = texture_data[i].data();

    lights.clear();
    light_tris.clear();
    for (const Light& l : scene.lights) {
        auto it = scene.material_ids.find(l.mtl_name);
        if (it == scene.material_ids.end()) throw std::runtime_error("flatten: light material missing: " + l.mtl_name);
        const Material& m = scene.materials[(size_t)it->second];
        trt_light fl;
        fl.mat = it->second;
        fl.radiance[0] = m.radiance.x; fl.radiance[1] = m.radiance.y; fl.radiance[2] = m.radiance.z;
        fl.area = (float)m.area;
        fl.tri_first = (uint32_t)light_tris.size();
        fl.tri_count = (uint32_t)m.triangles.size();
        for (const Triangle& t : m.triangles) {
            trt_light_tri lt;
            for (int k = 0; k < 3; ++k) {
                lt.v[k][0] = t.v[k].x; lt.v[k][1] = t.v[k].y; lt.v[k][2] = t.v[k].z;
                lt.vn[k][0] = t.vn[k].x; lt.vn[k][1] = t.vn[k].y; lt.vn[k][2] = t.vn[k].z;
            }
            lt.cum_area = (float)t.area;
            light_tris.push_back(lt);
        }
        lights.push_back(fl);
    }

    flat = trt_scene{};
    flat.n_tris = (uint32_t)n;
    flat.tri_v = tri_v.data();
    flat.tri_vn = tri_vn.data();
    flat.tri_vt = tri_vt.data();
    flat.tri_mat = tri_mat.data();
    flat.n_nodes = (uint32_t)nodes.size();
    flat.nodes = nodes.data();
    flat.bvh_depth = bvh.depth;
    flat.n_materials = (uint32_t)materials.size();
    flat.materials = materials.data();
    flat.n_lights = (uint32_t)lights.size();
    flat.lights = lights.data();
    flat.n_light_tris = (uint32_t)light_tris.size();
    flat.light_tris = light_tris.data();
    flat.n_textures = (uint32_t)textures.size();
    flat.textures = textures.data();
    const Camera& c = scene.camera;
    flat.camera.eye[0] = c.eye.x; flat.camera.eye[1] = c.eye.y; flat.camera.eye[2] = c.eye.z;
    flat.camera.lower_left_corner[0] = c.lower_left_corner.x; flat.camera.lower_left_corner[1] = c.lower_left_corner.y; flat.camera.lower_left_corner[2] = c.lower_left_corner.z;
    flat.camera.horizontal[0] = c.horizontal.x; flat.camera.horizontal[1] = c.horizontal.y; flat.camera.horizontal[2] = c.horizontal.z;
    flat.camera.vertical[0] = c.vertical.x; flat.camera.vertical[1] = c.vertical.y; flat.camera.vertical[2] = c.vertical.z;
}

}  // namespace trt
