// bvh.cpp — host BVH builders producing the flat 64-B node array of trt.h.
// Role of buildBVH (bvh.cpp:16-144 of the reference, called at main.cpp:76):
// top-down SAH over triangle centroids, leaves of <= leaf_num triangles, node
// boxes padded by +-0.001 (bvh.cpp:31-40), triangles reordered into leaf
// order.  Written from scratch on an index permutation (the reference sorts
// whole 168-byte Triangle objects three times per node); topology is free to
// differ — only nearest-hit and the tie rules matter.
#include <algorithm>
#include <cstring>
#include <limits>
#include <numeric>
#include <stdexcept>

#include "scene.h"

namespace trt {
namespace {

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

    // 32-bin SAH on the centroid bounds; O(n) per node.
    bool binnedSplit(size_t lo, size_t hi, size_t& mid_out)
    {
        constexpr int NB = 32;
        Box cb;
        for (size_t i = lo; i < hi; ++i) cb.grow(prims[idx[i]].c);
        float best = std::numeric_limits<float>::max();
        int best_axis = -1, best_bin = 0;
        for (int axis = 0; axis < 3; ++axis) {
            const float ext = cb.hi[axis] - cb.lo[axis];
            if (!(ext > 0.f)) continue;
            const float scale = (float)NB / ext;
            Box bb[NB];
            uint32_t cnt[NB] = {0};
            for (size_t i = lo; i < hi; ++i) {
                const Prim& p = prims[idx[i]];
                int b = (int)((p.c[axis] - cb.lo[axis]) * scale);
                b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                bb[b].grow(p.box);
                cnt[b]++;
            }
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
        const float ext = cb.hi[best_axis] - cb.lo[best_axis];
        const float scale = (float)NB / ext;
        const float base = cb.lo[best_axis];
        auto it = std::partition(idx.begin() + lo, idx.begin() + hi, [&](uint32_t a) {
            int b = (int)((prims[a].c[best_axis] - base) * scale);
            b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
            return b < best_bin;
        });
        mid_out = (size_t)(it - idx.begin());
        return mid_out > lo && mid_out < hi;
    }

    // Chooses the split of idx[lo,hi) (reordering that range) and returns its position.
    size_t split(size_t lo, size_t hi, uint32_t depth)
    {
        const size_t n = hi - lo;
        size_t mid = lo + n / 2;
        const bool use_sweep = kind == BVH_SWEEP_SAH || (kind == BVH_AUTO && n <= 65536);
        bool ok = false;
        if (depth < 56) {
            if (use_sweep) { int axis; ok = sweepSplit(lo, hi, axis, mid); }
            else ok = binnedSplit(lo, hi, mid);
        }
        if (!ok) mid = lo + n / 2;  // degenerate input (coincident centroids) or a runaway depth: median by index
        return mid;
    }

    // Builds the subtree over idx[lo,hi) into `out` (appending, parents before children) and returns its
    // child reference; inner references are absolute indices into `out`.
    uint32_t build(size_t lo, size_t hi, uint32_t depth, std::vector<trt_bvh_node>& out, uint32_t& deepest)
    {
        const size_t n = hi - lo;
        if (n <= (size_t)leaf_num) {
            if (depth > deepest) deepest = depth;
            return TRT_MAKE_LEAF(lo, n);
        }
        const size_t mid = split(lo, hi, depth);
        const uint32_t me = (uint32_t)out.size();
        out.emplace_back();
        const Box b0 = bounds(lo, mid), b1 = bounds(mid, hi);
        uint32_t c0, c1;
        if (n >= PARALLEL_MIN) {
            // big subtrees: the two halves are independent -> OpenMP tasks, each into its own node vector
            // (own scratch too), spliced back in pre-order with their inner references shifted
            std::vector<trt_bvh_node> left, right;
            uint32_t dl = 0, dr = 0, rl = 0, rr = 0;
#pragma omp task shared(left, dl, rl) if (n >= PARALLEL_MIN)
            {
                Builder sub(prims, idx, leaf_num, kind);
                rl = sub.build(lo, mid, depth + 1, left, dl);
            }
#pragma omp task shared(right, dr, rr) if (n >= PARALLEL_MIN)
            {
                Builder sub(prims, idx, leaf_num, kind);
                rr = sub.build(mid, hi, depth + 1, right, dr);
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
            c0 = build(lo, mid, depth + 1, out, deepest);
            c1 = build(mid, hi, depth + 1, out, deepest);
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
#pragma omp parallel
#pragma omp single
    root = b.build(0, n, 0, out.nodes, deepest);
    if (root != 0) throw std::runtime_error("buildBVH: internal error (root index)");
    out.depth = deepest;

    // reorder the triangles into leaf order (the reference's in-place sorts)
    std::vector<Triangle> sorted;
    sorted.reserve(n);
    for (size_t i = 0; i < n; ++i) sorted.push_back(std::move(triangles[idx[i]]));
    triangles.swap(sorted);
    return out;
}

void FlatScene::build(const Scene& scene, const FlatBVH& bvh)
{
    const size_t n = scene.triangles.size();
    tri_v.resize(n * 9);
    tri_vn.resize(n * 9);
    tri_vt.resize(n * 6);
    tri_mat.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const Triangle& t = scene.triangles[i];
        for (int k = 0; k < 3; ++k) {
            tri_v[i * 9 + k * 3 + 0] = t.v[k].x; tri_v[i * 9 + k * 3 + 1] = t.v[k].y; tri_v[i * 9 + k * 3 + 2] = t.v[k].z;
            tri_vn[i * 9 + k * 3 + 0] = t.vn[k].x; tri_vn[i * 9 + k * 3 + 1] = t.vn[k].y; tri_vn[i * 9 + k * 3 + 2] = t.vn[k].z;
            tri_vt[i * 6 + k * 2 + 0] = t.vt[k].x; tri_vt[i * 6 + k * 2 + 1] = t.vt[k].y;
        }
        if (t.mtl_id < 0 || t.mtl_id >= (int)scene.materials.size()) throw std::runtime_error("flatten: triangle without material");
        tri_mat[i] = t.mtl_id;
    }
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
