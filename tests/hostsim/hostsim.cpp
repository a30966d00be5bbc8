// hostsim.cpp — TEST INFRASTRUCTURE.  Compiles the device-side path functions
// (tinyraytracing_amd/csrc/trt_path.h: traversal, triangle/box tests,
// shadeBegin / lightSample / shadeNext) with g++ and drives them per path in
// the wavefront kernels' order of operations, so their arithmetic can be checked
// bit-for-bit against the oracle on a machine without a GPU.  It is not a render
// back end: nothing in the product loads this library.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "trt_path.h"
#include "trt_wide.h"
#include "trt_oct_build.h"

using namespace trtd;

namespace {
int g_node_kind = 0;  // hostsim_set_node_kind: 0 = exact 4-wide nodes; 1 = the 8-wide compressed nodes of trt_oct.h where the tree allows them
struct ArrayStack {
    uint32_t s[1024];
    void push(int sp, uint32_t v) { s[sp] = v; }
    uint32_t pop(int sp) const { return s[sp]; }
};

struct OctArrayStack {
    OctGroup s[256];
    void push(int sp, OctGroup g) { s[sp] = g; }
    OctGroup pop(int sp) const { return s[sp]; }
};

struct HostScene {
    std::vector<TriIsect> isect;
    std::vector<TriShade> shade;
    std::vector<MaterialDev> mats;
    std::vector<TextureDev> tex;
    std::vector<uint8_t> tex_bytes;
    std::vector<float> cum;
    std::vector<LightDev> lights;
    std::vector<LightTriDev> ltris;
    WideTree wide;
    OctTree oct;
    std::vector<f4> leaf_boxes;
    std::vector<LightBox> light_boxes;
    std::vector<uint32_t> plane_bits;
    int nk = 0;  // node kind the traversal walks: 0 exact 4-wide nodes, 1 compressed 8-wide nodes (TRT_NODE_KIND)
    SceneDev sc{};
    explicit HostScene(const trt_scene* s)
    {
        isect.resize(s->n_tris);
        shade.resize(s->n_tris);
        for (uint32_t i = 0; i < s->n_tris; ++i) {
            const int32_t mat = s->tri_mat[i];
            isect[i] = makeTriIsect(s->tri_v + (size_t)i * 9, mat, s->materials[mat].is_emissive != 0);
            std::memcpy(shade[i].vn, s->tri_vn + (size_t)i * 9, 36);
            std::memcpy(shade[i].vt, s->tri_vt + (size_t)i * 6, 24);
            shade[i].mat = mat;
        }
        mats.resize(s->n_materials);
        for (uint32_t i = 0; i < s->n_materials; ++i) mats[i] = makeMaterialDev(s->materials[i]);
        lights.resize(s->n_lights);
        for (uint32_t i = 0; i < s->n_lights; ++i) lights[i] = makeLightDev(s->lights[i], s->materials);
        ltris.resize(s->n_light_tris);
        for (uint32_t i = 0; i < s->n_light_tris; ++i) ltris[i] = makeLightTriDev(s->light_tris[i]);
        tex.resize(s->n_textures);
        for (uint32_t i = 0; i < s->n_textures; ++i) {
            tex[i].width = s->textures[i].width;
            tex[i].height = s->textures[i].height;
            tex[i].offset = tex_bytes.size();
            const size_t nb = (size_t)tex[i].width * tex[i].height * 3;
            tex_bytes.insert(tex_bytes.end(), s->textures[i].rgb, s->textures[i].rgb + nb);
        }
        sc.nodes = s->nodes;
        wide = collapseBvh(s->nodes, s->n_nodes);
        sc.wnodes = wide.nodes.data();
        sc.n_wnodes = (uint32_t)wide.nodes.size();
        if (g_node_kind != 0) oct = buildOct(s->nodes, s->n_nodes, s->n_tris, isect.data());
        sc.onodes = oct.ok ? oct.nodes.data() : nullptr;
        sc.tri_trav = oct.ok ? oct.tri_trav.data() : nullptr;
        sc.n_onodes = (uint32_t)oct.nodes.size();
        leaf_boxes = leafBoxesOf(s->nodes, s->n_nodes, s->n_tris);
        light_boxes = lightBoxesOf(leaf_boxes, s->tri_mat, s->n_tris, s->lights, s->n_lights);
        sc.leaf_box = leaf_boxes.data();
        sc.leaf_alpha = sceneLeafAlpha(s->nodes, s->n_nodes);
        sc.plane_shift = 32u - wide_detail::planeFilterBuild(s->nodes, s->n_nodes, plane_bits);
        sc.plane_bits = plane_bits.data();
        sc.cull_alpha = wide_detail::boxesNested(s->nodes, s->n_nodes) ? sc.leaf_alpha : std::numeric_limits<float>::infinity();  // as trt_create
        nk = (oct.ok && g_node_kind != 0) ? 1 : 0;
        sc.tri_isect = isect.data();
        sc.tri_shade = shade.data();
        sc.materials = mats.data();
        sc.lights = lights.data();
        sc.light_tris = ltris.data();
        bool mono = true;
        for (uint32_t l = 0; l < s->n_lights; ++l)
            for (uint32_t k = 0; k < s->lights[l].tri_count; ++k) {
                const float c = s->light_tris[s->lights[l].tri_first + k].cum_area;
                if (!(c == c) || (k && c < s->light_tris[s->lights[l].tri_first + k - 1].cum_area)) mono = false;
            }
        cum.resize(s->n_light_tris);
        for (uint32_t k = 0; k < s->n_light_tris; ++k) cum[k] = s->light_tris[k].cum_area;
        sc.light_cum = mono ? cum.data() : nullptr;
        sc.textures = tex.data();
        sc.tex_bytes = tex_bytes.data();
        sc.n_tris = s->n_tris;
        sc.n_nodes = s->n_nodes;
        sc.n_lights = s->n_lights;
        sc.light0_area = s->n_lights ? s->lights[0].area : 0.0f;
        sc.cam = s->camera;
    }
};
}  // namespace

// 0 (default, as trt_create): exact wide nodes; 1: compressed nodes where the tree allows them.  Returns the previous setting.
extern "C" int hostsim_set_node_kind(int nk)
{
    const int old = g_node_kind;
    g_node_kind = nk;
    return old;
}
// 1 when `s` can be walked with the compressed 8-wide nodes (nested, finite boxes; leaves of more than 3 triangles are split into slots)
extern "C" int hostsim_compressible(const trt_scene* s)
{
    std::vector<TriIsect> isect(s->n_tris);
    for (uint32_t i = 0; i < s->n_tris; ++i) isect[i] = makeTriIsect(s->tri_v + (size_t)i * 9, s->tri_mat[i], s->materials[s->tri_mat[i]].is_emissive != 0);
    return buildOct(s->nodes, s->n_nodes, s->n_tris, isect.data()).ok ? 1 : 0;
}
// FNV-1a hashes of the trees the collapses build with `threads` host threads: [0] 4-wide (dynamic programme), [1] 4-wide (greedy),
// [2] 8-wide nodes, [3] their triangle records, [4..6] stack need / dropped boxes / levels — the results must not depend on the thread count.
extern "C" int hostsim_tree_hashes(const trt_scene* s, unsigned threads, uint64_t out[8])
{
    auto fnv = [](const void* p, size_t n) {
        uint64_t h = 1469598103934665603ull;
        const uint8_t* b = static_cast<const uint8_t*>(p);
        for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
        return h;
    };
    std::vector<TriIsect> isect(s->n_tris);
    for (uint32_t i = 0; i < s->n_tris; ++i) isect[i] = makeTriIsect(s->tri_v + (size_t)i * 9, s->tri_mat[i], s->materials[s->tri_mat[i]].is_emissive != 0);
    const WideTree a = collapseBvh(s->nodes, s->n_nodes, threads), b = collapseBvhGreedy(s->nodes, s->n_nodes, threads);
    const OctTree t = buildOct(s->nodes, s->n_nodes, s->n_tris, isect.data(), threads);
    const std::vector<f4> lb = leafBoxesOf(s->nodes, s->n_nodes, s->n_tris, threads);
    out[0] = fnv(a.nodes.data(), a.nodes.size() * sizeof(WideNode));
    out[1] = fnv(b.nodes.data(), b.nodes.size() * sizeof(WideNode));
    out[2] = t.ok ? fnv(t.nodes.data(), t.nodes.size() * sizeof(OctNode)) : 0;
    out[3] = t.ok ? fnv(t.tri_trav.data(), t.tri_trav.size() * sizeof(TriIsect)) : 0;
    out[4] = ((uint64_t)a.stack_need << 32) | b.stack_need;
    out[5] = a.dropped * 1000003ull + b.dropped;
    out[6] = t.levels;
    out[7] = fnv(lb.data(), lb.size() * sizeof(f4));
    return 0;
}
// nodes / levels of the oct tree (0 when it cannot be built)
extern "C" int hostsim_oct_info(const trt_scene* s, uint64_t out[3])
{
    std::vector<TriIsect> isect(s->n_tris);
    for (uint32_t i = 0; i < s->n_tris; ++i) isect[i] = makeTriIsect(s->tri_v + (size_t)i * 9, s->tri_mat[i], s->materials[s->tri_mat[i]].is_emissive != 0);
    const OctTree t = buildOct(s->nodes, s->n_nodes, s->n_tris, isect.data());
    out[0] = t.ok ? t.nodes.size() : 0; out[1] = t.levels; out[2] = t.tri_trav.size();
    return t.ok ? 0 : 1;
}

extern "C" int hostsim_render(const trt_scene* s, const trt_params* p, float* out_rgb, uint64_t rays[3])
{
    if (!s || !p || !out_rgb) return 1;
    HostScene hs(s);
    std::vector<int32_t> rows;
    for (int y = p->y0; y < p->y1; ++y)
        if (p->row_mod <= 1 || ((y / p->row_block) % p->row_mod) == p->row_rem) rows.push_back(y);
    const uint32_t tw = (uint32_t)(p->x1 - p->x0), npix = (uint32_t)rows.size() * tw;
    TileDesc td;
    td.rows = rows.data();
    td.tile_w = (int32_t)tw; td.x0 = p->x0; td.width = p->width; td.height = p->height;
    td.npix = npix; td.seed = p->seed; td.spp = (uint32_t)p->spp;
    td.fixed_nee = (p->flags & TRT_FLAG_FIXED_NEE) ? 1u : 0u;
    td.fixed_pixels = (p->flags & TRT_FLAG_FIXED_PIXELS) ? 1u : 0u;
    td.ray_offset = (p->flags & TRT_FLAG_RAY_OFFSET) ? 1u : 0u;
    td.specular_ks = (p->flags & TRT_FLAG_SPECULAR_KS) ? 1u : 0u;
    td.npix_magic = magicOf(npix); td.tile_w_magic = magicOf(tw);
    td.grid_ok = 0u;  // the host form of cameraRay divides
    for (double& g : td.grid_rcp) g = 0.0;
    uint64_t r_cam = 0, r_sh = 0, r_ind = 0;
    const uint32_t S = (uint32_t)p->spp;  // one chunk: path id = s * npix + pixel
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : r_cam, r_sh, r_ind)
    for (long pl = 0; pl < (long)npix; ++pl) {
        ArrayStack stk;
        OctArrayStack ostk;
        uint32_t ni = 0, nt = 0;
        double acc[3] = {0, 0, 0};
        for (uint32_t sidx = 0; sidx < S; ++sidx) {
            const uint32_t pid = sidx * npix + (uint32_t)pl;
            // k_gen_primary
            const uint32_t r = (uint32_t)pl / tw, c = (uint32_t)pl - r * tw;
            const int y = rows[r], x = p->x0 + (int)c;
            Stream rng;
            rng.key = trt_rng_make_key(td.seed, (uint32_t)y * (uint32_t)td.width + (uint32_t)x, sidx);
            rng.ctr = 0;
            const float u1 = rng.next(), u2 = rng.next();
            f3 o, d;
            cameraRay(hs.sc.cam, td.width, td.height, y, x, u1, u2, o, d, td.fixed_pixels != 0u);
            f4 ra = mk4(o.x, o.y, o.z, d.x), rb = mk4(d.y, d.z, u2f(pid), u2f(packMeta(rng.ctr, TRT_META_CAMERA, 0))), bt = mk4(1, 1, 1, 0);
            f3 L = mk3(0, 0, 0);
            r_cam++;
            for (;;) {
                // k_trace_closest
                const Hit h = hs.nk ? traceClosestOct<OctArrayStack, ArrayStack, false>(hs.sc, mk3(ra.x, ra.y, ra.z), mk3(ra.w, rb.x, rb.y), ostk, stk, ni, nt)
                                    : traceClosest<ArrayStack, false, 0>(hs.sc, mk3(ra.x, ra.y, ra.z), mk3(ra.w, rb.x, rb.y), stk, ni, nt);
                const f4 hit4 = mk4(h.t, u2f((uint32_t)h.tri), h.u, h.v);
                // k_shade
                ShadeCtx cx;
                shadeBegin(hs.sc, td, 0, ra, rb, bt, hit4, cx);
                if (cx.add_L) L = L + cx.addL;
                for (uint32_t li = 0; li < hs.sc.n_lights; ++li) {
                    f3 wo, contrib;
                    const bool fixed = td.fixed_nee != 0u;
                    float t_max = TRT_INF;
                    if (!cx.shade_ok || !lightSample(hs.sc, cx.vx, *cx.m, li, cx.rng, wo, contrib, fixed, t_max)) continue;
                    const f3 w = cx.beta * contrib;
                    r_sh++;
                    // k_trace_shadow
                    const Hit sh = hs.nk ? traceClosestOct<OctArrayStack, ArrayStack, false>(hs.sc, rayOrigin(cx, wo), wo, ostk, stk, ni, nt, t_max, fixed, !fixed, &hs.light_boxes[li])
                                         : traceClosest<ArrayStack, false, 0>(hs.sc, rayOrigin(cx, wo), wo, stk, ni, nt, t_max, fixed, !fixed);
                    if (fixed ? sh.tri < 0 : (sh.tri >= 0 && (sh.flags >> 8) == (uint32_t)hs.sc.lights[li].mat)) L = L + w;
                }
                f4 nra, nrb, nbt;
                if (!shadeNext(cx, p->max_depth, nra, nrb, nbt)) break;
                ra = nra; rb = nrb; bt = nbt;
                r_ind++;
            }
            // k_resolve
            const float spp = (float)p->spp;
            acc[0] += (double)(L.x / spp);
            acc[1] += (double)(L.y / spp);
            acc[2] += (double)(L.z / spp);
        }
        out_rgb[(size_t)pl * 3 + 0] = (float)acc[0];
        out_rgb[(size_t)pl * 3 + 1] = (float)acc[1];
        out_rgb[(size_t)pl * 3 + 2] = (float)acc[2];
    }
    if (rays) { rays[0] = r_cam; rays[1] = r_sh; rays[2] = r_ind; }
    return 0;
}

extern "C" int hostsim_trace(const trt_scene* s, uint64_t n, const float* org, const float* dir, float* t, int32_t* tri, float* uv, uint64_t counts[2])
{
    if (!s || !org || !dir || !t || !tri) return 1;
    HostScene hs(s);
    uint64_t ci = 0, ct = 0;
#pragma omp parallel for schedule(static) reduction(+ : ci, ct)
    for (long long i = 0; i < (long long)n; ++i) {
        ArrayStack stk;
        OctArrayStack ostk;
        uint32_t ni = 0, nt = 0;
        const Hit h = hs.nk ? traceClosestOct<OctArrayStack, ArrayStack, true>(hs.sc, ld3(org + i * 3), ld3(dir + i * 3), ostk, stk, ni, nt)
                            : traceClosest<ArrayStack, true, 0>(hs.sc, ld3(org + i * 3), ld3(dir + i * 3), stk, ni, nt);
        t[i] = h.t;
        tri[i] = h.tri;
        if (uv) { uv[i * 2] = h.u; uv[i * 2 + 1] = h.v; }
        ci += ni;
        ct += nt;
    }
    if (counts) { counts[0] = ci; counts[1] = ct; }
    return 0;
}

// how many of the rays take the exact form behind the oct traversal (its result failed octResultCounts): tests/test_hostsim_parity.py
extern "C" uint64_t hostsim_oct_fallbacks(const trt_scene* s, uint64_t n, const float* org, const float* dir)
{
    const int old = g_node_kind;
    g_node_kind = 1;
    HostScene hs(s);
    g_node_kind = old;
    if (!hs.nk) return ~0ull;
    uint64_t bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
    for (long long i = 0; i < (long long)n; ++i) {
        OctArrayStack ostk;
        uint32_t ni = 0, nt = 0;
        const f3 o = ld3(org + i * 3), d = ld3(dir + i * 3);
        const Hit h = traceOctPass<OctArrayStack, false>(hs.sc, o, d, ostk, ni, nt, TRT_INF, false, false);
        bad += octResultCounts(hs.sc, h.t, h.tri, o, mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z)) ? 0u : 1u;
    }
    return bad;
}

// per-ray work of the closest-hit search on either node kind: visits[i], tests[i] (tools / tests: where do the long traversals come from?)
extern "C" int hostsim_trace_counts(const trt_scene* s, int nk, uint64_t n, const float* org, const float* dir, uint32_t* visits, uint32_t* tests)
{
    const int old = g_node_kind;
    g_node_kind = nk;
    HostScene hs(s);
    g_node_kind = old;
    if (nk && !hs.nk) return 1;
#pragma omp parallel for schedule(dynamic, 256)
    for (long long i = 0; i < (long long)n; ++i) {
        ArrayStack stk;
        OctArrayStack ostk;
        uint32_t ni = 0, nt = 0;
        const f3 o = ld3(org + i * 3), d = ld3(dir + i * 3);
        if (hs.nk) (void)traceOctPass<OctArrayStack, true>(hs.sc, o, d, ostk, ni, nt, TRT_INF, false, false);
        else (void)traceClosestPass<ArrayStack, true, 0, false>(hs.sc, o, d, stk, ni, nt);
        visits[i] = ni;
        tests[i] = nt;
    }
    return 0;
}

// The sequence of steps the wave driver takes for one ray on the oct nodes (tools/pool_sim.py: what would grouping rays by phase across the waves
// of a block buy?): per ray up to `cap` bytes, 0 = a node step, k = 1..2 = a leaf step that tests k triangles (the driver tests up to two of
// the lane's group per leaf step).  The loop is traceOctPass's (trt_oct.h) with a tape; `t_init` / `redo` / `light` as for a parity-mode shadow ray
// (light < 0: a closest-hit ray).
extern "C" int hostsim_oct_step_tape_chunk(const trt_scene* s, uint64_t n, const float* org, const float* dir, const float* t_init, int light, uint32_t cap, uint8_t* tape, uint32_t* len, uint32_t per_leaf_step)
{
    const int old = g_node_kind;
    g_node_kind = 1;
    HostScene hs(s);
    g_node_kind = old;
    if (!hs.nk) return 1;
    const LightBox* lbox = light >= 0 && (uint32_t)light < hs.light_boxes.size() ? &hs.light_boxes[(size_t)light] : nullptr;
#pragma omp parallel for schedule(dynamic, 256)
    for (long long i = 0; i < (long long)n; ++i) {
        OctArrayStack stk;
        const f3 o = ld3(org + i * 3), d = ld3(dir + i * 3);
        uint8_t* out = tape + (size_t)i * cap;
        uint32_t m = 0;
        auto put = [&](uint8_t v) { if (m < cap) out[m] = v; ++m; };
        const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        const OctRay R = makeOctRay(o, d, inv);
        float best_t = t_init ? t_init[i] : TRT_INF, stop_t = -TRT_INF;
        int32_t best_tri = -1;
        uint32_t best_flags = 0u;
        bool skip = false;
        if (lbox) {
            float e;
            if (!boxTest(lbox->lo[0], lbox->lo[1], lbox->lo[2], lbox->hi[0], lbox->hi[1], lbox->hi[2], o, inv, e)) skip = true;
            stop_t = trt_leaf_floor(e, hs.sc.leaf_alpha);
        }
        for (int pass = 0; pass < 2 && !skip; ++pass) {
            int sp = 0;
            OctGroup ng, tg;
            ng.x = 0u; ng.y = 0x80000000u;
            tg.x = 0u; tg.y = 0u;
            bool stop = false;
            for (;;) {
                if (ng.y & 0xFF000000u) {
                    const uint32_t ni = octNextChild(ng, R);
                    if (ng.y & 0xFF000000u) stk.push(sp++, ng);
                    put(0);
                    octVisit(hs.sc.onodes, ni, R, trt_cull_bound(best_t, hs.sc.leaf_alpha), ng, tg);
                }
                uint32_t in_step = 0;
                while (tg.y) {
                    const uint32_t b = (uint32_t)__builtin_ctz(tg.y);
                    tg.y &= tg.y - 1u;
                    const TriIsect T = hs.sc.tri_trav[tg.x + b];
                    float t, un, vn, det;
                    if (triTest(T, o, d, t, un, vn, det)) octFold(t, f2u(T.c.w), f2u(T.c.z), best_t, best_tri, best_flags);
                    if (++in_step == per_leaf_step || !tg.y) { put((uint8_t)in_step); in_step = 0; }
                    if (lbox && best_tri >= 0 && best_t < stop_t && in_step == 0u) { tg.y = 0u; stop = true; }
                }
                if (stop) break;
                if (!(ng.y & 0xFF000000u)) {
                    if (sp == 0) break;
                    ng = stk.pop(--sp);
                }
            }
            if (stop || !t_init || best_tri >= 0 || !(best_t < TRT_INF)) break;
            best_t = TRT_INF;  // nothing in front of the hint: search again without it
        }
        len[i] = m;
    }
    return 0;
}

extern "C" int hostsim_oct_step_tape(const trt_scene* s, uint64_t n, const float* org, const float* dir, const float* t_init, int light, uint32_t cap, uint8_t* tape, uint32_t* len)
{
    return hostsim_oct_step_tape_chunk(s, n, org, dir, t_init, light, cap, tape, len, 2u);  // TRT_OCT_LEAF_LOOP
}

// divMagic(n, d, magicOf(d)) against n / d for a list of numerators: returns the number of mismatches (tests/test_hostsim_parity.py)
extern "C" uint64_t hostsim_div_magic_mismatches(uint32_t d, const uint32_t* n, uint64_t count)
{
    const uint32_t m = magicOf(d);
    uint64_t bad = 0;
    for (uint64_t i = 0; i < count; ++i) bad += divMagic(n[i], d, m) != n[i] / d ? 1u : 0u;
    return bad;
}
