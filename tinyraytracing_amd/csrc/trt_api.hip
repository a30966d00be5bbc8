// trt_api.hip — C-ABI implementation (include/trt.h) for gfx950: scene upload,
// the wavefront render loop that replaces main.cpp:79-113, and the ray-batch
// traversal entry.  No CPU compute path exists in this library.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <limits>
#include <chrono>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "trt.h"
#include "trt_kernels.h"
#include "trt_wide.h"
#include "trt_oct_build.h"

using namespace trtd;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIPC(expr)                                                                                         \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail(e_ == hipErrorOutOfMemory ? TRT_ENOMEM : TRT_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

constexpr uint32_t MAX_TRACE_BLOCKS = 8192;                       // persistent grid cap of the traversal kernels
constexpr uint32_t SPILL_STRIDE = MAX_TRACE_BLOCKS * TRT_TRACE_BLOCK;
constexpr uint32_t MAX_BOUNCES = TRT_MAX_PATH_DEPTH + 2;
constexpr uint32_t COUNT_ROW = 16;                                // counters per bounce: [0] = queue length, [1+l] = shadow rays of light l
// Device layout of the counters: counter c of bounce b lives at d_counts[c * COUNT_STRIDE + b], so the
// counters k_shade bumps in one launch lie 16 KiB apart (different L2 channels: the atomic units work in
// parallel) instead of in one cache line.
constexpr uint32_t COUNT_STRIDE = (MAX_BOUNCES + 2 + 1023u) & ~1023u;
// Rows PAIR_ROW and PAIR_ROW + 1 hold, as one 64-bit word per bounce b, the two counters k_shade reserves with one atomic:
// low word = length of the queue of bounce b + 1, high word = shadow rays of the last light at bounce b.
constexpr uint32_t PAIR_ROW = COUNT_ROW - 2;
static_assert(1 + TRT_MAX_LIGHTS <= (int)PAIR_ROW, "counter rows overlap");
inline unsigned long long* pairCounter(uint32_t* d_counts, uint32_t b) { return reinterpret_cast<unsigned long long*>(d_counts + (size_t)PAIR_ROW * COUNT_STRIDE) + b; }
constexpr uint32_t MAX_BVH_DEPTH = 256;
// node kind of the persistent traversal kernels when the tree qualifies for both (DESIGN.md §4.1 has the A/B)
#ifndef TRT_DEFAULT_NODE_KIND
#define TRT_DEFAULT_NODE_KIND 1
#endif

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need)
    {
        if (need <= bytes) return TRT_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        HIPC(hipMalloc(&p, need));
        bytes = need;
        return TRT_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

}  // namespace

struct trt_handle {
    int device = 0;
    SceneDev sc{};
    std::vector<void*> scene_allocs;
    std::vector<uint32_t> light_mats;
    std::vector<LightBox> light_boxes;  // per light: the union of the boxes of the leaves that hold its triangles (trt_kernels.h LightBox)
    uint32_t depth = 0;       // stack entries a traversal can need (wide tree), + 1
    uint32_t bvh2_depth = 0;  // depth of the caller's BVH2
    uint32_t shade_tabs = 0;  // which k_shade<TABS> this scene runs
    uint32_t shade_pad_lds = 0;  // TRT_SHADE_PAD_LDS (probe): dynamic LDS bytes added to every k_shade launch — takes blocks off the CU to measure how the kernel's time hangs on its occupancy
    const void* lds_image = nullptr;  // the tables of shade_tabs, packed
    uint32_t lds_image_bytes = 0;
    uint32_t lds_tab[5] = {0, 0, 0, 0, 0};  // bytes of materials / lights / light CDF / light triangles / (tiny scenes) shading triangles that k_shade stages in LDS
    int trace_impl = 3;       // wave driver of the traversal kernels (0 wave-uniform walk of a tiny tree, 3 persistent waves + step scheduler)
    int node_kind = 0;        // what the persistent traversal kernels walk: 0 exact 128-B 4-wide nodes, 1 compressed 80-B 8-wide nodes (trt_oct.h)
    uint32_t oct_levels = 0;  // nodes on the longest root path of the oct tree
    // Grid of the traversal kernels for a queue of n rays.  A persistent wave refills finished lanes from its own slice
    // of the queue, which only pays when the slice holds several batches: aim for rays_per_wave rays per wave, but do
    // not go below the fill_blocks that fill the chip's wave slots, nor above one block per 256 rays.
    // (TRT_TRACE_RPW / TRT_TRACE_FILLB / TRT_TRACE_MAXB in the environment at trt_create: tuning.)  Per handle: no
    // process-wide mutable state in this library.
    uint32_t rays_per_wave = 256, fill_blocks = 2048, max_blocks = 8192;
    uint32_t traceGrid(uint32_t n) const
    {
        uint32_t b = (n + TRT_TRACE_BLOCK - 1) / TRT_TRACE_BLOCK;
        if (rays_per_wave > 64) {
            const uint32_t want = (uint32_t)(((uint64_t)n + 4ull * rays_per_wave - 1) / (4ull * rays_per_wave));
            b = std::min(b, std::max(fill_blocks, want));
        }
        b = std::min(std::max(b, 8u), std::min(max_blocks, 8192u));
        return (b + 7u) & ~7u;  // multiple of 8 for the XCD swizzle; <= MAX_TRACE_BLOCKS (8192) since that is one too
    }
    uint32_t tail_n = 131072;  // queue length at or below which k_tail finishes the pass (TRT_TAIL_N overrides)
    DevBuf arena, spill, small_buf, out_buf, io_buf;
    size_t spill_words_per_slot = 0;
    hipStream_t slot_streams[2] = {nullptr, nullptr};  // one per concurrent pass (trt_render_device)
    int n_slots = 2;                                   // TRT_SLOTS=1 in the environment vetoes TRT_FLAG_OVERLAP
    uint32_t* pinned_counts = nullptr;                 // host-pinned landing zone of the per-bounce queue lengths
    uint32_t slot_seq[2] = {0, 0};                     // last sequence number published per slot: monotonic over the handle's life, so a
                                                       // word left behind by an earlier (even a failed) call never equals an expected one
    int fail_at_bounce = -1;                           // TRT_TEST_FAIL_AT_BOUNCE at trt_create: the next render reports an injected failure
                                                       // after issuing that bounce (exercises the error path; consumed once)
    std::vector<hipEvent_t> events;
    ~trt_handle()
    {
        for (void* p : scene_allocs) (void)hipFree(p);
        arena.release();
        spill.release();
        small_buf.release();
        out_buf.release();
        io_buf.release();
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        for (hipStream_t st : slot_streams) if (st) (void)hipStreamDestroy(st);
        if (pinned_counts) (void)hipHostFree(pinned_counts);
    }
};

namespace {

template <class T>
int upload(trt_handle* h, const T* src, size_t count, const T** dst)
{
    void* p = nullptr;
    const size_t bytes = (std::max<size_t>(count, 1) * sizeof(T) + 15) & ~(size_t)15;  // readable in whole 16-B words
    HIPC(hipMalloc(&p, bytes));
    h->scene_allocs.push_back(p);
    if (count) HIPC(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
    *dst = (const T*)p;
    return TRT_OK;
}

// Structural check of the flat BVH and its true depth (the kernels size their
// stacks from it): every child reference in range, every leaf range in range,
// no node reachable twice (no cycles / DAGs), and every triangle under child0 has
// a lower index than every triangle under child1 — the post-BVH order of the
// reference (bvh.cpp sorts the triangle array in place and recurses on its two halves),
// which is what lets the between-leaves tie rule "r1 if r1 emissive else r2" (bvh.cpp:168-172)
// be applied by triangle index in any visiting order.
int validateBvh(const trt_scene* s, uint32_t* depth_out, unsigned threads = 1)
{
    // One walk per subtree of a cut of the tree (the cut itself, a few hundred nodes, is walked here first), subtrees side by side on
    // `threads` host threads: marks are atomic bytes, so a node or triangle that two walks reach is caught whichever gets there second.
    const uint32_t nn = s->n_nodes;
    std::unique_ptr<std::atomic<uint8_t>[]> seen(new std::atomic<uint8_t>[nn]);
    std::unique_ptr<std::atomic<uint8_t>[]> tri_seen(new std::atomic<uint8_t>[std::max<uint32_t>(s->n_tris, 1u)]);
    par::forRange(nn, threads, 1u << 20, [&](size_t b, size_t e) { for (size_t i = b; i < e; ++i) seen[i].store(0, std::memory_order_relaxed); });
    par::forRange(s->n_tris, threads, 1u << 20, [&](size_t b, size_t e) { for (size_t i = b; i < e; ++i) tri_seen[i].store(0, std::memory_order_relaxed); });
    // (min, max) triangle index under every inner node: index order of siblings, children before parents
    std::unique_ptr<uint32_t[]> lo(new uint32_t[nn]), hi(new uint32_t[nn]);
    std::unique_ptr<uint8_t[]> has(new uint8_t[nn]);
    // what one node says about itself: its own checks, its inner children appended to `kids`; nullptr or the complaint
    auto visit = [&](uint32_t ni, uint32_t dep, uint32_t* kids, int& n_kids) -> const char* {
        n_kids = 0;
        if (ni >= nn) return "bvh: child index out of range";
        if (seen[ni].exchange(1, std::memory_order_relaxed)) return "bvh: node reachable twice";
        if (dep > MAX_BVH_DEPTH) return "bvh: deeper than 256 levels";
        // A NaN box coordinate is refused: glm::min / glm::max (bvh.cpp:238-242) let a NaN through from their FIRST operand only, so what such a box does to a ray
        // depends on the axis it sits on — a behaviour nobody builds on, and one the kernels' v_min / v_max (which drop a NaN from either side) would have to pay
        // two instructions per slab to mimic.  +-inf and every finite value, nested or not, are fine (tests: poisoned geometry; boxes that do not nest).
        {
            const trt_bvh_node& nd = s->nodes[ni];
            for (int a = 0; a < 3; ++a)
                if (nd.lo0[a] != nd.lo0[a] || nd.hi0[a] != nd.hi0[a] || nd.lo1[a] != nd.lo1[a] || nd.hi1[a] != nd.hi1[a]) return "bvh: a box coordinate is NaN";
        }
        const uint32_t ch[2] = {s->nodes[ni].child0, s->nodes[ni].child1};
        for (uint32_t c : ch) {
            if (c & TRT_LEAF_BIT) {
                const uint32_t first = TRT_LEAF_FIRST(c), count = TRT_LEAF_COUNT(c);
                if ((uint64_t)first + count > s->n_tris) return "bvh: leaf range out of bounds";
                for (uint32_t i = first; i < first + count; ++i)
                    if (tri_seen[i].exchange(1, std::memory_order_relaxed)) return "bvh: triangle in two leaves";
            } else {
                kids[n_kids++] = c;
            }
        }
        return nullptr;
    };
    auto order_rule = [&](uint32_t n) -> const char* {  // children already done
        uint32_t clo[2], chi[2];
        bool chas[2];
        const uint32_t ch[2] = {s->nodes[n].child0, s->nodes[n].child1};
        for (int c = 0; c < 2; ++c) {
            if (ch[c] & TRT_LEAF_BIT) {
                const uint32_t first = TRT_LEAF_FIRST(ch[c]), count = TRT_LEAF_COUNT(ch[c]);
                chas[c] = count != 0;
                clo[c] = first;
                chi[c] = first + count - (count ? 1u : 0u);
            } else {
                chas[c] = has[ch[c]] != 0;
                clo[c] = lo[ch[c]];
                chi[c] = hi[ch[c]];
            }
        }
        if (chas[0] && chas[1] && !(chi[0] < clo[1])) return "bvh: triangles under child0 must precede those under child1 (post-BVH order)";
        has[n] = chas[0] || chas[1];
        lo[n] = std::min(chas[0] ? clo[0] : 0xFFFFFFFFu, chas[1] ? clo[1] : 0xFFFFFFFFu);
        hi[n] = std::max(chas[0] ? chi[0] : 0u, chas[1] ? chi[1] : 0u);
        return nullptr;
    };
    struct Ref { uint32_t node, depth; };
    std::vector<Ref> top{{0u, 1u}};  // the cut: breadth first, parents before children
    size_t head = 0;
    uint32_t max_depth = 0;
    const size_t want = threads > 1 ? (size_t)threads * 16 : 1;
    while (threads > 1 && head < top.size() && top.size() - head < want) {
        const Ref r = top[head++];
        uint32_t kids[2];
        int nk;
        if (const char* why = visit(r.node, r.depth, kids, nk)) return fail(TRT_EINVAL, why);
        max_depth = std::max(max_depth, r.depth);
        for (int k = 0; k < nk; ++k) top.push_back({kids[k], r.depth + 1});
    }
    const size_t n_roots = top.size() - head;
    std::vector<const char*> why_of(n_roots, nullptr);
    std::vector<uint32_t> depth_of(n_roots, 0);
    par::forTasks(n_roots, threads, [&](size_t ti) {
        std::vector<Ref> st{top[head + ti]};
        std::vector<uint32_t> order;
        while (!st.empty()) {
            const Ref r = st.back();
            st.pop_back();
            uint32_t kids[2];
            int nk;
            if (const char* why = visit(r.node, r.depth, kids, nk)) { why_of[ti] = why; return; }
            order.push_back(r.node);
            depth_of[ti] = std::max(depth_of[ti], r.depth);
            for (int k = 0; k < nk; ++k) st.push_back({kids[k], r.depth + 1});
        }
        for (size_t k = order.size(); k-- > 0;)
            if (const char* why = order_rule(order[k])) { why_of[ti] = why; return; }
    });
    for (size_t ti = 0; ti < n_roots; ++ti) {
        if (why_of[ti]) return fail(TRT_EINVAL, why_of[ti]);
        max_depth = std::max(max_depth, depth_of[ti]);
    }
    for (size_t k = head; k-- > 0;)
        if (const char* why = order_rule(top[k].node)) return fail(TRT_EINVAL, why);
    *depth_out = max_depth;
    return TRT_OK;
}

int checkParams(const trt_handle* h, const trt_params* p)
{
    if (!h || !p) return fail(TRT_EINVAL, "null handle/params");
    if (p->width < 2 || p->height < 2) return fail(TRT_EINVAL, "width and height must be >= 2 (x = j/(W-1), main.cpp:88)");
    if (p->spp < 1) return fail(TRT_EINVAL, "spp must be >= 1");
    if (p->x0 < 0 || p->y0 < 0 || p->x1 > p->width || p->y1 > p->height || p->x0 >= p->x1 || p->y0 >= p->y1) return fail(TRT_EINVAL, "tile rectangle outside the image or empty");
    if (p->row_mod > 1 && (p->row_block < 1 || p->row_rem < 0 || p->row_rem >= p->row_mod)) return fail(TRT_EINVAL, "bad row interleave");
    if (p->max_depth < 0) return fail(TRT_EINVAL, "max_depth must be >= 0");
    if ((uint64_t)p->width * (uint64_t)p->height > 0xFFFFFFFFull) return fail(TRT_EINVAL, "image too large");
    return TRT_OK;
}

bool rowSelected(const trt_params* p, int y) { return p->row_mod <= 1 || ((y / p->row_block) % p->row_mod) == p->row_rem; }

uint32_t tailGrid(uint32_t n)
{
    uint32_t b = (n + TRT_TRACE_BLOCK - 1) / TRT_TRACE_BLOCK;
    b = std::min(std::max(b, 8u), MAX_TRACE_BLOCKS);
    return (b + 7u) & ~7u;
}
// Traversal kernels: the LDS stack holds 8 or 16 levels without spill code when the scene's verified BVH
// depth fits, else 16 levels + a global spill area (16 KiB per block keeps 8 waves per SIMD resident);
// the wave driver (trt_kernels.h) is the static one for shallow trees, the scheduler one otherwise.
template <bool COUNT, bool PRIMARY>
void launchTraceClosest(const trt_handle* h, hipStream_t stream, uint32_t* spill, const RaySource& src, f4* hit, uint32_t n, DeviceStats* d_stats, RedoList redo);
template <bool COUNT>
void launchTraceShadow(const trt_handle* h, hipStream_t stream, uint32_t* spill, const ShadowQueue& sq, uint32_t n, uint32_t light_mat, f4* Lacc, DeviceStats* d_stats, uint32_t any, RedoList redo, const LightBox& lbox);

struct Timer {
    trt_handle* h;
    bool on;
    size_t used = 0;
    struct Span { int k; size_t e0, e1; };
    std::vector<Span> spans;
    static constexpr size_t RESERVED = 3;  // render begin / end / resolve chain
    hipEvent_t get(size_t i)
    {
        while (h->events.size() <= i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            h->events.push_back(e);
        }
        return h->events[i];
    }
    // per-kernel spans only with TRT_FLAG_TIMING; recorded on the stream the kernel is launched on
    void begin(int k, hipStream_t stream)
    {
        if (!on) return;
        hipEvent_t e = get(RESERVED + used);
        if (e) (void)hipEventRecord(e, stream);
        spans.push_back({k, RESERVED + used, 0});
        used++;
    }
    void end(hipStream_t stream)
    {
        if (!on) return;
        hipEvent_t e = get(RESERVED + used);
        if (e) (void)hipEventRecord(e, stream);
        spans.back().e1 = RESERVED + used;
        used++;
    }
};

// One kernel per (driver, LDS depth, node kind); the scene picks the combination once, in trt_create.
#define TRT_LAUNCH_CLOSEST(DEPTH, SPILL, IMPL, NK) \
    hipLaunchKernelGGL((k_trace_closest<COUNT, DEPTH, SPILL, IMPL, PRIMARY, NK>), g, b, 0, stream, h->sc, src, hit, n, spill, SPILL_STRIDE, d_stats, redo)
#define TRT_LAUNCH_SHADOW(DEPTH, SPILL, IMPL, NK) \
    hipLaunchKernelGGL((k_trace_shadow<COUNT, DEPTH, SPILL, IMPL, NK>), g, b, 0, stream, h->sc, sq, n, light_mat, Lacc, spill, SPILL_STRIDE, d_stats, any, redo, lbox)
#define TRT_BY_DEPTH(LAUNCH, IMPL, NK)                                       \
    do {                                                                     \
        if (h->depth <= 16) LAUNCH(16, false, IMPL, NK);                     \
        else LAUNCH(TRT_LDS_STACK_MAX, true, IMPL, NK);                      \
    } while (0)
// The oct tree needs one 8-byte entry per level below the root: OCT_LDS_LEVELS of them in LDS (20 KiB per block: eight blocks per CU),
// deeper ones — staircase has 10 levels, the 10 M-triangle mesh 11 — in the global spill area.  One instantiation serves every tree.
constexpr uint32_t OCT_LDS_LEVELS = 10;
#define TRT_BY_OCT_DEPTH(LAUNCH) LAUNCH(10, true, 3, 1)
// Behind every traversal launch of a per-lane driver: k_trace_fix (a few blocks) traces the rays of the launch's redo list again in the
// exact form (trt_kernels.h, RedoList).  The wave-uniform walk applies the rule on the spot and has no list.
template <bool COUNT, bool PRIMARY>
void launchTraceClosest(const trt_handle* h, hipStream_t stream, uint32_t* spill, const RaySource& src, f4* hit, uint32_t n, DeviceStats* d_stats, RedoList redo)
{
    const dim3 g(h->traceGrid(n)), b(TRT_TRACE_BLOCK);
    if (h->trace_impl == 0) { TRT_LAUNCH_CLOSEST(1, false, 0, 0); return; }
    if (h->node_kind == 1) TRT_BY_OCT_DEPTH(TRT_LAUNCH_CLOSEST);
    else TRT_BY_DEPTH(TRT_LAUNCH_CLOSEST, 3, 0);
    hipLaunchKernelGGL((k_trace_fix<false, PRIMARY, 0>), dim3(TRT_FIX_BLOCKS), b, 0, stream, h->sc, src, hit, (const f4*)nullptr, 0u, (f4*)nullptr, spill, SPILL_STRIDE, redo, 0u, d_stats);
}

template <bool COUNT>
void launchTraceShadow(const trt_handle* h, hipStream_t stream, uint32_t* spill, const ShadowQueue& sq, uint32_t n, uint32_t light_mat, f4* Lacc, DeviceStats* d_stats, uint32_t any, RedoList redo, const LightBox& lbox)
{
    const dim3 g(h->traceGrid(n)), b(TRT_TRACE_BLOCK);
    if (h->trace_impl == 0) { TRT_LAUNCH_SHADOW(1, false, 0, 0); return; }
    if (h->node_kind == 1) TRT_BY_OCT_DEPTH(TRT_LAUNCH_SHADOW);
    else TRT_BY_DEPTH(TRT_LAUNCH_SHADOW, 3, 0);
    RaySource src;
    src.ra = sq.sa;
    src.rb = sq.sb;
    src.s0 = 0;
    hipLaunchKernelGGL((k_trace_fix<true, false, 0>), dim3(TRT_FIX_BLOCKS), b, 0, stream, h->sc, src, (f4*)nullptr, (const f4*)sq.sw, light_mat, Lacc, spill, SPILL_STRIDE, redo, any, d_stats);
}

}  // namespace

namespace {
// Everything trt_create derives from the caller's scene on the host — checks, triangle records, the 4-wide and 8-wide collapses, the
// leaf and light boxes, the tuning read from the environment — independent of the device: built once (host threads: trt_wide.h `par`,
// TRT_HOST_THREADS), uploaded to every device of a group (trt_group_create).
struct SceneImage {
    const trt_scene* s = nullptr;
    unsigned threads = 1;
    bool dbg = false;
    uint32_t bvh2_depth = 0;
    int trace_impl = 3;
    int node_kind = 0;
    std::vector<TriIsect> isect;
    std::vector<TriShade> shade;
    WideTree wide;
    OctTree oct;
    std::vector<f4> leaf_boxes;
    std::vector<LightBox> light_boxes;
    std::vector<MaterialDev> mats;
    std::vector<LightDev> lights;
    std::vector<LightTriDev> ltris;
    std::vector<float> cum;
    bool cum_monotone = true;
    std::vector<TextureDev> tex;
    std::vector<uint8_t> tex_bytes;
    float leaf_alpha = 0.0f;
    float cull_alpha = 0.0f;  // leaf_alpha, or +inf for a tree whose boxes do not nest (SceneDev::cull_alpha)
    std::vector<uint32_t> plane_bits;  // Bloom filter over the box planes (trt_path.h planeMaybe)
    uint32_t plane_lg = 0;
    bool nested = true;
};

struct Lap {  // TRT_DEBUG: where the start-up time of a big scene goes (tools/create_cost.py)
    bool on;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void operator()(const char* what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "trt_create: %-32s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

int buildSceneImage(const trt_scene* s, SceneImage& im)
{
    if (s->n_nodes < 1 || !s->nodes) return fail(TRT_EINVAL, "scene needs at least the root node");
    if (s->n_tris > TRT_MAX_TRIS) return fail(TRT_EINVAL, "too many triangles");
    if (s->n_tris && (!s->tri_v || !s->tri_vn || !s->tri_vt || !s->tri_mat)) return fail(TRT_EINVAL, "null triangle arrays");
    if (s->n_lights > (uint32_t)TRT_MAX_LIGHTS) return fail(TRT_EINVAL, "more than 8 lights");
    if (s->n_materials < 1 || !s->materials) return fail(TRT_EINVAL, "scene needs materials");
    if (s->n_materials >= (1u << 24)) return fail(TRT_EINVAL, "too many materials");
    for (uint32_t i = 0; i < s->n_tris; ++i)
        if (s->tri_mat[i] < 0 || (uint32_t)s->tri_mat[i] >= s->n_materials) return fail(TRT_EINVAL, "triangle material id out of range");
    for (uint32_t i = 0; i < s->n_materials; ++i)
        if (s->materials[i].tex >= (int32_t)s->n_textures) return fail(TRT_EINVAL, "material texture id out of range");
    for (uint32_t i = 0; i < s->n_lights; ++i) {
        const trt_light& L = s->lights[i];
        if (L.mat < 0 || (uint32_t)L.mat >= s->n_materials) return fail(TRT_EINVAL, "light material id out of range");
        if ((uint64_t)L.tri_first + L.tri_count > s->n_light_tris) return fail(TRT_EINVAL, "light triangle range out of bounds");
    }
    for (uint32_t i = 0; i < s->n_textures; ++i)
        if (s->textures[i].width < 1 || s->textures[i].height < 1 || !s->textures[i].rgb) return fail(TRT_EINVAL, "empty texture");
    im.s = s;
    im.dbg = std::getenv("TRT_DEBUG") != nullptr;
    im.threads = par::defaultThreads();
    if (const char* e = std::getenv("TRT_HOST_THREADS")) im.threads = (unsigned)std::min(256, std::max(1, std::atoi(e)));
    Lap lap{im.dbg};
    if (int e = validateBvh(s, &im.bvh2_depth, im.threads)) return e;
    lap("checks, validateBvh");

    // the wave-uniform walk needs a 32-bit reach mask, and it evaluates the nodes in index order: every inner child must
    // come after its parent (all builders here emit parents first; a caller's tree that does not is walked per lane)
    bool tiny = s->n_nodes <= 32 && s->n_tris <= 64;
    for (uint32_t n = 0; tiny && n < s->n_nodes; ++n) {
        const uint32_t ch[2] = {s->nodes[n].child0, s->nodes[n].child1};
        for (uint32_t c : ch)
            if (!(c & TRT_LEAF_BIT) && c <= n) tiny = false;
    }
    im.trace_impl = tiny ? 0 : 3;
    if (const char* e = std::getenv("TRT_TRACE_IMPL")) { if (std::atoi(e) == 3) im.trace_impl = 3; }  // tests: the per-lane driver on a tiny tree too

    // 48-B intersection records and 64-B shading records
    im.isect.resize(s->n_tris);
    im.shade.resize(s->n_tris);
    par::forRange(s->n_tris, im.threads, 65536, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; ++i) {
            const int32_t mat = s->tri_mat[i];
            im.isect[i] = makeTriIsect(s->tri_v + i * 9, mat, s->materials[mat].is_emissive != 0);
            std::memcpy(im.shade[i].vn, s->tri_vn + i * 9, sizeof(float) * 9);
            std::memcpy(im.shade[i].vt, s->tri_vt + i * 6, sizeof(float) * 6);
            im.shade[i].mat = mat;
        }
    });
    lap("triangle records");
    {   // the 4-wide collapse every per-lane traversal can walk; its stack bound sizes the LDS stack / the spill area
        bool greedy = s->n_tris > 4000000u;  // measured: trt_wide.h
        if (const char* e = std::getenv("TRT_WIDE_GREEDY")) greedy = std::atoi(e) != 0;
        im.wide = greedy ? collapseBvhGreedy(s->nodes, s->n_nodes, im.threads) : collapseBvh(s->nodes, s->n_nodes, im.threads);
        lap("4-wide collapse");
    }
    // the caller's box of every leaf: a hit in front of its own leaf's box does not count (leafEntry(), trt_path.h)
    im.leaf_boxes = leafBoxesOf(s->nodes, s->n_nodes, s->n_tris, im.threads);
    im.light_boxes = lightBoxesOf(im.leaf_boxes, s->tri_mat, s->n_tris, s->lights, s->n_lights);
    if (const char* e = std::getenv("TRT_SHADOW_STOP"))  // 0: all of space as every light's box, i.e. no early end of a shadow ray (A/B)
        if (std::atoi(e) == 0) im.light_boxes.assign(s->n_lights, LightBox{{-3.0e38f, -3.0e38f, -3.0e38f}, {3.0e38f, 3.0e38f, 3.0e38f}});
    im.leaf_alpha = sceneLeafAlpha(s->nodes, s->n_nodes);
    // Culling by distance rests on nested boxes; a foreign tree that breaks the premise is walked without it (every box the ray passes is entered, bvh.cpp:162-166)
    im.nested = wide_detail::boxesNested(s->nodes, s->n_nodes, im.threads);
    im.cull_alpha = im.nested ? im.leaf_alpha : std::numeric_limits<float>::infinity();
    im.plane_lg = wide_detail::planeFilterBuild(s->nodes, s->n_nodes, im.plane_bits, im.threads);
    if (im.dbg && !im.nested) std::fprintf(stderr, "[trt] the boxes of this tree do not nest: traversal without distance culling\n");
    lap("leaf boxes, light boxes");
    // The 8-wide compressed nodes (trt_oct.h) for the persistent traversal kernels, when the tree qualifies (nested, finite; larger
    // leaves are laid out as several slots with the leaf's own box): TRT_NODE_KIND=0/1 in the environment forces either kind (A/B runs, tests).
    bool want_oct = im.trace_impl != 0 && TRT_DEFAULT_NODE_KIND == 1;
    if (const char* e = std::getenv("TRT_NODE_KIND")) want_oct = im.trace_impl != 0 && std::atoi(e) == 1;
    if (want_oct) {
        im.oct = buildOct(s->nodes, s->n_nodes, s->n_tris, im.isect.data(), im.threads);
        im.node_kind = im.oct.ok ? 1 : 0;
        lap("8-wide collapse");
        if (im.dbg) std::fprintf(stderr, "trt_create: oct tree %s (%s): %zu nodes, %u levels, %u leaves of more than 3 triangles split\n", im.oct.ok ? "built" : "not built", im.oct.why, im.oct.nodes.size(), im.oct.levels, im.oct.split_leaves);
    }
    if (im.dbg) std::fprintf(stderr, "trt_create: %zu wide nodes, node kind %d, stack need %u, %u host threads\n", im.wide.nodes.size(), im.node_kind, im.wide.stack_need, im.threads);

    im.mats.resize(s->n_materials);
    for (uint32_t i = 0; i < s->n_materials; ++i) im.mats[i] = makeMaterialDev(s->materials[i]);
    im.lights.resize(s->n_lights);
    for (uint32_t i = 0; i < s->n_lights; ++i) im.lights[i] = makeLightDev(s->lights[i], s->materials);
    im.ltris.resize(s->n_light_tris);
    for (uint32_t i = 0; i < s->n_light_tris; ++i) im.ltris[i] = makeLightTriDev(s->light_tris[i]);
    // packed CDF for the bisection in lightSample; only when every light's CDF is non-decreasing and NaN-free
    im.cum.resize(s->n_light_tris);
    for (uint32_t k = 0; k < s->n_light_tris; ++k) im.cum[k] = s->light_tris[k].cum_area;
    for (uint32_t l = 0; l < s->n_lights; ++l)
        for (uint32_t k = 0; k < s->lights[l].tri_count; ++k) {
            const float c = im.cum[s->lights[l].tri_first + k];
            if (!(c == c) || (k && c < im.cum[s->lights[l].tri_first + k - 1])) im.cum_monotone = false;
        }
    im.tex.resize(s->n_textures);
    for (uint32_t i = 0; i < s->n_textures; ++i) {
        im.tex[i].width = s->textures[i].width;
        im.tex[i].height = s->textures[i].height;
        im.tex[i].offset = im.tex_bytes.size();
        const size_t nb = (size_t)im.tex[i].width * im.tex[i].height * 3;
        im.tex_bytes.insert(im.tex_bytes.end(), s->textures[i].rgb, s->textures[i].rgb + nb);
    }
    return TRT_OK;
}

// The device half of trt_create: a handle on `device` from the host image (thread-safe against other devices' calls: touches only
// the handle, the image read-only and this thread's HIP device).
int createOnDevice(const SceneImage& im, int device, trt_handle** out)
{
    const trt_scene* s = im.s;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(TRT_ENODEV, "no HIP device");
    if (device < 0 || device >= ndev) return fail(TRT_ENODEV, "device ordinal out of range");
    HIPC(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPC(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(TRT_ENODEV, std::string("this library is built for gfx950 only, device is ") + prop.gcnArchName);

    Lap lap{im.dbg};
    std::unique_ptr<trt_handle> h(new trt_handle);
    h->device = device;
    h->bvh2_depth = im.bvh2_depth;
    if (const char* e = std::getenv("TRT_TAIL_N")) h->tail_n = (uint32_t)std::strtoul(e, nullptr, 10);
    h->sc.refill_min = s->n_tris <= 200000u ? 48u : 32u;
    // a triangle step costs about half a node step: deep trees run it already when 2 lanes at nodes face 3 at leaves
    // (blob-2M, soup-1M -2.6 % per step; the cg22 scenes are 0.7 % better off with the plain majority)
    h->sc.sched_in_w = s->n_tris <= 200000u ? 1u : 2u;
    h->sc.sched_lf_w = s->n_tris <= 200000u ? 1u : 3u;
    if (const char* e = std::getenv("TRT_SCHED_W")) {  // "in:leaf" weights of the scheduler driver (tuning)
        unsigned a = 0, b = 0;
        if (std::sscanf(e, "%u:%u", &a, &b) == 2 && a > 0 && b > 0 && a < 1024 && b < 1024) { h->sc.sched_in_w = a; h->sc.sched_lf_w = b; }
    }
    // leaf steps of the oct driver: two triangles each on trees with leaves of <= 3; a tree whose leaves were split into several slots
    // (the reference's leaf size 8) brings up to 24 triangles per node, all of which have to be tested: four per step there
    // (profiles/r04_leaf8.txt has the sweep)
    h->sc.leaf_loop = im.oct.split_leaves * 8u > s->n_tris / 8u ? 4u : 2u;  // at least an eighth of the triangles in such leaves (8 per leaf assumed)
    if (const char* e = std::getenv("TRT_LEAF_LOOP")) h->sc.leaf_loop = std::min(24u, std::max(1u, (uint32_t)std::strtoul(e, nullptr, 10)));
    if (const char* e = std::getenv("TRT_REFILL_MIN")) h->sc.refill_min = std::min(64u, std::max(1u, (uint32_t)std::strtoul(e, nullptr, 10)));
    if (im.dbg) std::fprintf(stderr, "trt_create: refill_min %u tail_n %u\n", h->sc.refill_min, h->tail_n);
    if (const char* e = std::getenv("TRT_TRACE_RPW")) h->rays_per_wave = (uint32_t)std::strtoul(e, nullptr, 10);
    if (const char* e = std::getenv("TRT_TRACE_FILLB")) h->fill_blocks = (uint32_t)std::strtoul(e, nullptr, 10);
    if (const char* e = std::getenv("TRT_TRACE_MAXB")) h->max_blocks = std::max(8u, (uint32_t)std::strtoul(e, nullptr, 10));
    h->trace_impl = im.trace_impl;
    h->node_kind = im.node_kind;
    h->oct_levels = im.oct.levels;
    h->depth = im.wide.stack_need + 1;
    h->light_boxes = im.light_boxes;

    if (int e = upload(h.get(), im.isect.data(), im.isect.size(), &h->sc.tri_isect)) return e;
    if (int e = upload(h.get(), im.plane_bits.data(), im.plane_bits.size(), &h->sc.plane_bits)) return e;
    h->sc.plane_shift = 32u - im.plane_lg;
    if (int e = upload(h.get(), im.shade.data(), im.shade.size(), &h->sc.tri_shade)) return e;
    if (int e = upload(h.get(), s->nodes, (size_t)s->n_nodes, &h->sc.nodes)) return e;
    h->sc.n_wnodes = (uint32_t)im.wide.nodes.size();
    if (int e = upload(h.get(), im.leaf_boxes.data(), im.leaf_boxes.size(), &h->sc.leaf_box)) return e;
    if (int e = upload(h.get(), im.wide.nodes.data(), im.wide.nodes.size(), &h->sc.wnodes)) return e;
    h->sc.onodes = nullptr;
    h->sc.tri_trav = nullptr;
    if (im.node_kind == 1) {
        h->sc.n_onodes = (uint32_t)im.oct.nodes.size();
        if (int e = upload(h.get(), im.oct.nodes.data(), im.oct.nodes.size(), &h->sc.onodes)) return e;
        if (int e = upload(h.get(), im.oct.tri_trav.data(), im.oct.tri_trav.size(), &h->sc.tri_trav)) return e;
    }
    if (int e = upload(h.get(), im.mats.data(), im.mats.size(), &h->sc.materials)) return e;
    if (int e = upload(h.get(), im.lights.data(), im.lights.size(), &h->sc.lights)) return e;
    if (int e = upload(h.get(), im.ltris.data(), im.ltris.size(), &h->sc.light_tris)) return e;
    h->sc.light_cum = nullptr;
    if (im.cum_monotone)
        if (int e = upload(h.get(), im.cum.data(), im.cum.size(), &h->sc.light_cum)) return e;
    if (int e = upload(h.get(), im.tex.data(), im.tex.size(), &h->sc.textures)) return e;
    if (int e = upload(h.get(), im.tex_bytes.data(), im.tex_bytes.size(), &h->sc.tex_bytes)) return e;
    lap("uploads");
    h->sc.n_tris = s->n_tris;
    h->sc.n_nodes = s->n_nodes;
    h->sc.n_lights = s->n_lights;
    h->sc.light0_area = s->n_lights ? s->lights[0].area : 0.0f;
    h->sc.leaf_alpha = im.leaf_alpha;
    h->sc.cull_alpha = im.cull_alpha;
    h->sc.cam = s->camera;
    for (uint32_t i = 0; i < s->n_lights; ++i) h->light_mats.push_back((uint32_t)s->lights[i].mat);
    {   // which small tables k_shade copies into LDS: in this order while they fit (uploads are padded to 16 B)
        // the per-triangle shading records only for scenes of a few dozen triangles (every block pays for the copy)
        const uint32_t want[5] = {(uint32_t)(s->n_materials * sizeof(MaterialDev)), (uint32_t)(s->n_lights * sizeof(LightDev)),
                                  h->sc.light_cum ? (uint32_t)(s->n_light_tris * sizeof(float)) : 0u, (uint32_t)(s->n_light_tris * sizeof(LightTriDev)),
                                  s->n_tris <= 64 ? (uint32_t)(s->n_tris * sizeof(TriShade)) : 0u};
        uint32_t used = 0;
        for (int k = 0; k < 5; ++k) {
            const uint32_t padded = (want[k] + 15u) & ~15u;
            if (want[k] && used + padded <= TRT_SHADE_LDS_TABLE_BYTES) { h->lds_tab[k] = want[k]; used += padded; }
        }
        if (std::getenv("TRT_SHADE_NO_LDS")) h->lds_tab[0] = h->lds_tab[1] = h->lds_tab[2] = h->lds_tab[3] = h->lds_tab[4] = 0;
        if (std::getenv("TRT_SHADE_NO_LDS_TRIS")) h->lds_tab[4] = 0;
        // k_shade is instantiated for these sets of staged tables; take the largest one that fits
        uint32_t have = 0;
        for (int k = 0; k < 5; ++k) have |= h->lds_tab[k] ? (1u << k) : 0u;
        h->shade_tabs = 0;
        for (uint32_t m : {31u, 15u, 7u, 3u})
            if ((have & m) == m) { h->shade_tabs = m; break; }
        // the staged tables once more, packed in LDS layout: a block fetches them with one coalesced pass
        const void* src[5] = {h->sc.materials, h->sc.lights, h->sc.light_cum, h->sc.light_tris, h->sc.tri_shade};
        size_t total = 0;
        for (int k = 0; k < 5; ++k)
            if (h->shade_tabs >> k & 1u) total += (h->lds_tab[k] + 15u) & ~15u;
        if (total) {
            void* img = nullptr;
            HIPC(hipMalloc(&img, total));
            h->scene_allocs.push_back(img);
            size_t off = 0;
            for (int k = 0; k < 5; ++k)
                if (h->shade_tabs >> k & 1u) {
                    const size_t padded = (h->lds_tab[k] + 15u) & ~15u;  // upload() padded the source the same way
                    HIPC(hipMemcpy((char*)img + off, src[k], padded, hipMemcpyDeviceToDevice));
                    off += padded;
                }
            h->lds_image = img;
            h->lds_image_bytes = (uint32_t)total;
        }
    }

    // traversal spill area: levels beyond the LDS stack, for the largest grid
    // (k_trace_fix / k_tail walk the caller's BVH2 itself for the rays of raySpecial(), trt_path.h: one entry per level of it)
    const uint32_t stack_levels = std::max(h->depth, h->bvh2_depth + 2u);
    uint32_t spill_levels = stack_levels > (uint32_t)TRT_LDS_STACK_MAX ? stack_levels - TRT_LDS_STACK_MAX + 1 : 1;
    if (h->node_kind == 1 && h->oct_levels > OCT_LDS_LEVELS) spill_levels = std::max(spill_levels, 2u * (h->oct_levels - OCT_LDS_LEVELS + 1));  // two words per level
    h->spill_words_per_slot = (size_t)spill_levels * SPILL_STRIDE;
    if (int e = h->spill.ensure(h->spill_words_per_slot * 2 * sizeof(uint32_t))) return e;  // one area per concurrent pass
    for (hipStream_t& st : h->slot_streams) HIPC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HIPC(hipHostMalloc((void**)&h->pinned_counts, 2 * (2 * COUNT_ROW + 16) * sizeof(uint32_t), hipHostMallocDefault));  // per slot: counters + sequence word
    std::memset(h->pinned_counts, 0, 2 * (2 * COUNT_ROW + 16) * sizeof(uint32_t));  // sequence words start at 0; the first one asked for is 1
    if (const char* e = std::getenv("TRT_SLOTS")) h->n_slots = std::atoi(e) >= 2 ? 2 : 1;
    if (const char* e = std::getenv("TRT_TEST_FAIL_AT_BOUNCE")) h->fail_at_bounce = std::atoi(e);
    if (const char* e = std::getenv("TRT_SHADE_PAD_LDS")) {
        h->shade_pad_lds = std::min(100000u, (uint32_t)std::strtoul(e, nullptr, 10));
        // more than 64 KiB per block needs the opt-in; a refusal shows as a launch error, not as a silent no-op
#define TRT_PAD_ATTR(T) \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shade<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->shade_pad_lds); \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shade<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->shade_pad_lds);
        TRT_PAD_ATTR(31u) TRT_PAD_ATTR(15u) TRT_PAD_ATTR(7u) TRT_PAD_ATTR(3u) TRT_PAD_ATTR(0u)
#undef TRT_PAD_ATTR
    }

    *out = h.release();
    return TRT_OK;
}
}  // namespace

extern "C" {

const char* trt_last_error(void) { return g_err.c_str(); }
int trt_abi_version(void) { return TRT_ABI_VERSION; }

int trt_rows_selected(const trt_params* p)
{
    if (!p) return -1;
    int n = 0;
    for (int y = p->y0; y < p->y1; ++y) n += rowSelected(p, y) ? 1 : 0;
    return n;
}

int trt_create(const trt_scene* s, int device, trt_handle** out)
{
    if (!s || !out) return fail(TRT_EINVAL, "trt_create: null argument");
    *out = nullptr;
    SceneImage im;
    if (int e = buildSceneImage(s, im)) return e;
    return createOnDevice(im, device, out);
}

void trt_destroy(trt_handle* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    delete h;
}

// One pass = all bounces of `sc_count` samples of every pixel of the tile.  Up to N_SLOTS passes are in
// flight at once, each on its own stream with its own queues: while one pass sits in the latency-bound
// shade kernel or in the host round trip that reads the queue lengths back, the other keeps the CUs busy
// with traversal.  Passes are independent (every sample has its own RNG stream and its own Lacc entry);
// only the per-pixel accumulation is ordered, so the resolves are issued strictly in pass order.
namespace {
constexpr int N_SLOTS = 2;
struct PassSlot {
    hipStream_t stream = nullptr;
    RayQueue Q[2];
    f4* hit = nullptr;
    f4* Lacc = nullptr;
    ShadowQueue SQ[TRT_MAX_LIGHTS];
    uint32_t* d_counts = nullptr;
    uint32_t* host_counts = nullptr;  // pinned, device-visible: 2 * COUNT_ROW counters + the sequence word
    uint32_t seq = 0;                 // last sequence number asked for
    uint32_t* spill = nullptr;
    RedoList redo{nullptr, nullptr};  // rays the traversal kernels hand to k_trace_fix (trt_kernels.h)
    enum State { IDLE, ISSUE, WAIT, RESOLVE } state = IDLE;
    uint32_t chunk = 0, s0 = 0, sc_count = 0, n_active = 0, b = 0;
    int cur = 0;
};
}  // namespace

namespace {
// The render loop behind trt_render_device / trt_render / trt_render_samples: samples [s_begin, s_end) of
// p->spp, added in sample order onto the per-pixel double sums (`accum_host`: in/out when given, else the
// sums start at zero and are dropped), then rounded to float into out_dev.
int renderCore(trt_handle* h, const trt_params* p, uint32_t s_begin, uint32_t s_end, float* out_dev, void* hip_stream, trt_stats* stats_out, double* accum_host)
{
    if (int e = checkParams(h, p)) return e;
    if (!out_dev) return fail(TRT_EINVAL, "null output buffer");
    if (s_begin >= s_end || s_end > (uint32_t)p->spp) return fail(TRT_EINVAL, "sample range must satisfy 0 <= begin < end <= spp");
    const uint32_t n_samples = s_end - s_begin;
    HIPC(hipSetDevice(h->device));
    hipStream_t stream = (hipStream_t)hip_stream;
    const bool count = (p->flags & TRT_FLAG_COUNT) != 0;
    const uint32_t nl = h->sc.n_lights;

    std::vector<int32_t> rows;
    for (int y = p->y0; y < p->y1; ++y)
        if (rowSelected(p, y)) rows.push_back(y);
    if (rows.empty()) return fail(TRT_EINVAL, "row interleave selects no rows of the tile");
    const uint32_t tw = (uint32_t)(p->x1 - p->x0);
    const uint64_t npix64 = (uint64_t)rows.size() * tw;
    if (npix64 > 0x7FFFFFFFull) return fail(TRT_EINVAL, "tile too large");
    const uint32_t npix = (uint32_t)npix64;

    // ---- chunking: how many samples of every pixel one pass holds; >= N_SLOTS passes when spp allows ----
    // Passes are as large as HBM allows: every pass ends in a tail of few, long paths, so fewer and larger passes
    // are faster (back 1080p x 256 spp: 3 passes in 32 GiB 101.4 ms, 1 pass in 93 GB 96.8 ms).  Default budget:
    // three quarters of what is free on the device (the scene is already resident); halved on an allocation failure.
    const uint64_t bytes_per_path = 2ull * 48 + 16 + 16 + (uint64_t)nl * 48 + 4;  // queues, hit, Lacc, shadow queues, redo list
    uint64_t budget = p->mem_budget;
    const bool own_budget = budget == 0;
    if (own_budget) {
        size_t free_b = 0, total_b = 0;
        HIPC(hipMemGetInfo(&free_b, &total_b));
        budget = (uint64_t)(free_b + h->arena.bytes) / 4 * 3;
    }
    const int n_slots = (n_samples >= 2 && (p->flags & TRT_FLAG_OVERLAP) && h->n_slots > 1) ? N_SLOTS : 1;
    int slots_used = 1;
    uint32_t s_chunk = 1, n_chunks = 1;
    uint64_t N = 0;
    size_t q16 = 0, per_slot = 0;
    for (;;) {
        const uint64_t cap_paths = std::min<uint64_t>(budget / bytes_per_path, 0x7FFF0000ull);
        uint64_t max_paths = cap_paths;
        if (max_paths < npix) return fail(TRT_ENOMEM, "mem_budget too small for one sample of every pixel of the tile; render smaller tiles");
        if (n_slots > 1 && max_paths / n_slots >= npix) max_paths /= n_slots;  // each slot gets its share of the budget
        slots_used = (max_paths * n_slots <= cap_paths) ? n_slots : 1;
        s_chunk = (uint32_t)std::min<uint64_t>((uint64_t)n_samples, max_paths / npix);
        n_chunks = (n_samples + s_chunk - 1) / s_chunk;
        if (slots_used > 1 && n_chunks < (uint32_t)slots_used) n_chunks = (uint32_t)std::min<uint32_t>((uint32_t)slots_used, n_samples);
        s_chunk = (n_samples + n_chunks - 1) / n_chunks;
        n_chunks = (n_samples + s_chunk - 1) / s_chunk;
        N = (uint64_t)npix * s_chunk;
        // ---- carve the arena: one set of queues per slot
        q16 = (size_t)N * sizeof(f4);
        per_slot = q16 * (3 * 2 + 1 + 1 + 3 * (size_t)nl) + (((size_t)N + 3) / 4) * sizeof(f4);  // + the redo list (one index per path)
        const int e = h->arena.ensure(per_slot * (size_t)slots_used);
        if (e == TRT_OK) break;
        if (e != TRT_ENOMEM || !own_budget || budget / 2 < bytes_per_path * npix) return e;
        (void)hipGetLastError();  // the failed hipMalloc
        budget /= 2;
    }
    const size_t rows_bytes = (rows.size() * sizeof(int32_t) + 255) & ~(size_t)255;
    const size_t counts_bytes = (size_t)COUNT_STRIDE * COUNT_ROW * sizeof(uint32_t);
    const size_t stats_bytes = 256;  // DeviceStats (128 B), then the redo counters of the pass slots (two words each)
    const size_t acc_bytes = (size_t)npix * 3 * sizeof(double);
    if (int e = h->small_buf.ensure(rows_bytes + counts_bytes * N_SLOTS + stats_bytes + acc_bytes)) return e;
    char* sb = (char*)h->small_buf.p;
    int32_t* d_rows = (int32_t*)sb;
    DeviceStats* d_stats = (DeviceStats*)(sb + rows_bytes + counts_bytes * N_SLOTS);
    static_assert(sizeof(DeviceStats) <= 128, "the redo counters of the pass slots live behind the statistics");
    uint32_t* d_redo = (uint32_t*)(sb + rows_bytes + counts_bytes * N_SLOTS + 128);  // per slot: length of the redo list, blocks of k_trace_fix that are through
    double* d_acc = (double*)(sb + rows_bytes + counts_bytes * N_SLOTS + stats_bytes);

    PassSlot slots[N_SLOTS];
    for (int k = 0; k < slots_used; ++k) {
        PassSlot& S = slots[k];
        S.stream = h->slot_streams[k];
        f4* base = (f4*)((char*)h->arena.p + per_slot * (size_t)k);
        auto take = [&]() { f4* r = base; base += N; return r; };
        for (int q = 0; q < 2; ++q) { S.Q[q].ra = take(); S.Q[q].rb = take(); S.Q[q].bt = take(); }
        S.hit = take();
        S.Lacc = take();
        for (uint32_t l = 0; l < (uint32_t)TRT_MAX_LIGHTS; ++l) S.SQ[l] = ShadowQueue{nullptr, nullptr, nullptr};
        for (uint32_t l = 0; l < nl; ++l) { S.SQ[l].sa = take(); S.SQ[l].sb = take(); S.SQ[l].sw = take(); }
        S.redo.idx = (uint32_t*)base;  // N indices behind the queues
        S.redo.count = d_redo + 2 * k;
        S.d_counts = (uint32_t*)(sb + rows_bytes + counts_bytes * (size_t)k);
        S.host_counts = h->pinned_counts + (size_t)k * (2 * COUNT_ROW + 16);
        S.seq = h->slot_seq[k];
        S.spill = (uint32_t*)h->spill.p + (size_t)k * h->spill_words_per_slot;
    }

    HIPC(hipMemcpyAsync(d_rows, rows.data(), rows.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    HIPC(hipMemsetAsync(d_stats, 0, stats_bytes, stream));  // statistics and redo counters
    if (accum_host) HIPC(hipMemcpyAsync(d_acc, accum_host, acc_bytes, hipMemcpyHostToDevice, stream));
    else HIPC(hipMemsetAsync(d_acc, 0, acc_bytes, stream));

    TileDesc td;
    td.rows = d_rows;
    td.tile_w = (int32_t)tw;
    td.x0 = p->x0;
    td.width = p->width;
    td.height = p->height;
    td.npix = npix;
    td.seed = p->seed;
    td.spp = (uint32_t)p->spp;
    td.fixed_nee = (p->flags & TRT_FLAG_FIXED_NEE) ? 1u : 0u;
    td.fixed_pixels = (p->flags & TRT_FLAG_FIXED_PIXELS) ? 1u : 0u;
    td.ray_offset = (p->flags & TRT_FLAG_RAY_OFFSET) ? 1u : 0u;
    td.specular_ks = (p->flags & TRT_FLAG_SPECULAR_KS) ? 1u : 0u;
    td.npix_magic = magicOf(npix);
    td.tile_w_magic = magicOf(tw);
    td.grid_ok = (p->width >= 2 && p->height >= 2 && p->width <= 65536 && p->height <= 65536) ? 1u : 0u;
    td.grid_rcp[0] = 1.0 / double(p->width - 1.0);
    td.grid_rcp[1] = 1.0 / double(p->height - 1.0);
    td.grid_rcp[2] = 1.0 / double(p->width);
    td.grid_rcp[3] = 1.0 / double(p->height);

    Timer tm;
    tm.h = h;
    tm.on = (p->flags & TRT_FLAG_TIMING) != 0;
    trt_stats st;
    std::memset(&st, 0, sizeof(st));
    // events 0/1 bracket the render on the caller's stream, 2 chains the ordered resolves
    hipEvent_t ev_begin = tm.get(0), ev_end = tm.get(1), ev_resolved = tm.get(2);
    if (!ev_begin || !ev_end || !ev_resolved) return fail(TRT_EHIP, "hipEventCreate failed");
    HIPC(hipEventRecord(ev_begin, stream));
    // From here on work is in flight on the slot streams.  Whatever way this function is left, nothing may still be
    // running on them when it returns: a late kernel of a failed call would write the arena, Lacc and the pinned
    // counters of the NEXT call on this handle.  The guard drains them (and keeps the sequence numbers monotonic).
    struct Drain {
        trt_handle* h;
        PassSlot* slots;
        int n;
        hipStream_t caller;
        bool armed = true;
        ~Drain()
        {
            for (int k = 0; k < n; ++k) h->slot_seq[k] = slots[k].seq;
            if (!armed) return;
            for (int k = 0; k < n; ++k) (void)hipStreamSynchronize(slots[k].stream);
            (void)hipStreamSynchronize(caller);
            (void)hipGetLastError();
        }
    } drain{h, slots, slots_used, stream};
    for (int k = 0; k < slots_used; ++k) HIPC(hipStreamWaitEvent(slots[k].stream, ev_begin, 0));

    uint32_t next_chunk = 0, resolved_upto = 0;
    auto startPass = [&](PassSlot& S) -> int {
        if (next_chunk >= n_chunks) { S.state = PassSlot::IDLE; return TRT_OK; }
        S.chunk = next_chunk++;
        S.s0 = s_begin + S.chunk * s_chunk;
        S.sc_count = std::min(s_chunk, s_end - S.s0);
        S.n_active = npix * S.sc_count;
        S.b = 0;
        S.cur = 0;
        HIPC(hipMemsetAsync(S.d_counts, 0, counts_bytes, S.stream));
        st.rays_camera += S.n_active;  // bounce 0 generates its camera rays inside the traversal and shade kernels
        S.state = PassSlot::ISSUE;
        return TRT_OK;
    };
    // trace + shade of the slot's current bounce, then the queue lengths on their way to the host
    auto issueFront = [&](PassSlot& S) -> int {
        RaySource src;
        src.ra = S.Q[S.cur].ra;
        src.rb = S.Q[S.cur].rb;
        src.td = td;
        src.s0 = S.s0;
        tm.begin(TRT_K_TRACE_CLOSEST, S.stream);
        if (S.b == 0) {
            if (count) launchTraceClosest<true, true>(h, S.stream, S.spill, src, S.hit, S.n_active, d_stats, S.redo);
            else launchTraceClosest<false, true>(h, S.stream, S.spill, src, S.hit, S.n_active, d_stats, S.redo);
        } else {
            if (count) launchTraceClosest<true, false>(h, S.stream, S.spill, src, S.hit, S.n_active, d_stats, S.redo);
            else launchTraceClosest<false, false>(h, S.stream, S.spill, src, S.hit, S.n_active, d_stats, S.redo);
        }
        tm.end(S.stream);
        st.launches[TRT_K_TRACE_CLOSEST]++;

        ShadeArgs A;
        A.qin = S.Q[S.cur];
        A.hit = S.hit;
        A.n = S.n_active;
        A.qout = S.Q[S.cur ^ 1];
        for (int l = 0; l < TRT_MAX_LIGHTS; ++l) A.sq[l] = S.SQ[l];
        A.pair_count = pairCounter(S.d_counts, S.b);
        A.shadow_counts = S.d_counts + (size_t)COUNT_STRIDE + S.b;  // light l: + l * COUNT_STRIDE
        A.shadow_count_stride = COUNT_STRIDE;
        A.Lacc = S.Lacc;
        A.td = td;
        A.s0 = S.s0;
        A.max_depth = p->max_depth;
        A.primary = S.b == 0 ? 1u : 0u;
        A.lds_mat_bytes = h->lds_tab[0];
        A.lds_light_bytes = h->lds_tab[1];
        A.lds_cum_bytes = h->lds_tab[2];
        A.lds_ltri_bytes = h->lds_tab[3];
        A.lds_tshade_bytes = h->lds_tab[4];
        A.lds_image = (const f4*)h->lds_image;
        A.lds_image_words = h->lds_image_bytes / 16u;
        const bool one_light = nl == 1u;  // k_shade's two flavours (trt_kernels.h): 512-thread blocks at 6 waves per SIMD, or 256-thread blocks at 5
        const uint32_t shade_block = one_light ? (uint32_t)TRT_SHADE1_BLOCK : (uint32_t)TRT_SHADEN_BLOCK;
        A.rows_lds = (rows.size() <= shadeRowsLds((int)shade_block) && p->height <= 65536) ? (uint32_t)rows.size() : 0u;
        A.stats = d_stats;
        tm.begin(TRT_K_SHADE, S.stream);
        {
            const dim3 grid(std::min<uint32_t>((S.n_active + shade_block - 1) / shade_block, 65536u)), blk(shade_block);
#define TRT_LAUNCH_SHADE(T) \
            if (one_light) hipLaunchKernelGGL((k_shade<T, true>), grid, blk, h->shade_pad_lds, S.stream, h->sc, A); \
            else hipLaunchKernelGGL((k_shade<T, false>), grid, blk, h->shade_pad_lds, S.stream, h->sc, A);
            switch (h->shade_tabs) {
                case 31u: TRT_LAUNCH_SHADE(31u) break;
                case 15u: TRT_LAUNCH_SHADE(15u) break;
                case 7u: TRT_LAUNCH_SHADE(7u) break;
                case 3u: TRT_LAUNCH_SHADE(3u) break;
                default: TRT_LAUNCH_SHADE(0u) break;
            }
#undef TRT_LAUNCH_SHADE
        }
        tm.end(S.stream);
        st.launches[TRT_K_SHADE]++;
        // (b, c) and (b + 1, c) of the counters in use -> host_counts[2 * c], [2 * c + 1], then the sequence word
        S.seq++;
        hipLaunchKernelGGL(k_publish_counts, dim3(1), dim3(64), 0, S.stream, S.d_counts, COUNT_STRIDE, S.b, 1u + nl, pairCounter(S.d_counts, S.b),
                           (volatile uint32_t*)S.host_counts, S.seq);
        S.state = PassSlot::WAIT;
        if (h->fail_at_bounce >= 0 && (int)S.b == h->fail_at_bounce) {  // test hook: fail with this bounce's kernels in flight
            h->fail_at_bounce = -1;
            return fail(TRT_EHIP, "injected failure (TRT_TEST_FAIL_AT_BOUNCE)");
        }
        return TRT_OK;
    };
    // queue lengths are back: shadow rays of this bounce, then the next bounce / the tail / the end of the pass
    auto completeBounce = [&](PassSlot& S) -> int {
        {   // spin on the sequence word the device writes after the counters; the stream is the fallback (and the error path)
            volatile uint32_t* flag = (volatile uint32_t*)S.host_counts + 2 * COUNT_ROW;
            const auto t0 = std::chrono::steady_clock::now();
            uint32_t spins = 0;
            while (*flag != S.seq) {
                if ((++spins & 0x3FFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
                    HIPC(hipStreamSynchronize(S.stream));  // long kernels: let the runtime wait; also surfaces a device error
                    if (*flag != S.seq) return fail(TRT_EHIP, "queue lengths did not arrive");
                    break;
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        for (uint32_t l = 0; l < nl; ++l) {
            const uint32_t ns = S.host_counts[2 * (1 + l)];
            if (ns > S.n_active) return fail(TRT_EHIP, "internal error: shadow queue longer than its input");
            if (!ns) continue;
            tm.begin(TRT_K_TRACE_SHADOW, S.stream);
            if (count) launchTraceShadow<true>(h, S.stream, S.spill, S.SQ[l], ns, h->light_mats[l], S.Lacc, d_stats, td.fixed_nee, S.redo, h->light_boxes[l]);
            else launchTraceShadow<false>(h, S.stream, S.spill, S.SQ[l], ns, h->light_mats[l], S.Lacc, d_stats, td.fixed_nee, S.redo, h->light_boxes[l]);
            tm.end(S.stream);
            st.launches[TRT_K_TRACE_SHADOW]++;
            st.rays_shadow += ns;
        }
        const uint32_t n_next = S.host_counts[1];
        if (n_next > S.n_active) return fail(TRT_EHIP, "internal error: queue grew");
        st.rays_indirect += n_next;
        S.n_active = n_next;
        S.cur ^= 1;
        S.b++;
        if (S.n_active > 0 && (S.n_active <= h->tail_n || S.b >= MAX_BOUNCES)) {
            // few paths left: finish them in one launch (k_tail) instead of ~3 launches + a host round trip per bounce
            TailArgs TA;
            TA.q = S.Q[S.cur];
            TA.n = S.n_active;
            TA.Lacc = S.Lacc;
            TA.td = td;
            TA.s0 = S.s0;
            TA.max_depth = p->max_depth;
            TA.spill = S.spill;
            TA.spill_stride = SPILL_STRIDE;
            TA.uniform = h->trace_impl == 0 ? 1u : 0u;
            TA.stats = d_stats;
            tm.begin(TRT_K_TAIL, S.stream);
            if (count) hipLaunchKernelGGL((k_tail<true, 0>), dim3(tailGrid(S.n_active)), dim3(TRT_TRACE_BLOCK), 0, S.stream, h->sc, TA);
            else hipLaunchKernelGGL((k_tail<false, 0>), dim3(tailGrid(S.n_active)), dim3(TRT_TRACE_BLOCK), 0, S.stream, h->sc, TA);
            tm.end(S.stream);
            st.launches[TRT_K_TAIL]++;
            S.n_active = 0;
        }
        S.state = S.n_active ? PassSlot::ISSUE : PassSlot::RESOLVE;
        return TRT_OK;
    };
    // per-pixel accumulation in sample order: pass c is resolved only after pass c-1 (on whichever stream that ran)
    auto tryResolve = [&](PassSlot& S) -> int {
        if (S.chunk != resolved_upto) return TRT_OK;  // an earlier pass is still in flight on the other slot
        if (resolved_upto > 0) HIPC(hipStreamWaitEvent(S.stream, ev_resolved, 0));  // recorded by the previous resolve, earlier in host order
        tm.begin(TRT_K_RESOLVE, S.stream);
        hipLaunchKernelGGL(k_resolve, dim3(std::min<uint32_t>((npix + 255) / 256, 65536u)), dim3(256), 0, S.stream, S.Lacc, d_acc, npix, S.sc_count, (float)p->spp);
        tm.end(S.stream);
        st.launches[TRT_K_RESOLVE]++;
        HIPC(hipEventRecord(ev_resolved, S.stream));
        resolved_upto++;
        return startPass(S);
    };

    for (int k = 0; k < slots_used; ++k)
        if (int e = startPass(slots[k])) return e;
    for (;;) {
        bool any = false;
        for (int k = 0; k < slots_used; ++k)
            if (slots[k].state == PassSlot::ISSUE) { if (int e = issueFront(slots[k])) return e; }
        for (int k = 0; k < slots_used; ++k) {
            PassSlot& S = slots[k];
            if (S.state == PassSlot::WAIT) { if (int e = completeBounce(S)) return e; }
            if (S.state == PassSlot::RESOLVE) { if (int e = tryResolve(S)) return e; }
            any = any || S.state != PassSlot::IDLE;
        }
        if (!any) break;
    }
    if (resolved_upto != n_chunks) return fail(TRT_EHIP, "internal error: passes left unresolved");

    HIPC(hipStreamWaitEvent(stream, ev_resolved, 0));
    tm.begin(TRT_K_RESOLVE, stream);
    hipLaunchKernelGGL(k_finalize, dim3(std::min<uint32_t>((npix * 3 + 255) / 256, 65536u)), dim3(256), 0, stream, d_acc, out_dev, npix * 3);
    tm.end(stream);
    HIPC(hipEventRecord(ev_end, stream));
    DeviceStats ds;
    HIPC(hipMemcpyAsync(&ds, d_stats, sizeof(ds), hipMemcpyDeviceToHost, stream));
    if (accum_host) HIPC(hipMemcpyAsync(accum_host, d_acc, acc_bytes, hipMemcpyDeviceToHost, stream));
    HIPC(hipStreamSynchronize(stream));
    HIPC(hipGetLastError());
    drain.armed = false;  // everything this call enqueued has completed

    float ms = 0.f;
    HIPC(hipEventElapsedTime(&ms, ev_begin, ev_end));
    st.render_ms = ms;
    for (const auto& sp : tm.spans) {
        float k_ms = 0.f;
        if (hipEventElapsedTime(&k_ms, h->events[sp.e0], h->events[sp.e1]) == hipSuccess) st.kernel_ms[sp.k] += k_ms;
    }
    st.shaded_hits = ds.shaded_hits;
    st.wave_steps[0] = ds.wave_inner_steps;
    st.wave_steps[1] = ds.wave_leaf_steps;
    st.rays_shadow += ds.tail_rays_shadow;
    st.rays_indirect += ds.tail_rays_indirect;
    for (int i = 0; i < 2; ++i) { st.inner_visits[i] = ds.inner_visits[i]; st.tri_tests[i] = ds.tri_tests[i]; }
    st.max_bounces = ds.max_depth_hit;
    st.redo_rays = ds.redo_rays;
    st.lane_census[0] = ds.census_inner; st.lane_census[1] = ds.census_leaf; st.lane_census[2] = ds.census_done; st.lane_census[3] = ds.census_iters;
    st.passes = n_chunks;
    st.rows_rendered = rows.size();
    st.inner_node_bytes = h->trace_impl == 0 ? (uint32_t)sizeof(trt_bvh_node) : (h->node_kind == 1 ? 80u : (uint32_t)sizeof(WideNode));
    if (stats_out) *stats_out = st;
    return TRT_OK;
}
}  // namespace

int trt_render_device(trt_handle* h, const trt_params* p, float* out_dev, void* hip_stream, trt_stats* stats_out)
{
    if (int e = checkParams(h, p)) return e;
    return renderCore(h, p, 0u, (uint32_t)p->spp, out_dev, hip_stream, stats_out, nullptr);
}

int trt_render_samples(trt_handle* h, const trt_params* p, int32_t sample_begin, int32_t sample_end, double* accum_host, float* out_host, trt_stats* stats)
{
    if (int e = checkParams(h, p)) return e;
    if (!accum_host) return fail(TRT_EINVAL, "null accumulator");
    if (sample_begin < 0 || sample_end <= sample_begin || sample_end > p->spp) return fail(TRT_EINVAL, "sample range must satisfy 0 <= begin < end <= spp");
    HIPC(hipSetDevice(h->device));
    const int nrows = trt_rows_selected(p);
    if (nrows < 1) return fail(TRT_EINVAL, "row interleave selects no rows of the tile");
    const size_t bytes = (size_t)nrows * (size_t)(p->x1 - p->x0) * 3 * sizeof(float);
    if (int e = h->out_buf.ensure(bytes)) return e;
    if (int e = renderCore(h, p, (uint32_t)sample_begin, (uint32_t)sample_end, (float*)h->out_buf.p, nullptr, stats, accum_host)) return e;
    if (out_host) HIPC(hipMemcpy(out_host, h->out_buf.p, bytes, hipMemcpyDeviceToHost));
    return TRT_OK;
}

int trt_render(trt_handle* h, const trt_params* p, float* out_host, trt_stats* stats)
{
    if (int e = checkParams(h, p)) return e;
    if (!out_host) return fail(TRT_EINVAL, "null output buffer");
    HIPC(hipSetDevice(h->device));
    const int nrows = trt_rows_selected(p);
    if (nrows < 1) return fail(TRT_EINVAL, "row interleave selects no rows of the tile");
    const size_t bytes = (size_t)nrows * (size_t)(p->x1 - p->x0) * 3 * sizeof(float);
    if (int e = h->out_buf.ensure(bytes)) return e;
    if (int e = trt_render_device(h, p, (float*)h->out_buf.p, nullptr, stats)) return e;
    HIPC(hipMemcpy(out_host, h->out_buf.p, bytes, hipMemcpyDeviceToHost));
    return TRT_OK;
}

int trt_trace_closest(trt_handle* h, uint64_t n, const float* org, const float* dir, float* t, int32_t* tri, float* uv, trt_stats* stats_out)
{
    if (!h || !org || !dir || !t || !tri) return fail(TRT_EINVAL, "trt_trace_closest: null argument");
    if (n == 0) return TRT_OK;
    if (n > 0x7FFF0000ull) return fail(TRT_EINVAL, "ray batch too large");
    HIPC(hipSetDevice(h->device));
    const uint32_t n32 = (uint32_t)n;
    const size_t in_bytes = (size_t)n * 3 * sizeof(float);
    const size_t q16 = (size_t)n * sizeof(f4);
    if (int e = h->io_buf.ensure(2 * in_bytes + 3 * q16 + 256 + 256 + (size_t)n * sizeof(uint32_t))) return e;  // + statistics, counters, redo list
    char* b = (char*)h->io_buf.p;
    f4* ra = (f4*)b;
    f4* rb = ra + n;
    f4* hit = rb + n;
    float* d_org = (float*)(hit + n);
    float* d_dir = d_org + (size_t)n * 3;
    DeviceStats* d_stats = (DeviceStats*)(d_dir + (size_t)n * 3);
    d_stats = (DeviceStats*)(((uintptr_t)d_stats + 15) & ~(uintptr_t)15);
    RedoList redo;  // rays for k_trace_fix (trt_kernels.h): the counter sits behind the statistics, the list behind it
    redo.count = (uint32_t*)((char*)d_stats + 128);
    redo.idx = (uint32_t*)((char*)d_stats + 256);
    HIPC(hipMemcpy(d_org, org, in_bytes, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(d_dir, dir, in_bytes, hipMemcpyHostToDevice));
    HIPC(hipMemset(d_stats, 0, 256));
    hipLaunchKernelGGL(k_pack_rays, dim3(std::min<uint32_t>((n32 + 255) / 256, 65536u)), dim3(256), 0, nullptr, d_org, d_dir, ra, rb, n32);
    struct Events {  // destroyed on every path out of this function
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events()
        {
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } ev;
    HIPC(hipEventCreate(&ev.e0));
    HIPC(hipEventCreate(&ev.e1));
    hipEvent_t e0 = ev.e0, e1 = ev.e1;
    HIPC(hipEventRecord(e0, nullptr));
    RaySource src{};
    src.ra = ra;
    src.rb = rb;
    launchTraceClosest<true, false>(h, nullptr, (uint32_t*)h->spill.p, src, hit, n32, d_stats, redo);
    HIPC(hipEventRecord(e1, nullptr));
    HIPC(hipDeviceSynchronize());
    HIPC(hipGetLastError());
    float ms = 0.f;
    HIPC(hipEventElapsedTime(&ms, e0, e1));
    std::vector<f4> hh(n);
    HIPC(hipMemcpy(hh.data(), hit, q16, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n; ++i) {
        t[i] = hh[i].x;
        tri[i] = (int32_t)f2u(hh[i].y);
        if (uv) { uv[i * 2] = hh[i].z; uv[i * 2 + 1] = hh[i].w; }
    }
    if (stats_out) {
        DeviceStats ds;
        HIPC(hipMemcpy(&ds, d_stats, sizeof(ds), hipMemcpyDeviceToHost));
        std::memset(stats_out, 0, sizeof(*stats_out));
        stats_out->inner_visits[0] = ds.inner_visits[0];
        stats_out->tri_tests[0] = ds.tri_tests[0];
        stats_out->wave_steps[0] = ds.wave_inner_steps;
        stats_out->wave_steps[1] = ds.wave_leaf_steps;
        stats_out->kernel_ms[TRT_K_TRACE_CLOSEST] = ms;
        stats_out->launches[TRT_K_TRACE_CLOSEST] = 1;
        stats_out->inner_node_bytes = h->trace_impl == 0 ? (uint32_t)sizeof(trt_bvh_node) : (h->node_kind == 1 ? 80u : (uint32_t)sizeof(WideNode));
        stats_out->redo_rays = ds.redo_rays;
        stats_out->lane_census[0] = ds.census_inner; stats_out->lane_census[1] = ds.census_leaf; stats_out->lane_census[2] = ds.census_done; stats_out->lane_census[3] = ds.census_iters;
    }
    return TRT_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ device groups
namespace {
// The five RCCL entry points the gather needs, bound at run time: a single-GPU user of this library never loads RCCL.
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Gather)(const void* send, void* recv, size_t count, int datatype, int root, void* comm, hipStream_t stream) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load(std::string& err)
    {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("cannot load librccl.so.1: ") + dlerror(); return false; }
        auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) err = std::string("librccl: missing symbol ") + n; return p; };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Gather = reinterpret_cast<decltype(Gather)>(sym("ncclGather"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Gather && GetErrorString;
    }
};
constexpr int NCCL_FLOAT32 = 7;  // ncclFloat32 (rccl.h ncclDataType_t)

// packed stripes of rank r (rows in increasing y) -> their rows of the tile image
__global__ __launch_bounds__(256) void k_uninterleave(const float* __restrict__ gathered, float* __restrict__ image, uint32_t tile_rows, uint32_t row_floats,
                                                       uint32_t row_block, uint32_t n_ranks, uint32_t pad_rows)
{
    const uint64_t total = (uint64_t)tile_rows * row_floats;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t y = (uint32_t)(i / row_floats), x = (uint32_t)(i - (uint64_t)y * row_floats);
        const uint32_t stripe = y / row_block, rank = stripe % n_ranks;
        const uint32_t packed = (stripe / n_ranks) * row_block + (y - stripe * row_block);  // row of y inside rank's packed buffer
        image[i] = gathered[((uint64_t)rank * pad_rows + packed) * row_floats + x];
    }
}
}  // namespace

// One host thread per device for the life of the group: it binds its device once and renders its stripes whenever the
// group posts a job (trt_group_render used to create and join n threads per call).
struct GroupWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    uint64_t posted = 0, finished = 0;  // job sequence numbers
    bool quit = false;
    // the job
    trt_handle* h = nullptr;
    trt_params p{};
    float* out = nullptr;
    hipStream_t stream = nullptr;
    bool skip = false;  // more devices than stripes: nothing to render
    // its result
    int rc = TRT_OK;
    std::string msg;
    trt_stats st{};
};

struct trt_group {
    std::vector<trt_handle*> handles;
    std::vector<int> devices;
    bool use_rccl = false;       // every entry another device (or the one-device test switch): RCCL gathers; else device copies (one-GPU rehearsal)
    Rccl rccl;
    std::vector<void*> comms;
    std::vector<hipStream_t> streams;
    std::vector<DevBuf> stripe;  // per rank: its packed stripes, padded to the largest rank's row count
    DevBuf gathered, image;      // on devices[0]
    std::vector<std::unique_ptr<GroupWorker>> workers;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // on devices[0]: around gather + un-interleave
    void startWorkers()
    {
        for (size_t k = 0; k < handles.size(); ++k) {
            workers.emplace_back(new GroupWorker);
            GroupWorker* w = workers.back().get();
            const int dev = devices[k];
            w->th = std::thread([w, dev]() {
                (void)hipSetDevice(dev);
                std::unique_lock<std::mutex> lk(w->mu);
                for (;;) {
                    w->cv.wait(lk, [w]() { return w->quit || w->posted != w->finished; });
                    if (w->quit) return;
                    lk.unlock();
                    if (w->skip) { std::memset(&w->st, 0, sizeof(w->st)); w->rc = TRT_OK; }
                    else {
                        w->rc = trt_render_device(w->h, &w->p, w->out, w->stream, &w->st);
                        if (w->rc) w->msg = trt_last_error();  // thread-local in the worker: carried over by hand
                    }
                    lk.lock();
                    w->finished = w->posted;
                    w->cv.notify_all();
                }
            });
        }
    }
    ~trt_group()
    {
        for (auto& w : workers) {
            { std::lock_guard<std::mutex> lk(w->mu); w->quit = true; }
            w->cv.notify_all();
            if (w->th.joinable()) w->th.join();
        }
        for (size_t k = 0; k < handles.size(); ++k) {
            (void)hipSetDevice(devices[k]);
            if (k < stripe.size()) stripe[k].release();
            if (k < streams.size() && streams[k]) (void)hipStreamDestroy(streams[k]);
            if (k < comms.size() && comms[k]) (void)rccl.CommDestroy(comms[k]);
            trt_destroy(handles[k]);
        }
        if (!devices.empty()) (void)hipSetDevice(devices[0]);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        gathered.release();
        image.release();
        if (rccl.lib) dlclose(rccl.lib);
    }
};

extern "C" {

int trt_group_size(const trt_group* g) { return g ? (int)g->handles.size() : 0; }

void trt_group_destroy(trt_group* g) { delete g; }

int trt_group_create(const trt_scene* scene, int n_devices, const int* devices, trt_group** out)
{
    if (!scene || !devices || !out || n_devices < 1 || n_devices > 64) return fail(TRT_EINVAL, "trt_group_create: bad argument");
    *out = nullptr;
    std::unique_ptr<trt_group> g(new trt_group);
    bool distinct = true;
    for (int a = 0; a < n_devices; ++a)
        for (int b = a + 1; b < n_devices; ++b)
            if (devices[a] == devices[b]) distinct = false;
    const char* force = std::getenv("TRT_GROUP_FORCE_RCCL");  // test switch: a group of one device takes the RCCL route too
    g->use_rccl = distinct && (n_devices > 1 || (force && std::atoi(force) != 0));
    {   // the host half of trt_create once (checks, collapses: seconds for 10 M triangles), the device half per member, side by side
        SceneImage im;
        if (int e = buildSceneImage(scene, im)) return e;
        std::vector<trt_handle*> hs(n_devices, nullptr);
        std::vector<int> rc(n_devices, TRT_OK);
        std::vector<std::string> msg(n_devices);
        std::vector<std::thread> th;
        for (int k = 0; k < n_devices; ++k)
            th.emplace_back([&, k] {
                rc[k] = createOnDevice(im, devices[k], &hs[k]);
                if (rc[k] != TRT_OK) msg[k] = trt_last_error();  // the message lives in this thread
            });
        for (std::thread& t : th) t.join();
        for (int k = 0; k < n_devices; ++k) {
            if (hs[k]) { g->handles.push_back(hs[k]); g->devices.push_back(devices[k]); }  // (owned by the group from here: freed with it on failure)
        }
        for (int k = 0; k < n_devices; ++k)
            if (rc[k] != TRT_OK) return fail(rc[k], msg[k]);  // TRT_ENODEV for an ordinal the node does not have
    }
    g->stripe.resize(n_devices);
    g->streams.assign(n_devices, nullptr);
    for (int k = 0; k < n_devices; ++k) {
        HIPC(hipSetDevice(devices[k]));
        HIPC(hipStreamCreateWithFlags(&g->streams[k], hipStreamNonBlocking));
    }
    HIPC(hipSetDevice(devices[0]));
    HIPC(hipEventCreate(&g->ev0));
    HIPC(hipEventCreate(&g->ev1));
    if (g->use_rccl) {
        std::string err;
        if (!g->rccl.load(err)) return fail(TRT_EHIP, err);
        g->comms.assign(n_devices, nullptr);
        const int rc = g->rccl.CommInitAll(g->comms.data(), n_devices, devices);
        if (rc != 0) return fail(TRT_EHIP, std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(rc));
    }
    g->startWorkers();
    *out = g.release();
    return TRT_OK;
}

int trt_group_render_device(trt_group* g, const trt_params* p_in, float* out_dev0, trt_stats* stats_out, double* gather_ms_out)
{
    if (!g || !p_in || !out_dev0) return fail(TRT_EINVAL, "trt_group_render_device: null argument");
    const int n = (int)g->handles.size();
    trt_params p = *p_in;
    if (p.row_block <= 0) p.row_block = 8;
    p.row_mod = n;
    p.row_rem = 0;
    if (int e = checkParams(g->handles[0], &p)) return e;
    const uint32_t tile_rows = (uint32_t)(p.y1 - p.y0), tw = (uint32_t)(p.x1 - p.x0), row_floats = tw * 3u;
    // Rows are selected on absolute y (stripes are counted from image row 0), and k_uninterleave computes the packed index the
    // same way only if the tile starts on a stripe boundary of rank 0: require that.
    if (p.y0 % (p.row_block * n) != 0) return fail(TRT_EINVAL, "trt_group_render: y0 must be a multiple of row_block * group size");
    uint32_t pad_rows = 0;
    std::vector<uint32_t> rows_of(n);
    for (int k = 0; k < n; ++k) {
        trt_params pk = p;
        pk.row_rem = k;
        rows_of[k] = (uint32_t)std::max(trt_rows_selected(&pk), 0);
        pad_rows = std::max(pad_rows, rows_of[k]);
    }
    const size_t stripe_bytes = (size_t)pad_rows * row_floats * sizeof(float);
    for (int k = 0; k < n; ++k) {
        HIPC(hipSetDevice(g->devices[k]));
        if (int e = g->stripe[k].ensure(stripe_bytes)) return e;
    }
    HIPC(hipSetDevice(g->devices[0]));
    if (int e = g->gathered.ensure(stripe_bytes * (size_t)n)) return e;

    // ---- every device renders its stripes on its own (resident) host thread
    for (int k = 0; k < n; ++k) {
        GroupWorker* w = g->workers[k].get();
        std::lock_guard<std::mutex> lk(w->mu);
        w->h = g->handles[k];
        w->p = p;
        w->p.row_rem = k;
        w->out = (float*)g->stripe[k].p;
        w->stream = g->streams[k];
        w->skip = rows_of[k] == 0;
        w->posted++;
        w->cv.notify_all();
    }
    for (int k = 0; k < n; ++k) {
        GroupWorker* w = g->workers[k].get();
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [w]() { return w->posted == w->finished; });
    }
    for (int k = 0; k < n; ++k)
        if (g->workers[k]->rc) return fail(g->workers[k]->rc, "device " + std::to_string(g->devices[k]) + ": " + g->workers[k]->msg);

    // ---- ONE gather to devices[0], then un-interleave there (every worker's stream is idle: trt_render_device synchronises it)
    HIPC(hipSetDevice(g->devices[0]));
    HIPC(hipEventRecord(g->ev0, g->streams[0]));
    const size_t count = (size_t)pad_rows * row_floats;
    if (g->use_rccl) {
        // Between GroupStart and GroupEnd nothing returns: an error is remembered, the group is closed, then it is reported.
        int rc = g->rccl.GroupStart();
        hipError_t herr = hipSuccess;
        if (rc == 0) {
            for (int k = 0; k < n && rc == 0 && herr == hipSuccess; ++k) {
                herr = hipSetDevice(g->devices[k]);
                // (a rank that is not the root receives nothing; it still gets a valid pointer — its own stripe — in case a build of the library checks the argument)
                if (herr == hipSuccess) rc = g->rccl.Gather(g->stripe[k].p, k == 0 ? g->gathered.p : g->stripe[k].p, count, NCCL_FLOAT32, 0, g->comms[k], g->streams[k]);
            }
            const int rc_end = g->rccl.GroupEnd();
            if (rc == 0) rc = rc_end;
        }
        (void)hipSetDevice(g->devices[0]);
        if (herr != hipSuccess) return fail(TRT_EHIP, std::string("hipSetDevice inside the gather: ") + hipGetErrorString(herr));
        if (rc != 0) return fail(TRT_EHIP, std::string("ncclGather: ") + g->rccl.GetErrorString(rc));
    } else {
        for (int k = 0; k < n; ++k)
            HIPC(hipMemcpyAsync((char*)g->gathered.p + stripe_bytes * (size_t)k, g->stripe[k].p, stripe_bytes, hipMemcpyDeviceToDevice, g->streams[0]));
    }
    // the root's share of the gather runs on streams[0]: the kernel below is ordered behind it by the stream
    const uint64_t total = (uint64_t)tile_rows * row_floats;
    hipLaunchKernelGGL(k_uninterleave, dim3((uint32_t)std::min<uint64_t>((total + 255) / 256, 65536ull)), dim3(256), 0, g->streams[0], (const float*)g->gathered.p,
                       out_dev0, tile_rows, row_floats, (uint32_t)p.row_block, (uint32_t)n, pad_rows);
    HIPC(hipEventRecord(g->ev1, g->streams[0]));
    HIPC(hipStreamSynchronize(g->streams[0]));
    HIPC(hipGetLastError());
    if (g->use_rccl)  // the senders' stripes may be overwritten by the next render only after their part of the gather has left
        for (int k = 1; k < n; ++k) { HIPC(hipSetDevice(g->devices[k])); HIPC(hipStreamSynchronize(g->streams[k])); }
    HIPC(hipSetDevice(g->devices[0]));
    float gms = 0.f;
    HIPC(hipEventElapsedTime(&gms, g->ev0, g->ev1));
    if (gather_ms_out) *gather_ms_out = gms;
    if (stats_out) {
        trt_stats t;
        std::memset(&t, 0, sizeof(t));
        for (int k = 0; k < n; ++k) {
            const trt_stats& s = g->workers[k]->st;
            t.rays_camera += s.rays_camera; t.rays_shadow += s.rays_shadow; t.rays_indirect += s.rays_indirect; t.shaded_hits += s.shaded_hits;
            for (int i = 0; i < 2; ++i) { t.inner_visits[i] += s.inner_visits[i]; t.tri_tests[i] += s.tri_tests[i]; t.wave_steps[i] += s.wave_steps[i]; }
            for (int i = 0; i < TRT_MAX_KERNELS; ++i) { t.launches[i] += s.launches[i]; t.kernel_ms[i] += s.kernel_ms[i]; }
            t.render_ms = std::max(t.render_ms, s.render_ms);
            t.passes = std::max(t.passes, s.passes);
            t.max_bounces = std::max(t.max_bounces, s.max_bounces);
            t.rows_rendered += s.rows_rendered;
            t.inner_node_bytes = std::max(t.inner_node_bytes, s.inner_node_bytes);
            t.redo_rays += s.redo_rays;
            for (int i = 0; i < 4; ++i) t.lane_census[i] += s.lane_census[i];
        }
        *stats_out = t;
    }
    return TRT_OK;
}

int trt_group_render(trt_group* g, const trt_params* p_in, float* out_host, trt_stats* stats_out, double* gather_ms_out)
{
    if (!g || !p_in || !out_host) return fail(TRT_EINVAL, "trt_group_render: null argument");
    if (p_in->x1 <= p_in->x0 || p_in->y1 <= p_in->y0) return fail(TRT_EINVAL, "tile rectangle outside the image or empty");
    const size_t bytes = (size_t)(p_in->y1 - p_in->y0) * (size_t)(p_in->x1 - p_in->x0) * 3 * sizeof(float);
    HIPC(hipSetDevice(g->devices[0]));
    if (int e = g->image.ensure(bytes)) return e;
    if (int e = trt_group_render_device(g, p_in, (float*)g->image.p, stats_out, gather_ms_out)) return e;
    HIPC(hipSetDevice(g->devices[0]));
    HIPC(hipMemcpy(out_host, g->image.p, bytes, hipMemcpyDeviceToHost));
    return TRT_OK;
}

}  // extern "C"
