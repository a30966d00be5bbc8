// ThreadSanitizer driver for the threaded host half of trt_create (trt_wide.h `par`, collapseBvh, collapseBvhGreedy, buildOct,
// leafBoxesOf): a random median-split BVH2 over random triangles, collapsed with 8 threads and with 1, results compared.
//   g++ -fsanitize=thread -O1 -g -std=c++17 -ffp-contract=off -Iinclude -Itinyraytracing_amd/csrc tools/tsan_collapse.cpp -o /tmp/tsan_collapse -pthread && /tmp/tsan_collapse
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "trt.h"
#include "trt_path.h"
#include "trt_wide.h"
#include "trt_oct_build.h"

using namespace trtd;

struct Tri { float v[9]; float c[3]; };

static uint32_t build(std::vector<Tri>& t, size_t lo, size_t hi, std::vector<trt_bvh_node>& out, float blo[3], float bhi[3])
{
    auto bounds = [&](size_t a, size_t b, float* l, float* h) {
        for (int k = 0; k < 3; ++k) { l[k] = 3e38f; h[k] = -3e38f; }
        for (size_t i = a; i < b; ++i)
            for (int v = 0; v < 3; ++v)
                for (int k = 0; k < 3; ++k) { l[k] = std::fmin(l[k], t[i].v[v * 3 + k]); h[k] = std::fmax(h[k], t[i].v[v * 3 + k]); }
        for (int k = 0; k < 3; ++k) { l[k] -= 0.001f; h[k] += 0.001f; }
    };
    bounds(lo, hi, blo, bhi);
    if (hi - lo <= 2) return TRT_MAKE_LEAF(lo, hi - lo);
    int axis = 0;
    for (int k = 1; k < 3; ++k)
        if (bhi[k] - blo[k] > bhi[axis] - blo[axis]) axis = k;
    const size_t mid = (lo + hi) / 2;
    std::nth_element(t.begin() + (long)lo, t.begin() + (long)mid, t.begin() + (long)hi, [&](const Tri& a, const Tri& b) { return a.c[axis] < b.c[axis]; });
    const uint32_t me = (uint32_t)out.size();
    out.emplace_back();
    float l0[3], h0[3], l1[3], h1[3];
    const uint32_t c0 = build(t, lo, mid, out, l0, h0), c1 = build(t, mid, hi, out, l1, h1);
    trt_bvh_node& nd = out[me];
    std::memcpy(nd.lo0, l0, 12); std::memcpy(nd.hi0, h0, 12); std::memcpy(nd.lo1, l1, 12); std::memcpy(nd.hi1, h1, 12);
    nd.child0 = c0; nd.child1 = c1; nd.reserved[0] = nd.reserved[1] = 0;
    return me;
}

int main()
{
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> U(-10.f, 10.f), S(-0.2f, 0.2f);
    const size_t n = 120000;
    std::vector<Tri> t(n);
    for (Tri& x : t) {
        const float p[3] = {U(rng), U(rng), U(rng)};
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) x.v[v * 3 + k] = p[k] + S(rng);
        for (int k = 0; k < 3; ++k) x.c[k] = (x.v[k] + x.v[3 + k] + x.v[6 + k]) / 3.0f;
    }
    std::vector<trt_bvh_node> nodes;
    float l[3], h[3];
    build(t, 0, n, nodes, l, h);
    std::vector<TriIsect> isect(n);
    for (size_t i = 0; i < n; ++i) isect[i] = makeTriIsect(t[i].v, 0, false);
    int bad = 0;
    for (unsigned threads : {8u, 3u}) {
        const WideTree a1 = collapseBvh(nodes.data(), (uint32_t)nodes.size(), 1), a8 = collapseBvh(nodes.data(), (uint32_t)nodes.size(), threads);
        const WideTree g1 = collapseBvhGreedy(nodes.data(), (uint32_t)nodes.size(), 1), g8 = collapseBvhGreedy(nodes.data(), (uint32_t)nodes.size(), threads);
        const OctTree o1 = buildOct(nodes.data(), (uint32_t)nodes.size(), (uint32_t)n, isect.data(), 1), o8 = buildOct(nodes.data(), (uint32_t)nodes.size(), (uint32_t)n, isect.data(), threads);
        const std::vector<f4> b1 = leafBoxesOf(nodes.data(), (uint32_t)nodes.size(), (uint32_t)n, 1), b8 = leafBoxesOf(nodes.data(), (uint32_t)nodes.size(), (uint32_t)n, threads);
        auto same = [](const void* p, const void* q, size_t bytes) { return std::memcmp(p, q, bytes) == 0; };
        bad += !(a1.nodes.size() == a8.nodes.size() && same(a1.nodes.data(), a8.nodes.data(), a1.nodes.size() * sizeof(WideNode)) && a1.stack_need == a8.stack_need && a1.dropped == a8.dropped);
        bad += !(g1.nodes.size() == g8.nodes.size() && same(g1.nodes.data(), g8.nodes.data(), g1.nodes.size() * sizeof(WideNode)) && g1.stack_need == g8.stack_need && g1.dropped == g8.dropped);
        bad += !(o1.ok && o8.ok && o1.nodes.size() == o8.nodes.size() && same(o1.nodes.data(), o8.nodes.data(), o1.nodes.size() * sizeof(OctNode)) &&
                 same(o1.tri_trav.data(), o8.tri_trav.data(), o1.tri_trav.size() * sizeof(TriIsect)) && o1.levels == o8.levels);
        bad += !same(b1.data(), b8.data(), b1.size() * sizeof(f4));
        std::printf("%u threads: %zu / %zu wide nodes, %zu oct nodes in %u levels: %s\n", threads, a8.nodes.size(), g8.nodes.size(), o8.nodes.size(), o8.levels, bad ? "DIFFERENT" : "same bytes as with one thread");
    }
    return bad ? 1 : 0;
}
