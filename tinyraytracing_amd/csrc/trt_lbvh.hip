// trt_lbvh.hip — the GPU BVH builder behind include/trt_build.h (libtrt_lbvh.so, gfx950).
//
// Role: buildBVH(scene.triangles, 0, n - 1, leaf_num) of the reference (bvh.cpp:16-144, called at main.cpp:76) for scenes of
// millions of triangles, as a linear BVH built entirely on the device:
//   K1 k_prim_boxes   triangle -> its box (32 B) and, reduced per grid, the bounds of the box centres
//   K2 k_morton       box centre -> 63-bit Morton code (21 bits per axis in the centre bounds), value = triangle index
//      rocprim::radix_sort_pairs (64-bit keys, one pass over the 63 bits that are used)
//   K3 k_hierarchy    inner node i of the radix tree over the sorted codes from i alone (Karras 2012: direction, range by
//                     doubling + bisection on the common-prefix length, split by bisection; equal codes are told apart by
//                     their position, so runs of duplicates become balanced subtrees)
//   K4 k_boxes_up + k_span_round   boxes bottom-up: one thread per triangle climbs, the second arrival at a node merges and goes on —
//                     inside a block's 1024 positions in LDS; the few nodes that span blocks in per-level launches behind it
//   K5 k_survive + rocprim::exclusive_scan + k_emit   inner nodes that hold more than leaf_num triangles become the flat
//                     64-B nodes of trt.h (a radix-tree node covers a contiguous range of the sorted order, so a subtree of
//                     <= leaf_num triangles IS a leaf (first, count)); boxes padded like the reference's (bvh.cpp:31-40)
//   K6 k_depth        inner nodes on the longest root path
//   K7-K11 (default)  the top of the tree by SAH: the radix tree cut into clusters of <= 2048 triangles, an exact sweep-SAH tree over
//                     the clusters built on the host, the cluster subtrees emitted below it (see "the top of the tree by SAH" below)
// All HBM-bound streaming or gather work, 4.1 ms of kernels for 10 M triangles (9 ms with the host's SAH over the clusters in between); the rest of the call is moving the
// vertices in (360 MB) and the nodes out (< 640 MB) over PCIe.  No MFMA, no LDS tiling: nothing here is a contraction.
#include <hip/hip_runtime.h>

#include <string.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "trt.h"
#include "trt_build.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIPC(expr)                                                                                                              \
    do {                                                                                                                        \
        hipError_t e_ = (expr);                                                                                                 \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? TRT_ENOMEM : TRT_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct DevMem {  // frees what a failed call leaves behind
    std::vector<void*> p;
    ~DevMem() { for (void* q : p) (void)hipFree(q); }
    template <class T>
    hipError_t alloc(T** out, size_t count)
    {
        void* q = nullptr;
        const hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) { p.push_back(q); *out = static_cast<T*>(q); }
        return e;
    }
};

constexpr uint32_t LEAF = 0x80000000u;  // in child references of the radix tree: a single triangle (sorted position), else an inner node
constexpr float PAD = 0.001f;           // bvh.cpp:31-40

// order-preserving map float -> uint32 (for atomicMin / atomicMax on floats of either sign)
__device__ inline uint32_t fkey(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float fkeyInv(uint32_t k)
{
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

struct Box8 {  // (lo.xyz, hi.x) (hi.yz, -, -)
    float4 a, b;
};

// K1: bounds[0..2] = min of the box centres (keys), bounds[3..5] = max
__global__ __launch_bounds__(256) void k_prim_boxes(const float* __restrict__ tri_v, uint32_t n, Box8* __restrict__ pbox, uint32_t* __restrict__ bounds)
{
    float cmin[3] = {3.0e38f, 3.0e38f, 3.0e38f}, cmax[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float* v = tri_v + (size_t)i * 9;
        float lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(v[a], fminf(v[3 + a], v[6 + a]));
            hi[a] = fmaxf(v[a], fmaxf(v[3 + a], v[6 + a]));
            const float c = 0.5f * lo[a] + 0.5f * hi[a];
            cmin[a] = fminf(cmin[a], c);
            cmax[a] = fmaxf(cmax[a], c);
        }
        Box8 b;
        b.a = make_float4(lo[0], lo[1], lo[2], hi[0]);
        b.b = make_float4(hi[1], hi[2], 0.0f, 0.0f);
        pbox[i] = b;
    }
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            cmin[a] = fminf(cmin[a], __shfl_xor(cmin[a], off));
            cmax[a] = fmaxf(cmax[a], __shfl_xor(cmax[a], off));
        }
    }
    if ((threadIdx.x & 63u) == 0) {
        for (int a = 0; a < 3; ++a) {
            atomicMin(&bounds[a], fkey(cmin[a]));
            atomicMax(&bounds[3 + a], fkey(cmax[a]));
        }
    }
}

__device__ inline unsigned long long spread21(unsigned long long x)
{
    x &= 0x1FFFFFull;
    x = (x | (x << 32)) & 0x001F00000000FFFFull;
    x = (x | (x << 16)) & 0x001F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

struct CentreFrame {
    float lo[3], scale[3];  // scale = 2^21 / extent (0 for a flat axis)
};

// K2
__global__ __launch_bounds__(256) void k_morton(const Box8* __restrict__ pbox, uint32_t n, CentreFrame f, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Box8 b = pbox[i];
    const float c[3] = {0.5f * b.a.x + 0.5f * b.a.w, 0.5f * b.a.y + 0.5f * b.b.x, 0.5f * b.a.z + 0.5f * b.b.y};
    unsigned long long q[3];
    for (int a = 0; a < 3; ++a) {
        float t = (c[a] - f.lo[a]) * f.scale[a];
        t = fminf(fmaxf(t, 0.0f), 2097151.0f);  // (a NaN centre lands on 0)
        q[a] = (unsigned long long)(uint32_t)t;
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    vals[i] = i;
}

// common-prefix length of the codes at sorted positions i and j (-1 outside the array); equal codes go on with the positions
__device__ inline int delta(const unsigned long long* __restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

// K3: inner node i in [0, n - 2]; node 0 is the root
__global__ __launch_bounds__(256) void k_hierarchy(const unsigned long long* __restrict__ keys, uint32_t n_prims, uint32_t* __restrict__ left, uint32_t* __restrict__ right,
                                                   uint32_t* __restrict__ first, uint32_t* __restrict__ last, uint32_t* __restrict__ parent, uint32_t* __restrict__ leaf_parent)
{
    const int n = (int)n_prims;
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;  // (ends: outside the array delta is -1 <= dmin)
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const uint32_t lc = (lo == gamma) ? (LEAF | (uint32_t)gamma) : (uint32_t)gamma;
    const uint32_t rc = (hi == gamma + 1) ? (LEAF | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    left[i] = lc;
    right[i] = rc;
    first[i] = (uint32_t)lo;
    last[i] = (uint32_t)hi;
    if (lc & LEAF) leaf_parent[gamma] = (uint32_t)i; else parent[gamma] = (uint32_t)i;
    if (rc & LEAF) leaf_parent[gamma + 1] = (uint32_t)i; else parent[gamma + 1] = (uint32_t)i;
    if (i == 0) parent[0] = 0xFFFFFFFFu;
}

// K4: nbox[i] = bounds of inner node i, in two parts.
// K4a: one thread per triangle climbs from its leaf carrying the box of the subtree it has completed; at a node the first of the two
// threads to arrive is done, the second merges the SIBLING's box into its own, stores the node's box and goes on.  A block owns the 1024
// consecutive sorted positions of its threads.  A radix-tree node's index is one end of its range, so a node whose whole range lies
// inside the block's positions — all but about ten of the block's 1023 — has a slot in LDS: arrival counter and box live there (LDS
// atomics, a workgroup fence), its final box goes to memory.  The climb ENDS at the first node that spans a block boundary: meeting
// another block's thread there would take agent-scope acquire/release, which on this chip (eight XCDs, an L2 each) is a write-back and
// an invalidation of an L2 per arrival — with every node handled that way the kernel took 36 of the builder's 40 ms for 10 M triangles,
// with only the spanning nodes 11.5 ms.
// K4b: the spanning nodes (about ten per block, compacted into a list) are finished level by level: one small launch per round computes
// every listed node whose two children are complete; the kernel boundary is all the coherence the rounds need.
constexpr uint32_t BOXES_UP_BLOCK = 1024;
__device__ inline bool spansBlocks(uint32_t first, uint32_t last) { return first / BOXES_UP_BLOCK != last / BOXES_UP_BLOCK; }

__global__ __launch_bounds__(BOXES_UP_BLOCK) void k_boxes_up(const Box8* __restrict__ pbox, const uint32_t* __restrict__ order, uint32_t n_prims, const uint32_t* __restrict__ left,
                                                             const uint32_t* __restrict__ right, const uint32_t* __restrict__ first, const uint32_t* __restrict__ last,
                                                             const uint32_t* __restrict__ parent, const uint32_t* __restrict__ leaf_parent, Box8* __restrict__ nbox)
{
    __shared__ float s_box[6][BOXES_UP_BLOCK];
    __shared__ uint32_t s_arrivals[BOXES_UP_BLOCK];
    const uint32_t own_lo = blockIdx.x * BOXES_UP_BLOCK;
    const uint32_t p = own_lo + threadIdx.x;
    s_arrivals[threadIdx.x] = 0u;
    __syncthreads();  // (the only barrier: threads leave the kernel one by one after it)
    if (p >= n_prims || n_prims < 2) return;
    float cur[6];
    {
        const Box8 x = pbox[order[p]];
        cur[0] = x.a.x; cur[1] = x.a.y; cur[2] = x.a.z; cur[3] = x.a.w; cur[4] = x.b.x; cur[5] = x.b.y;
    }
    uint32_t came_from = LEAF | p, node = leaf_parent[p];
    for (uint32_t guard = 0; guard < 4096u; ++guard) {  // (a root path is at most 63 + 32 nodes long; the guard only bounds a corrupted tree)
        if (spansBlocks(first[node], last[node])) return;  // K4b's
        __threadfence_block();  // my subtree's box (s_box, if it is a node's) before my arrival
        const uint32_t before = atomicAdd(&s_arrivals[node - own_lo], 1u);
        __threadfence_block();
        if (before == 0u) return;  // the other child is still under way: its thread will finish this node
        const uint32_t l = left[node], r = right[node], sib = (l == came_from) ? r : l;
        float b[6];
        if (sib & LEAF) {
            const Box8 x = pbox[order[sib & ~LEAF]];
            b[0] = x.a.x; b[1] = x.a.y; b[2] = x.a.z; b[3] = x.a.w; b[4] = x.b.x; b[5] = x.b.y;
        } else {  // both children of a node inside the block are inside it
            for (int k = 0; k < 6; ++k) b[k] = s_box[k][sib - own_lo];
        }
        for (int a = 0; a < 3; ++a) { cur[a] = fminf(cur[a], b[a]); cur[3 + a] = fmaxf(cur[3 + a], b[3 + a]); }
        for (int k = 0; k < 6; ++k) s_box[k][node - own_lo] = cur[k];
        Box8 out;
        out.a = make_float4(cur[0], cur[1], cur[2], cur[3]);
        out.b = make_float4(cur[4], cur[5], 0.0f, 0.0f);
        nbox[node] = out;
        const uint32_t up = parent[node];
        if (up == 0xFFFFFFFFu) return;  // the root (a scene of one block)
        came_from = node;
        node = up;
    }
}

// K4b, list: flag = 1 for the nodes that span blocks; `pos` = exclusive scan of the flags
__global__ __launch_bounds__(256) void k_span_flags(const uint32_t* __restrict__ first, const uint32_t* __restrict__ last, uint32_t n_inner, uint32_t* __restrict__ flag)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_inner) flag[i] = spansBlocks(first[i], last[i]) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_span_list(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos, uint32_t n_inner, uint32_t* __restrict__ list)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_inner && flag[i]) list[pos[i]] = i;
}
// K4b, one round: a child is complete if it is a triangle, a node K4a finished (it does not span), or marked done by an EARLIER launch (done_in: read
// only; this launch marks in done_out, and the two swap between rounds).  `remaining` counts the listed nodes that had to wait.
__global__ __launch_bounds__(256) void k_span_round(const uint32_t* __restrict__ list, uint32_t m, const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                                    const uint32_t* __restrict__ first, const uint32_t* __restrict__ last, const Box8* __restrict__ pbox,
                                                    const uint32_t* __restrict__ order, Box8* nbox, const uint32_t* done_in, uint32_t* done_out, uint32_t* __restrict__ remaining)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const uint32_t node = list[t];
    if (done_in[node]) { done_out[node] = 1u; return; }
    const uint32_t ch[2] = {left[node], right[node]};
    bool ready = true;
    for (int k = 0; k < 2; ++k)
        if (!(ch[k] & LEAF) && spansBlocks(first[ch[k]], last[ch[k]]) && !done_in[ch[k]]) ready = false;
    if (!ready) { atomicAdd(remaining, 1u); return; }
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int k = 0; k < 2; ++k) {
        const Box8 x = (ch[k] & LEAF) ? pbox[order[ch[k] & ~LEAF]] : nbox[ch[k]];
        lo[0] = fminf(lo[0], x.a.x); lo[1] = fminf(lo[1], x.a.y); lo[2] = fminf(lo[2], x.a.z);
        hi[0] = fmaxf(hi[0], x.a.w); hi[1] = fmaxf(hi[1], x.b.x); hi[2] = fmaxf(hi[2], x.b.y);
    }
    Box8 out;
    out.a = make_float4(lo[0], lo[1], lo[2], hi[0]);
    out.b = make_float4(hi[1], hi[2], 0.0f, 0.0f);
    nbox[node] = out;
    done_out[node] = 1u;
}

// K5a: 1 for the inner nodes that stay nodes (more than leaf_num triangles below them)
__global__ __launch_bounds__(256) void k_survive(const uint32_t* __restrict__ first, const uint32_t* __restrict__ last, uint32_t n_inner, uint32_t leaf_num, uint32_t* __restrict__ keep)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_inner) keep[i] = (last[i] - first[i] + 1u > leaf_num) ? 1u : 0u;
}

// K5b
__global__ __launch_bounds__(256) void k_emit(const Box8* __restrict__ pbox, const uint32_t* __restrict__ order, const Box8* __restrict__ nbox, const uint32_t* __restrict__ left,
                                              const uint32_t* __restrict__ right, const uint32_t* __restrict__ first, const uint32_t* __restrict__ last,
                                              const uint32_t* __restrict__ keep, const uint32_t* __restrict__ new_index, uint32_t n_inner, trt_bvh_node* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner || !keep[i]) return;
    trt_bvh_node nd;
    const uint32_t ch[2] = {left[i], right[i]};
    for (int k = 0; k < 2; ++k) {
        Box8 x;
        uint32_t ref;
        if (ch[k] & LEAF) {
            const uint32_t pos = ch[k] & ~LEAF;
            x = pbox[order[pos]];
            ref = TRT_MAKE_LEAF(pos, 1u);
        } else {
            x = nbox[ch[k]];
            ref = keep[ch[k]] ? new_index[ch[k]] : TRT_MAKE_LEAF(first[ch[k]], last[ch[k]] - first[ch[k]] + 1u);
        }
        float* lo = k ? nd.lo1 : nd.lo0;
        float* hi = k ? nd.hi1 : nd.hi0;
        lo[0] = x.a.x - PAD; lo[1] = x.a.y - PAD; lo[2] = x.a.z - PAD;
        hi[0] = x.a.w + PAD; hi[1] = x.b.x + PAD; hi[2] = x.b.y + PAD;
        if (k) nd.child1 = ref; else nd.child0 = ref;
    }
    nd.reserved[0] = nd.reserved[1] = 0;
    out[new_index[i]] = nd;
}

// K6: per triangle, the kept inner nodes above it
__global__ __launch_bounds__(256) void k_depth(const uint32_t* __restrict__ parent, const uint32_t* __restrict__ leaf_parent, const uint32_t* __restrict__ keep, uint32_t n_prims,
                                               uint32_t* __restrict__ depth_max)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t d = 0;
    if (p < n_prims && n_prims >= 2) {
        uint32_t node = leaf_parent[p];
        for (uint32_t guard = 0; guard < 4096u && node != 0xFFFFFFFFu; ++guard) {
            d += keep[node];
            node = parent[node];
        }
    }
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(d, off); d = o > d ? o : d; }
    if ((threadIdx.x & 63u) == 0 && d) atomicMax(depth_max, d);
}

// ------------------------------------------------------------------ the top of the tree by SAH ----
// A radix tree splits at fixed planes of the scene box whatever lies on either side; most of what that costs a ray is
// lost in the upper levels (measured on the CPU with the host builder standing in: SAH above 4096 triangles and plain
// spatial medians below is within 4-8 % of SAH throughout, medians throughout 25-45 % behind).  So the radix
// tree is cut into CLUSTERS — the maximal subtrees of <= cluster_size triangles, a few thousand of them for 10 M triangles —,
// the clusters' boxes and sizes go to the host, an exact sweep-SAH tree over them (cost = area x triangles below) becomes
// the top of the BVH, and each cluster's radix subtree hangs below its leaf.  The triangle order follows: clusters in the
// top tree's leaf order, Morton order inside a cluster (a shift per cluster).

// K7: marks the first sorted position of every cluster and notes its root there (LEAF | position for a single triangle)
__global__ __launch_bounds__(256) void k_cluster_starts(const uint32_t* __restrict__ first, const uint32_t* __restrict__ last, const uint32_t* __restrict__ parent,
                                                        const uint32_t* __restrict__ leaf_parent, uint32_t n_prims, uint32_t cluster_size, uint32_t* __restrict__ start,
                                                        uint32_t* __restrict__ root_at)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    auto isTop = [&](uint32_t i) { return last[i] - first[i] + 1u > cluster_size; };
    if (t < n_prims - 1u) {  // inner node t
        if (!isTop(t) && t != 0u && isTop(parent[t])) { start[first[t]] = 1u; root_at[first[t]] = t; }
    }
    if (t < n_prims) {  // triangle at sorted position t
        if (isTop(leaf_parent[t])) { start[t] = 1u; root_at[t] = LEAF | t; }
    }
}

struct ClusterRec {  // what the host needs of a cluster
    float lo[3], hi[3];
    uint32_t first, count;
};

// K8: incl = inclusive scan of start
__global__ __launch_bounds__(256) void k_cluster_gather(const uint32_t* __restrict__ start, const uint32_t* __restrict__ incl, const uint32_t* __restrict__ root_at,
                                                        const uint32_t* __restrict__ first, const uint32_t* __restrict__ last, const Box8* __restrict__ pbox,
                                                        const uint32_t* __restrict__ order, const Box8* __restrict__ nbox, uint32_t n_prims, ClusterRec* __restrict__ rec,
                                                        uint32_t* __restrict__ croot)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_prims || !start[p]) return;
    const uint32_t c = incl[p] - 1u, root = root_at[p];
    const Box8 b = (root & LEAF) ? pbox[order[p]] : nbox[root];
    ClusterRec r;
    r.lo[0] = b.a.x; r.lo[1] = b.a.y; r.lo[2] = b.a.z; r.hi[0] = b.a.w; r.hi[1] = b.b.x; r.hi[2] = b.b.y;
    r.first = p;
    r.count = (root & LEAF) ? 1u : last[root] - first[root] + 1u;
    rec[c] = r;
    croot[c] = root;
}

// K9a: inner nodes INSIDE clusters that stay nodes
__global__ __launch_bounds__(256) void k_survive_in_clusters(const uint32_t* __restrict__ first, const uint32_t* __restrict__ last, uint32_t n_inner, uint32_t leaf_num,
                                                             uint32_t cluster_size, uint32_t* __restrict__ keep)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner) return;
    const uint32_t cnt = last[i] - first[i] + 1u;
    keep[i] = (cnt > leaf_num && cnt <= cluster_size) ? 1u : 0u;
}

// K9b: the cluster subtrees as nodes [base, base + kept), triangle positions shifted to the new order
__global__ __launch_bounds__(256) void k_emit_clusters(const Box8* __restrict__ pbox, const uint32_t* __restrict__ order, const Box8* __restrict__ nbox, const uint32_t* __restrict__ left,
                                                       const uint32_t* __restrict__ right, const uint32_t* __restrict__ first, const uint32_t* __restrict__ last,
                                                       const uint32_t* __restrict__ keep, const uint32_t* __restrict__ new_index, const uint32_t* __restrict__ incl,
                                                       const int32_t* __restrict__ shift, uint32_t n_inner, uint32_t base, trt_bvh_node* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner || !keep[i]) return;
    const int32_t sh = shift[incl[first[i]] - 1u];
    trt_bvh_node nd;
    const uint32_t ch[2] = {left[i], right[i]};
    for (int k = 0; k < 2; ++k) {
        Box8 x;
        uint32_t ref;
        if (ch[k] & LEAF) {
            const uint32_t pos = ch[k] & ~LEAF;
            x = pbox[order[pos]];
            ref = TRT_MAKE_LEAF((uint32_t)((int32_t)pos + sh), 1u);
        } else {
            x = nbox[ch[k]];
            ref = keep[ch[k]] ? base + new_index[ch[k]] : TRT_MAKE_LEAF((uint32_t)((int32_t)first[ch[k]] + sh), last[ch[k]] - first[ch[k]] + 1u);
        }
        float* lo = k ? nd.lo1 : nd.lo0;
        float* hi = k ? nd.hi1 : nd.hi0;
        lo[0] = x.a.x - PAD; lo[1] = x.a.y - PAD; lo[2] = x.a.z - PAD;
        hi[0] = x.a.w + PAD; hi[1] = x.b.x + PAD; hi[2] = x.b.y + PAD;
        if (k) nd.child1 = ref; else nd.child0 = ref;
    }
    nd.reserved[0] = nd.reserved[1] = 0;
    out[new_index[i]] = nd;
}

// K10: per cluster: where its root went (kept inner roots only) — and the new triangle order
__global__ __launch_bounds__(256) void k_cluster_root_index(const uint32_t* __restrict__ croot, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ new_index,
                                                            uint32_t n_clusters, uint32_t* __restrict__ cidx)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_clusters) return;
    const uint32_t r = croot[c];
    cidx[c] = (!(r & LEAF) && keep[r]) ? new_index[r] : 0xFFFFFFFFu;
}
__global__ __launch_bounds__(256) void k_permute_order(const uint32_t* __restrict__ order, const uint32_t* __restrict__ incl, const int32_t* __restrict__ shift, uint32_t n_prims,
                                                       uint32_t* __restrict__ order2)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_prims) order2[(uint32_t)((int32_t)p + shift[incl[p] - 1u])] = order[p];
}
// K11: kept nodes between a triangle and the root of its cluster, maximum per cluster
__global__ __launch_bounds__(256) void k_cluster_depth(const uint32_t* __restrict__ parent, const uint32_t* __restrict__ leaf_parent, const uint32_t* __restrict__ keep,
                                                       const uint32_t* __restrict__ incl, uint32_t n_prims, uint32_t* __restrict__ cdepth)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t d = 0, c = 0xFFFFFFFFu;
    if (p < n_prims) {
        c = incl[p] - 1u;
        uint32_t node = leaf_parent[p];
        for (uint32_t guard = 0; guard < 4096u && node != 0xFFFFFFFFu; ++guard) {  // (`keep` is 0 below the leaves' roots and above the cluster's root)
            if (keep[node]) ++d;
            else if (d) break;  // past the cluster's root
            node = parent[node];
        }
    }
    // neighbours share clusters: one atomic per run of equal cluster ids in the wave (the lane that starts the run carries the run's maximum)
    const uint32_t lane = threadIdx.x & 63u;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t od = __shfl_down(d, off), oc = __shfl_down(c, off);
        if (lane + off < 64u && oc == c) d = od > d ? od : d;
    }
    const uint32_t pc = __shfl_up(c, 1);
    if (p < n_prims && d && (lane == 0u || pc != c)) atomicMax(&cdepth[c], d);
}

// ---- host: exact sweep SAH over the clusters, cost = half-area x triangles
struct TopNode {
    int32_t child[2];  // >= 0: top node; < 0: ~cluster
    float lo[2][3], hi[2][3];
};
struct TopBuilder {
    // The three centre orders are sorted ONCE and kept through the recursion by stable partitions (as host/bvh.cpp does for its exact-SAH
    // subtrees): a node is evaluated by scanning its range of each order, O(n) per node instead of three to four sorts.
    const std::vector<ClusterRec>& cl;
    std::vector<uint32_t> ord[3], tmp;
    std::vector<uint8_t> side;
    std::vector<TopNode> nodes;
    std::vector<uint32_t> leaf_order, top_depth;
    std::vector<float> suffix_area;
    std::vector<uint64_t> suffix_count;
    explicit TopBuilder(const std::vector<ClusterRec>& c) : cl(c), tmp(c.size()), side(c.size(), 0), top_depth(c.size(), 0), suffix_area(c.size()), suffix_count(c.size())
    {
        for (int a = 0; a < 3; ++a) {
            ord[a].resize(c.size());
            for (size_t i = 0; i < c.size(); ++i) ord[a][i] = (uint32_t)i;
            std::sort(ord[a].begin(), ord[a].end(), [&](uint32_t x, uint32_t y) {
                const float cx = centre(x, a), cy = centre(y, a);
                return cx < cy || (cx == cy && x < y);
            });
        }
    }
    static float halfArea(const float* lo, const float* hi)
    {
        const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return x * y + x * z + y * z;
    }
    float centre(uint32_t c, int a) const { return 0.5f * cl[c].lo[a] + 0.5f * cl[c].hi[a]; }
    // returns the reference of the subtree over positions [lo, hi) of the three orders (the same SET in each) and its bounds; parents before children
    int32_t build(size_t lo, size_t hi, uint32_t depth, float* blo, float* bhi)
    {
        if (hi - lo == 1) {
            const uint32_t c = ord[0][lo];
            leaf_order.push_back(c);
            top_depth[c] = depth;
            for (int a = 0; a < 3; ++a) { blo[a] = cl[c].lo[a]; bhi[a] = cl[c].hi[a]; }
            return ~(int32_t)c;
        }
        const size_t m = hi - lo;
        float best = std::numeric_limits<float>::max();
        int best_axis = -1;
        size_t best_mid = lo + m / 2;
        for (int a = 0; a < 3 && depth < 48; ++a) {  // (a runaway depth: medians from there on, as in host/bvh.cpp)
            const uint32_t* o = ord[a].data() + lo;
            float l3[3] = {3.0e38f, 3.0e38f, 3.0e38f}, h3[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
            uint64_t cnt = 0;
            for (size_t i = m; i-- > 1;) {
                const ClusterRec& r = cl[o[i]];
                for (int k = 0; k < 3; ++k) { l3[k] = std::fmin(l3[k], r.lo[k]); h3[k] = std::fmax(h3[k], r.hi[k]); }
                cnt += r.count;
                suffix_area[i] = halfArea(l3, h3);
                suffix_count[i] = cnt;
            }
            for (int k = 0; k < 3; ++k) { l3[k] = 3.0e38f; h3[k] = -3.0e38f; }
            cnt = 0;
            for (size_t i = 1; i < m; ++i) {
                const ClusterRec& r = cl[o[i - 1]];
                for (int k = 0; k < 3; ++k) { l3[k] = std::fmin(l3[k], r.lo[k]); h3[k] = std::fmax(h3[k], r.hi[k]); }
                cnt += r.count;
                const float cost = halfArea(l3, h3) * (float)cnt + suffix_area[i] * (float)suffix_count[i];
                if (cost < best) { best = cost; best_axis = a; best_mid = lo + i; }
            }
        }
        if (best_axis < 0) { best_axis = 0; best_mid = lo + m / 2; }
        for (size_t i = lo; i < hi; ++i) side[ord[best_axis][i]] = i < best_mid ? 0 : 1;
        for (int a = 0; a < 3; ++a) {
            if (a == best_axis) continue;
            size_t l = lo, r = 0;
            for (size_t i = lo; i < hi; ++i) {
                const uint32_t c = ord[a][i];
                if (side[c]) tmp[r++] = c; else ord[a][l++] = c;  // (l <= i: nothing unread is overwritten)
            }
            for (size_t k = 0; k < r; ++k) ord[a][l + k] = tmp[k];
        }
        const int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        float l0[3], h0[3], l1[3], h1[3];
        const int32_t c0 = build(lo, best_mid, depth + 1, l0, h0);
        const int32_t c1 = build(best_mid, hi, depth + 1, l1, h1);
        TopNode& t = nodes[(size_t)me];
        t.child[0] = c0; t.child[1] = c1;
        for (int a = 0; a < 3; ++a) {
            t.lo[0][a] = l0[a]; t.hi[0][a] = h0[a]; t.lo[1][a] = l1[a]; t.hi[1][a] = h1[a];
            blo[a] = std::fmin(l0[a], l1[a]); bhi[a] = std::fmax(h0[a], h1[a]);
        }
        return me;
    }
};

}  // namespace

extern "C" {

const char* trt_build_last_error(void) { return g_err.c_str(); }

int trt_build_lbvh(const float* tri_v, uint32_t n_tris, int leaf_num, int device, trt_bvh_node* nodes_out, uint32_t node_capacity, uint32_t* n_nodes_out,
                   uint32_t* order_out, uint32_t* depth_out, double ms_out[2])
{
    const auto t_host = std::chrono::steady_clock::now();
    if (!nodes_out || !n_nodes_out || !order_out || (n_tris && !tri_v)) return fail(TRT_EINVAL, "trt_build_lbvh: null argument");
    if (leaf_num < 1 || leaf_num > (int)TRT_MAX_LEAF_TRIS) return fail(TRT_EINVAL, "trt_build_lbvh: leaf_num must be in 1..15");
    if (n_tris > TRT_MAX_TRIS) return fail(TRT_EINVAL, "trt_build_lbvh: too many triangles");
    if (node_capacity < 1) return fail(TRT_EINVAL, "trt_build_lbvh: node_capacity must be at least 1");
    if (depth_out) *depth_out = 1;
    if (ms_out) ms_out[0] = ms_out[1] = 0.0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(TRT_ENODEV, "no HIP device");
    if (device < 0 || device >= ndev) return fail(TRT_ENODEV, "device ordinal out of range");
    // the caller's current device comes back on every exit path (declared first: restored after the buffers and events below are released)
    struct DeviceGuard { int prev = -1; ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); } } dev_guard;
    if (hipGetDevice(&dev_guard.prev) != hipSuccess) dev_guard.prev = -1;
    HIPC(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPC(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(TRT_ENODEV, std::string("this library is built for gfx950 only, device is ") + prop.gcnArchName);

    const uint32_t n = n_tris;
    DevMem mem;
    float* d_v = nullptr;
    Box8* d_pbox = nullptr;
    uint32_t* d_bounds = nullptr;
    HIPC(mem.alloc(&d_v, (size_t)n * 9));
    HIPC(mem.alloc(&d_pbox, n));
    HIPC(mem.alloc(&d_bounds, 8));
    hipStream_t stream = nullptr;  // the default stream: every call below is ordered on it
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    HIPC(hipEventCreate(&ev0));
    HIPC(hipEventCreate(&ev1));
    struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evg{ev0, ev1};
    if (n) HIPC(hipMemcpy(d_v, tri_v, (size_t)n * 9 * sizeof(float), hipMemcpyHostToDevice));
    {
        const uint32_t init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
        HIPC(hipMemcpy(d_bounds, init, sizeof(init), hipMemcpyHostToDevice));
    }
    HIPC(hipEventRecord(ev0, stream));
    const uint32_t grid_n = (n + 255u) / 256u;
    if (n) {
        hipLaunchKernelGGL(k_prim_boxes, dim3(std::min(grid_n, 2048u)), dim3(256), 0, stream, d_v, n, d_pbox, d_bounds);
        HIPC(hipGetLastError());
    }

    // ---- a scene that fits one leaf still gets a root: child0 = every triangle, child1 = an empty leaf (as host/bvh.cpp)
    if (n <= (uint32_t)leaf_num) {
        std::vector<Box8> pb(std::max<uint32_t>(n, 1u));
        if (n) HIPC(hipMemcpy(pb.data(), d_pbox, (size_t)n * sizeof(Box8), hipMemcpyDeviceToHost));
        float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        for (uint32_t i = 0; i < n; ++i) {
            const float b[6] = {pb[i].a.x, pb[i].a.y, pb[i].a.z, pb[i].a.w, pb[i].b.x, pb[i].b.y};
            for (int a = 0; a < 3; ++a) {
                lo[a] = i ? std::fmin(lo[a], b[a]) : b[a];
                hi[a] = i ? std::fmax(hi[a], b[3 + a]) : b[3 + a];
            }
        }
        trt_bvh_node root;
        std::memset(&root, 0, sizeof(root));
        for (int a = 0; a < 3; ++a) { root.lo0[a] = root.lo1[a] = lo[a] - PAD; root.hi0[a] = root.hi1[a] = hi[a] + PAD; }
        root.child0 = TRT_MAKE_LEAF(0, n);
        root.child1 = TRT_MAKE_LEAF(0, 0);
        nodes_out[0] = root;
        *n_nodes_out = 1;
        for (uint32_t i = 0; i < n; ++i) order_out[i] = i;
        if (ms_out) ms_out[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host).count();
        return TRT_OK;
    }

    const uint32_t n_inner = n - 1;
    unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
    uint32_t *d_vals = nullptr, *d_order = nullptr, *d_left = nullptr, *d_right = nullptr, *d_first = nullptr, *d_last = nullptr, *d_parent = nullptr, *d_leaf_parent = nullptr,
             *d_arrivals = nullptr, *d_keep = nullptr, *d_new = nullptr, *d_depth = nullptr;
    Box8* d_nbox = nullptr;
    HIPC(mem.alloc(&d_keys, n));
    HIPC(mem.alloc(&d_keys2, n));
    HIPC(mem.alloc(&d_vals, n));
    HIPC(mem.alloc(&d_order, n));
    HIPC(mem.alloc(&d_left, n_inner));
    HIPC(mem.alloc(&d_right, n_inner));
    HIPC(mem.alloc(&d_first, n_inner));
    HIPC(mem.alloc(&d_last, n_inner));
    HIPC(mem.alloc(&d_parent, n_inner));
    HIPC(mem.alloc(&d_leaf_parent, n));
    HIPC(mem.alloc(&d_arrivals, n_inner));
    HIPC(mem.alloc(&d_keep, n_inner));
    HIPC(mem.alloc(&d_new, n_inner));
    HIPC(mem.alloc(&d_depth, 1));
    HIPC(mem.alloc(&d_nbox, n_inner));

    // the centre bounds are needed on the host to form the Morton frame (six words)
    uint32_t bk[8];
    HIPC(hipMemcpy(bk, d_bounds, sizeof(bk), hipMemcpyDeviceToHost));
    CentreFrame frame;
    for (int a = 0; a < 3; ++a) {
        const float lo = fkeyInv(bk[a]), hi = fkeyInv(bk[3 + a]);
        const float ext = hi - lo;
        frame.lo[a] = lo;
        frame.scale[a] = (ext > 0.0f && std::isfinite(ext)) ? 2097152.0f / ext : 0.0f;
    }
    hipLaunchKernelGGL(k_morton, dim3(grid_n), dim3(256), 0, stream, d_pbox, n, frame, d_keys, d_vals);
    HIPC(hipGetLastError());
    {
        size_t tmp_bytes = 0;
        HIPC(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys, d_keys2, d_vals, d_order, (size_t)n, 0u, 63u, stream));
        char* d_tmp = nullptr;
        HIPC(mem.alloc(&d_tmp, tmp_bytes));
        HIPC(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_keys, d_keys2, d_vals, d_order, (size_t)n, 0u, 63u, stream));
    }
    const uint32_t grid_i = (n_inner + 255u) / 256u;
    hipLaunchKernelGGL(k_hierarchy, dim3(grid_i), dim3(256), 0, stream, d_keys2, n, d_left, d_right, d_first, d_last, d_parent, d_leaf_parent);
    HIPC(hipGetLastError());
    hipLaunchKernelGGL(k_boxes_up, dim3((n + BOXES_UP_BLOCK - 1u) / BOXES_UP_BLOCK), dim3(BOXES_UP_BLOCK), 0, stream, d_pbox, d_order, n, d_left, d_right, d_first, d_last, d_parent, d_leaf_parent, d_nbox);
    HIPC(hipGetLastError());
    if (n > BOXES_UP_BLOCK) {  // K4b: the nodes that span blocks, level by level (d_keep / d_new / d_arrivals serve as flag / position / done here)
        uint32_t* d_done2 = nullptr;
        uint32_t* d_list = nullptr;
        HIPC(mem.alloc(&d_done2, n_inner));
        hipLaunchKernelGGL(k_span_flags, dim3(grid_i), dim3(256), 0, stream, d_first, d_last, n_inner, d_keep);
        HIPC(hipGetLastError());
        {
            size_t tmp_bytes = 0;
            HIPC(rocprim::exclusive_scan(nullptr, tmp_bytes, d_keep, d_new, 0u, (size_t)n_inner, rocprim::plus<uint32_t>(), stream));
            char* d_tmp = nullptr;
            HIPC(mem.alloc(&d_tmp, tmp_bytes));
            HIPC(rocprim::exclusive_scan(d_tmp, tmp_bytes, d_keep, d_new, 0u, (size_t)n_inner, rocprim::plus<uint32_t>(), stream));
        }
        uint32_t tail[2] = {0, 0};
        HIPC(hipMemcpy(&tail[0], d_new + (n_inner - 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(&tail[1], d_keep + (n_inner - 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
        const uint32_t m = tail[0] + tail[1];
        if (m) {
            HIPC(mem.alloc(&d_list, m));
            hipLaunchKernelGGL(k_span_list, dim3(grid_i), dim3(256), 0, stream, d_keep, d_new, n_inner, d_list);
            HIPC(hipGetLastError());
            HIPC(hipMemsetAsync(d_arrivals, 0, (size_t)n_inner * sizeof(uint32_t), stream));
            HIPC(hipMemsetAsync(d_done2, 0, (size_t)n_inner * sizeof(uint32_t), stream));
            uint32_t* done_in = d_arrivals;
            uint32_t* done_out = d_done2;
            uint32_t remaining = m;
            for (uint32_t round = 0; remaining != 0u; ++round) {
                if (round >= 160u) return fail(TRT_EHIP, "trt_build_lbvh: internal error (the spanning nodes do not finish)");
                // eight rounds between two looks at the counter (a tree of 10 M triangles needs about 25)
                for (int k = 0; k < 8; ++k) {
                    HIPC(hipMemsetAsync(d_depth, 0, sizeof(uint32_t), stream));
                    hipLaunchKernelGGL(k_span_round, dim3((m + 255u) / 256u), dim3(256), 0, stream, d_list, m, d_left, d_right, d_first, d_last, d_pbox, d_order, d_nbox, done_in, done_out, d_depth);
                    HIPC(hipGetLastError());
                    std::swap(done_in, done_out);
                }
                round += 7u;
                HIPC(hipMemcpy(&remaining, d_depth, sizeof(uint32_t), hipMemcpyDeviceToHost));
            }
        }
    }
    // ---- the top of the tree by SAH over clusters of the radix tree (TRT_LBVH_CLUSTER triangles at most; 0: the radix tree as it is)
    // Largest cluster, by node visits per ray against the host SAH tree (round 4's sweeps: tools/lbvh_cluster_sweep.py, profiles/r04_lbvh_quality.txt; the CPU
    // emulation of tools/lbvh_study.py agrees to the point):
    //   below 50 k triangles    2      Morton PAIRS under a full SAH: staircase +8.0 % (round 3's n / 64 = 490: +20.7 %), veach-mis +12.6 % (+22.2 %), soup-50k
    //                                  +1.3 % (+12.0 %).  On these scenes the loss sits between groups of 8 and groups of 16 Morton neighbours (staircase: 8 -> +10.4 %,
    //                                  12 -> +9.9 %, 16 -> +22.6 %, and SAH splits INSIDE the groups of 16 do not bring it back); the host's SAH over n / 2 clusters is ~10 ms
    //   below 500 k             16     blob-60k +2.6 % (+14.4 %), blob-150k +1.2 % (+11.7 % with round 3's 2048)
    //   below 4 M               128    blob-2M +2.9 % (+7.4 %); soup-1M +5.2 %, the same as with 2048
    //   above                   2048   blob-10M +5.1 %; smaller clusters are erratic there (512: +7.1 %, 64: +3.3 %) and the host's SAH over 150 k clusters takes 160 ms
    uint32_t cluster = n < 50000u ? 2u : (n < 500000u ? 16u : (n < 4000000u ? 128u : 2048u));
    if (const char* e = std::getenv("TRT_LBVH_CLUSTER")) cluster = (uint32_t)std::max(0L, std::atol(e));
    if (cluster && cluster < (uint32_t)leaf_num) cluster = (uint32_t)leaf_num;
    uint32_t n_out = 0, depth = 0;
    if (cluster && n > cluster) {
        uint32_t *d_start = nullptr, *d_root_at = nullptr, *d_incl = nullptr;
        HIPC(mem.alloc(&d_start, n));
        HIPC(mem.alloc(&d_root_at, n));
        HIPC(mem.alloc(&d_incl, n));
        HIPC(hipMemsetAsync(d_start, 0, (size_t)n * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(k_cluster_starts, dim3(grid_n), dim3(256), 0, stream, d_first, d_last, d_parent, d_leaf_parent, n, cluster, d_start, d_root_at);
        HIPC(hipGetLastError());
        {
            size_t tmp_bytes = 0;
            HIPC(rocprim::inclusive_scan(nullptr, tmp_bytes, d_start, d_incl, (size_t)n, rocprim::plus<uint32_t>(), stream));
            char* d_tmp = nullptr;
            HIPC(mem.alloc(&d_tmp, tmp_bytes));
            HIPC(rocprim::inclusive_scan(d_tmp, tmp_bytes, d_start, d_incl, (size_t)n, rocprim::plus<uint32_t>(), stream));
        }
        uint32_t n_clusters = 0;
        HIPC(hipMemcpy(&n_clusters, d_incl + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (n_clusters < 2 || n_clusters > n) return fail(TRT_EHIP, "trt_build_lbvh: internal error (cluster count)");
        ClusterRec* d_rec = nullptr;
        uint32_t *d_croot = nullptr, *d_cidx = nullptr, *d_cdepth = nullptr, *d_order2 = nullptr;
        int32_t* d_shift = nullptr;
        HIPC(mem.alloc(&d_rec, n_clusters));
        HIPC(mem.alloc(&d_croot, n_clusters));
        HIPC(mem.alloc(&d_cidx, n_clusters));
        HIPC(mem.alloc(&d_cdepth, n_clusters));
        HIPC(mem.alloc(&d_shift, n_clusters));
        HIPC(mem.alloc(&d_order2, n));
        hipLaunchKernelGGL(k_cluster_gather, dim3(grid_n), dim3(256), 0, stream, d_start, d_incl, d_root_at, d_first, d_last, d_pbox, d_order, d_nbox, n, d_rec, d_croot);
        HIPC(hipGetLastError());
        std::vector<ClusterRec> rec(n_clusters);
        HIPC(hipMemcpy(rec.data(), d_rec, (size_t)n_clusters * sizeof(ClusterRec), hipMemcpyDeviceToHost));
        uint64_t covered = 0;
        for (const ClusterRec& r : rec) covered += r.count;
        if (covered != n) return fail(TRT_EHIP, "trt_build_lbvh: internal error (clusters do not cover the triangles)");
        TopBuilder top(rec);
        float blo[3], bhi[3];
        if (top.build(0, n_clusters, 0, blo, bhi) != 0) return fail(TRT_EHIP, "trt_build_lbvh: internal error (top tree)");
        const uint32_t T = (uint32_t)top.nodes.size();
        std::vector<int32_t> shift(n_clusters);
        std::vector<uint32_t> new_first(n_clusters);
        {
            uint64_t at = 0;
            for (uint32_t c : top.leaf_order) {
                new_first[c] = (uint32_t)at;
                shift[c] = (int32_t)((int64_t)at - (int64_t)rec[c].first);
                at += rec[c].count;
            }
        }
        HIPC(hipMemcpy(d_shift, shift.data(), (size_t)n_clusters * sizeof(int32_t), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_survive_in_clusters, dim3(grid_i), dim3(256), 0, stream, d_first, d_last, n_inner, (uint32_t)leaf_num, cluster, d_keep);
        HIPC(hipGetLastError());
        {
            size_t tmp_bytes = 0;
            HIPC(rocprim::exclusive_scan(nullptr, tmp_bytes, d_keep, d_new, 0u, (size_t)n_inner, rocprim::plus<uint32_t>(), stream));
            char* d_tmp = nullptr;
            HIPC(mem.alloc(&d_tmp, tmp_bytes));
            HIPC(rocprim::exclusive_scan(d_tmp, tmp_bytes, d_keep, d_new, 0u, (size_t)n_inner, rocprim::plus<uint32_t>(), stream));
        }
        uint32_t tail[2] = {0, 0};
        HIPC(hipMemcpy(&tail[0], d_new + (n_inner - 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(&tail[1], d_keep + (n_inner - 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
        const uint32_t n_in = tail[0] + tail[1];
        if ((uint64_t)T + n_in > n_inner) return fail(TRT_EHIP, "trt_build_lbvh: internal error (node count)");
        n_out = T + n_in;
        if (n_out > node_capacity) return fail(TRT_EINVAL, "trt_build_lbvh: node_capacity too small (n_tris - 1 always suffices)");
        trt_bvh_node* d_out = nullptr;
        HIPC(mem.alloc(&d_out, n_in));
        hipLaunchKernelGGL(k_emit_clusters, dim3(grid_i), dim3(256), 0, stream, d_pbox, d_order, d_nbox, d_left, d_right, d_first, d_last, d_keep, d_new, d_incl, d_shift, n_inner, T, d_out);
        HIPC(hipGetLastError());
        const uint32_t grid_c = (n_clusters + 255u) / 256u;
        hipLaunchKernelGGL(k_cluster_root_index, dim3(grid_c), dim3(256), 0, stream, d_croot, d_keep, d_new, n_clusters, d_cidx);
        HIPC(hipGetLastError());
        hipLaunchKernelGGL(k_permute_order, dim3(grid_n), dim3(256), 0, stream, d_order, d_incl, d_shift, n, d_order2);
        HIPC(hipGetLastError());
        HIPC(hipMemsetAsync(d_cdepth, 0, (size_t)n_clusters * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(k_cluster_depth, dim3(grid_n), dim3(256), 0, stream, d_parent, d_leaf_parent, d_keep, d_incl, n, d_cdepth);
        HIPC(hipGetLastError());
        HIPC(hipEventRecord(ev1, stream));
        std::vector<uint32_t> cidx(n_clusters), cdepth(n_clusters);
        HIPC(hipMemcpy(cidx.data(), d_cidx, (size_t)n_clusters * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(cdepth.data(), d_cdepth, (size_t)n_clusters * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (n_in) HIPC(hipMemcpy(nodes_out + T, d_out, (size_t)n_in * sizeof(trt_bvh_node), hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(order_out, d_order2, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint32_t k = 0; k < T; ++k) {
            const TopNode& t = top.nodes[k];
            trt_bvh_node nd;
            std::memset(&nd, 0, sizeof(nd));
            for (int side = 0; side < 2; ++side) {
                uint32_t ref;
                if (t.child[side] >= 0) {
                    ref = (uint32_t)t.child[side];
                } else {
                    const uint32_t c = (uint32_t)~t.child[side];
                    ref = cidx[c] != 0xFFFFFFFFu ? T + cidx[c] : TRT_MAKE_LEAF(new_first[c], rec[c].count);
                }
                float* lo = side ? nd.lo1 : nd.lo0;
                float* hi = side ? nd.hi1 : nd.hi0;
                for (int a = 0; a < 3; ++a) { lo[a] = t.lo[side][a] - PAD; hi[a] = t.hi[side][a] + PAD; }
                if (side) nd.child1 = ref; else nd.child0 = ref;
            }
            nodes_out[k] = nd;
        }
        for (uint32_t c = 0; c < n_clusters; ++c) depth = std::max(depth, top.top_depth[c] + cdepth[c]);
    } else {
        hipLaunchKernelGGL(k_survive, dim3(grid_i), dim3(256), 0, stream, d_first, d_last, n_inner, (uint32_t)leaf_num, d_keep);
        HIPC(hipGetLastError());
        {
            size_t tmp_bytes = 0;
            HIPC(rocprim::exclusive_scan(nullptr, tmp_bytes, d_keep, d_new, 0u, (size_t)n_inner, rocprim::plus<uint32_t>(), stream));
            char* d_tmp = nullptr;
            HIPC(mem.alloc(&d_tmp, tmp_bytes));
            HIPC(rocprim::exclusive_scan(d_tmp, tmp_bytes, d_keep, d_new, 0u, (size_t)n_inner, rocprim::plus<uint32_t>(), stream));
        }
        uint32_t tail[2] = {0, 0};
        HIPC(hipMemcpy(&tail[0], d_new + (n_inner - 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(&tail[1], d_keep + (n_inner - 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
        n_out = tail[0] + tail[1];
        if (n_out < 1 || n_out > n_inner) return fail(TRT_EHIP, "trt_build_lbvh: internal error (node count)");
        if (n_out > node_capacity) return fail(TRT_EINVAL, "trt_build_lbvh: node_capacity too small (n_tris - 1 always suffices)");
        trt_bvh_node* d_out = nullptr;
        HIPC(mem.alloc(&d_out, n_out));
        hipLaunchKernelGGL(k_emit, dim3(grid_i), dim3(256), 0, stream, d_pbox, d_order, d_nbox, d_left, d_right, d_first, d_last, d_keep, d_new, n_inner, d_out);
        HIPC(hipGetLastError());
        HIPC(hipMemsetAsync(d_depth, 0, sizeof(uint32_t), stream));
        hipLaunchKernelGGL(k_depth, dim3(grid_n), dim3(256), 0, stream, d_parent, d_leaf_parent, d_keep, n, d_depth);
        HIPC(hipGetLastError());
        HIPC(hipEventRecord(ev1, stream));
        HIPC(hipMemcpy(nodes_out, d_out, (size_t)n_out * sizeof(trt_bvh_node), hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(order_out, d_order, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(&depth, d_depth, sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    *n_nodes_out = n_out;
    if (depth_out) *depth_out = depth;
    if (ms_out) {
        float ms = 0.0f;
        HIPC(hipEventElapsedTime(&ms, ev0, ev1));
        ms_out[0] = ms;
        ms_out[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host).count();
    }
    return TRT_OK;
}

}  // extern "C"
