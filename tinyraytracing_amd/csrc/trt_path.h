// trt_path.h — per-lane device functions of the path-tracing hot path:
// BVH traversal, ray/triangle and ray/box tests, path-vertex set-up, light
// sampling (NEE), Russian roulette and BSDF sampling.  The wavefront kernels
// in trt_kernels.h are thin drivers around these.
//
// Every function is `TRT_HD` and free of device-only intrinsics, so the same
// source also compiles with g++ for tests/hostsim (a CPU check of this file's
// arithmetic against the oracle; test infrastructure, not a product path).
//
// Arithmetic contract (include/trt_prims.h, DESIGN.md "Formulation"): fp32,
// compiled with -ffp-contract=off, FMA only where written.  Reference
// citations are file:line under RayTracingOnCPU/.
#pragma once
#include <stdint.h>
#include <math.h>

#include "trt.h"
#include "trt_prims.h"
#include "trt_exact.h"

#ifndef TRT_PREFETCH
#define TRT_PREFETCH 1
#endif

namespace trtd {

// ------------------------------------------------------------------ types ----
struct alignas(16) f4 {
    float x, y, z, w;
};
struct f3 {
    float x, y, z;
};

TRT_HD inline f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
TRT_HD inline f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
TRT_HD inline f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
TRT_HD inline f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
TRT_HD inline f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
TRT_HD inline f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
TRT_HD inline f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
TRT_HD inline f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
TRT_HD inline f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
// glm evaluation order: dot = (x + y) + z; normalize = v * (1 / sqrt(dot(v, v)))
TRT_HD inline float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
TRT_HD inline f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
// sqrtf(x) and 1.0f / sqrtf(x) through trt_exact.h: the same bits (proven over all 2^32 inputs) in a third of the instructions
TRT_HD inline float length(f3 a) { return trt_sqrt(dot(a, a)); }
TRT_HD inline f3 normalize(f3 a) { return a * trt_rsqrt2(dot(a, a)); }
// glm::reflect / glm::refract (pathTracing.cpp:177,184,202)
TRT_HD inline f3 reflect(f3 I, f3 N) { return I - (N * dot(N, I)) * 2.0f; }
TRT_HD inline f3 refract(f3 I, f3 N, float eta)
{
    const float dn = dot(N, I);
    const float k = 1.0f - eta * eta * (1.0f - dn * dn);
    if (k < 0.0f) return mk3(0.f, 0.f, 0.f);
    return I * eta - N * (eta * dn + trt_sqrt(k));
}

TRT_HD inline uint32_t f2u(float f) { return trt_f2u(f); }
TRT_HD inline float u2f(uint32_t u) { return trt_u2f(u); }

// ------------------------------------------------------- device scene view ----
// 48-B intersection record per triangle (post-BVH order), three 16-B loads:
//   a = (v0.x, v0.y, v0.z, e1.x)  b = (e1.y, e1.z, e2.x, e2.y)
//   c = (e2.z, tol, bits(flags), 0)     tol = 1e-5 * |e1 x e2|
//   flags: bit 0 = material is emissive (Triangle::is_emissive), bits 8.. = material id
struct TriIsect {
    f4 a, b, c;
};
// 64-B shading record: vertex normals, texture coordinates, material id
struct alignas(16) TriShade {
    float vn[9];
    float vt[6];
    int32_t mat;
};
// Table records are padded to multiples of 16 B and copied whole, so a lane fetches one with a few
// 16-B loads instead of one dword load per field (k_shade is bound by load-instruction issue otherwise).
struct alignas(16) MaterialDev {
    float Kd[3], Ks[3], Tr[3];
    float Ns, Ni;
    float radiance[3];
    int32_t is_emissive;
    int32_t tex;  // -1 = none
    // Values shade() recomputes at every vertex although they depend on the material alone; formed once here
    // by the very operations of the per-vertex code (IEEE division / square root: same bits on host and device):
    float Kd_pi[3];   // Kd / pi                      (pathTracing.cpp:67, untextured materials)
    float sel_kd;     // |Kd| / (|Kd| + |Ks|)          (lobe selection, pathTracing.cpp:188-191)
    float sel_kdks;   // sel_kd + |Ks| / (|Kd| + |Ks|)
    float rf0;        // ((n1 - n2) / (n1 + n2))^2     (Schlick, pathTracing.cpp:157-160; the same either way)
    float Ni_inv;     // 1 / Ni
    int32_t no_spec;  // the Phong term of lightSample() is exactly zero for this material and need not be evaluated
};
static_assert(sizeof(MaterialDev) == 96, "MaterialDev must be six 16-byte words");
// trt_light_tri (19 floats) padded to 80 B
struct alignas(16) LightTriDev {
    float v[3][3];
    float vn[3][3];
    float cum_area;
    float pad;
};
static_assert(sizeof(LightTriDev) == 80, "LightTriDev must be five 16-byte words");
// trt_light (7 words) padded to 32 B
struct alignas(16) LightDev {
    int32_t mat;
    float radiance[3];
    float area;
    uint32_t tri_first, tri_count;
    float pdf;  // 1 / area (pathTracing.cpp:63), formed once
};
static_assert(sizeof(LightDev) == 32, "LightDev must be two 16-byte words");
struct TextureDev {
    int32_t width, height;
    uint64_t offset;  // into tex_bytes
};

// 4-wide inner node, 128 B = eight 16-B loads, children side by side (one float4 per box plane):
//   q[0..2] = lo.x, lo.y, lo.z of children 0..3    q[3..5] = hi.x, hi.y, hi.z
//   q[6]    = child references (trt.h encoding; an inner reference indexes wnodes)    q[7] unused
// An unused slot holds an all-NaN box (fails every slab test) and TRT_WIDE_EMPTY.  Built from the
// caller's BVH2 by collapseBvh (trt_wide.h), which also says why the closest hit cannot change.
#define TRT_WIDE 4
#define TRT_WIDE_EMPTY 0xFFFFFFFFu
struct WideNode {
    f4 q[8];
};

struct OctNode;  // trt_oct.h
struct SceneDev {
    const trt_bvh_node* nodes;   // the caller's BVH2: the wave-uniform walk of tiny trees (trt_kernels.h, IMPL 0)
    const WideNode* wnodes;      // its 4-wide collapse, exact boxes (NK = 0; always present: the exact form of k_trace_fix / k_tail walks it)
    const OctNode* onodes;       // its 8-wide collapse, quantised conservative boxes (NK = 1, trt_oct.h); null when the tree does not qualify
    const TriIsect* tri_trav;    // NK = 1: the triangle records in the order the oct nodes address them
    const f4* leaf_box;          // for every triangle i the caller's box of its leaf at [2 i] = (lo.xyz, hi.x), [2 i + 1] = (hi.y, hi.z, -, -): leafEntry()
    const TriIsect* tri_isect;
    const TriShade* tri_shade;
    const MaterialDev* materials;
    const LightDev* lights;
    const LightTriDev* light_tris;
    const float* light_cum;  // light_tris[k].cum_area packed (the CDF of pathTracing.cpp:40); null = scan the structs
    const TextureDev* textures;
    const uint8_t* tex_bytes;
    uint32_t n_tris, n_nodes, n_wnodes, n_onodes, n_lights;
    uint32_t refill_min;  // persistent traversal drivers: lanes to have free before a refill (trt_kernels.h)
    uint32_t sched_in_w, sched_lf_w;  // scheduler driver: node step iff sched_in_w * (lanes at nodes) >= sched_lf_w * (lanes at leaves)
    uint32_t leaf_loop;   // oct driver: triangles a lane tests per leaf step (trt_kernels.h; 2 for trees with leaves of <= 3, more where the caller's leaves are larger)
    float light0_area;  // Q3: every light's CDF draw spans lights[0].area (pathTracing.cpp:38)
    float leaf_alpha;   // absolute part of the leaf-box rule's tolerance (trt_leaf_floor, trt_prims.h): sceneLeafAlpha()
    const uint32_t* plane_bits;  // one-hash Bloom filter over every box plane of the caller's tree, keyed (axis, coordinate): planeMaybe() — null = "maybe" for every query
    uint32_t plane_shift;        // 32 - log2(bits of the filter)
    float cull_alpha;   // what trt_cull_bound() gets on the 4-wide nodes: leaf_alpha where the boxes of the tree nest, +inf (never cull by distance) where they do not (trt_wide.h boxesNested)
    trt_camera cam;
};

// alpha of the leaf-box rule (trt_prims.h) for a tree: 2^-17 of the largest |coordinate| of the root's two child boxes.
// The same function on every side (trt_create, the hostsim, the oracle restates it), NaNs ignored.
TRT_HD inline float sceneLeafAlpha(const trt_bvh_node* nodes, uint32_t n_nodes)
{
    if (!n_nodes) return 0.0f;
    const trt_bvh_node& r = nodes[0];
    float m = 0.0f;
    for (int a = 0; a < 3; ++a) m = fmaxf(m, fmaxf(fmaxf(fabsf(r.lo0[a]), fabsf(r.hi0[a])), fmaxf(fabsf(r.lo1[a]), fabsf(r.hi1[a]))));
    return trt_leaf_alpha(m);
}

struct Hit {
    float t;         // TRT_INF on a miss (HitRecord::distance default, bvh.h:10)
    int32_t tri;     // -1 on a miss
    float u, v;      // barycentric weights of v1, v2
    uint32_t flags;  // TriIsect flags of the hit triangle: bit 0 emissive, bits 8.. material id
};

TRT_HD inline MaterialDev makeMaterialDev(const trt_material& m)
{
    MaterialDev d;
    for (int k = 0; k < 3; ++k) { d.Kd[k] = m.Kd[k]; d.Ks[k] = m.Ks[k]; d.Tr[k] = m.Tr[k]; d.radiance[k] = m.radiance[k]; }
    d.Ns = m.Ns; d.Ni = m.Ni;
    d.is_emissive = m.is_emissive; d.tex = m.tex;
    for (int k = 0; k < 3; ++k) d.Kd_pi[k] = m.Kd[k] / TRT_PI;
    const float Kd_len = length(ld3(m.Kd)), Ks_len = length(ld3(m.Ks));
    const float kd = Kd_len / (Kd_len + Ks_len), ks = Ks_len / (Kd_len + Ks_len);
    d.sel_kd = kd;
    d.sel_kdks = kd + ks;
    const float q = (1.0f - m.Ni) / (1.0f + m.Ni);  // (Ni - 1) / (Ni + 1) is its exact negative: one square serves both sides
    d.rf0 = q * q;
    d.Ni_inv = 1.0f / m.Ni;
    // Ks == 0 (either sign) makes the term +-0 whenever pow01 is finite, i.e. for 0 < Ns < inf; adding -0 never changes
    // kd_pi, adding +0 changes only a -0 component: excluded.  Textured materials (kd_pi from texels, >= +0) qualify too.
    bool no_spec = m.Ns > 0.0f && m.Ns < 3.0e38f;
    for (int k = 0; k < 3; ++k) {
        const bool ks_pos_zero = f2u(m.Ks[k]) == 0u, ks_neg_zero = f2u(m.Ks[k]) == 0x80000000u;
        no_spec = no_spec && (ks_neg_zero || (ks_pos_zero && f2u(d.Kd_pi[k]) != 0x80000000u));
    }
    d.no_spec = no_spec ? 1 : 0;
    return d;
}
TRT_HD inline LightTriDev makeLightTriDev(const trt_light_tri& t)
{
    LightTriDev d;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) { d.v[a][b] = t.v[a][b]; d.vn[a][b] = t.vn[a][b]; }
    d.cum_area = t.cum_area;
    d.pad = 0.0f;
    return d;
}
// The radiance of the direct term is the light MATERIAL's (pathTracing.cpp:65 reads scene.materials[light_triangle.mtl_name].radiance), not trt_light::radiance:
// the loaders write the same three numbers into both (scene.cpp:50-52), a caller of the C-ABI who does not gets what the reference would compute.
TRT_HD inline LightDev makeLightDev(const trt_light& l, const trt_material* materials)
{
    LightDev d;
    d.mat = l.mat;
    for (int k = 0; k < 3; ++k) d.radiance[k] = materials[l.mat].radiance[k];
    d.area = l.area; d.tri_first = l.tri_first; d.tri_count = l.tri_count; d.pdf = 1.0f / l.area;
    return d;
}

// Build-time helper shared by trt_create and the hostsim: the 48-B record of one triangle.
TRT_HD inline TriIsect makeTriIsect(const float* v9, int32_t mat, bool emissive)
{
    const f3 v0 = ld3(v9), e1 = ld3(v9 + 3) - v0, e2 = ld3(v9 + 6) - v0;
    const f3 g = cross(e1, e2);
    const float tol = TRT_PARALLEL_EPS * sqrtf(dot(g, g));
    TriIsect r;
    r.a = mk4(v0.x, v0.y, v0.z, e1.x);
    r.b = mk4(e1.y, e1.z, e2.x, e2.y);
    r.c = mk4(e2.z, tol, u2f(((uint32_t)mat << 8) | (emissive ? 1u : 0u)), 0.0f);
    return r;
}

// ------------------------------------------------ interactTriangle (a6) ----
// bvh.cpp:177-209 in Moller-Trumbore form: the parallel cut
// |N.d| < 1e-5 is |det| < tol, t < 0.0005 misses, and "strictly inside"
// (bvh.cpp:196-198) is u > 0, v > 0, u + v < 1.  Evaluated without early
// exits (every lane of a wave runs the same instructions; a miss is a mask),
// with det made positive by flipping the signs of all four scalars.  Returns
// true for a hit; t = tn/det is formed only then, and the barycentrics
// u = un/det, v = vn/det are left to the caller (needed once per ray).
// triCandidate: everything up to the acceptance test; tn, un, vn, det are the numerators / the denominator
// (det > 0) of t, u, v.  triTest = triCandidate + the division + the t < 0.0005 cut.
// The two halves of the acceptance test are returned separately (ok_det: not parallel; ok_in: strictly inside): a wave vote on
// each is the compare's own lane mask, a vote on their conjunction is two more VALU instructions (trt_kernels.h, ballotb()).
TRT_HD inline void triCandidateParts(const TriIsect& T, f3 o, f3 d, float& tn_out, float& un_out, float& vn_out, float& det_out, bool& ok_det, bool& ok_in)
{
    const float v0x = T.a.x, v0y = T.a.y, v0z = T.a.z;
    const float e1x = T.a.w, e1y = T.b.x, e1z = T.b.y;
    const float e2x = T.b.z, e2y = T.b.w, e2z = T.c.x;
    const float px = fmaf(d.y, e2z, -(d.z * e2y));
    const float py = fmaf(d.z, e2x, -(d.x * e2z));
    const float pz = fmaf(d.x, e2y, -(d.y * e2x));
    const float det_s = fmaf(e1z, pz, fmaf(e1y, py, e1x * px));
    const float tx = o.x - v0x, ty = o.y - v0y, tz = o.z - v0z;
    const float un_s = fmaf(tz, pz, fmaf(ty, py, tx * px));
    const float qx = fmaf(ty, e1z, -(tz * e1y));
    const float qy = fmaf(tz, e1x, -(tx * e1z));
    const float qz = fmaf(tx, e1y, -(ty * e1x));
    const float vn_s = fmaf(d.z, qz, fmaf(d.y, qy, d.x * qx));
    const float tn_s = fmaf(e2z, qz, fmaf(e2y, qy, e2x * qx));
    const uint32_t sgn = f2u(det_s) & 0x80000000u;  // exact negation of all four when det < 0
    const float det = u2f(f2u(det_s) ^ sgn), un = u2f(f2u(un_s) ^ sgn), vn = u2f(f2u(vn_s) ^ sgn), tn = u2f(f2u(tn_s) ^ sgn);
    // bvh.cpp:185,196-198: !(det < tol) && un > 0 && vn > 0 && un + vn < det.  The three "> 0" tests as one
    // integer comparison: x > 0 (NaN: false, +inf: true) <=> bits(x) - 1 < 0x7F800000 unsigned, and
    // un + vn < det <=> det - (un + vn) > 0 (exact with gradual underflow; inf - inf = NaN is false on both sides).
    const float rest = det - (un + vn);
    const uint32_t bu = f2u(un) - 1u, bv = f2u(vn) - 1u, br = f2u(rest) - 1u;
    const uint32_t bmax = bu > bv ? (bu > br ? bu : br) : (bv > br ? bv : br);
    tn_out = tn;
    un_out = un;
    vn_out = vn;
    det_out = det;
    ok_det = !(det < T.c.y);
    ok_in = bmax < 0x7F800000u;
}
TRT_HD inline bool triCandidate(const TriIsect& T, f3 o, f3 d, float& tn_out, float& un_out, float& vn_out, float& det_out)
{
    bool ok_det, ok_in;
    triCandidateParts(T, o, d, tn_out, un_out, vn_out, det_out, ok_det, ok_in);
    return ok_det && ok_in;
}
TRT_HD inline bool triTest(const TriIsect& T, f3 o, f3 d, float& t_out, float& un_out, float& vn_out, float& det_out)
{
    float tn;
    if (!triCandidate(T, o, d, tn, un_out, vn_out, det_out)) return false;
    const float t = tn / det_out;
    if (t < TRT_T_MIN) return false;  // bvh.cpp:189
    t_out = t;
    return true;
}

// ---------------------------------------------------- interactAABB (a4) ----
// bvh.cpp:231-245: slab test; `entry` receives t0 (used for ordering/culling).
// fminf/fmaxf differ from glm's ternaries only for NaN operands (an axis with
// d == 0 and the origin exactly on the padded plane).
TRT_HD inline bool slabResult(float inx, float iny, float inz, float outx, float outy, float outz, float& entry)
{
    const float t1 = fminf(fmaxf(inx, outx), fminf(fmaxf(iny, outy), fmaxf(inz, outz)));
    const float t0 = fmaxf(fminf(inx, outx), fmaxf(fminf(iny, outy), fminf(inz, outz)));
    entry = t0;
    // bvh.cpp:243-244 returns r = (t1 >= t0) ? ((t0 > 0) ? t0 : t1) : -1 and the caller descends iff r > 0 (bvh.cpp:162-166).
    // With t1 >= t0 true (so neither is NaN): t0 > 0 gives r = t0 > 0, and then t1 >= t0 > 0 as well; otherwise r = t1.  Either
    // way r > 0 <=> t1 > 0, so the decision is (t1 >= t0) && (t1 > 0) for every input — two compares, no select.
    return (t1 >= t0) && (t1 > 0.0f);
}
TRT_HD inline bool boxTest(float lox, float loy, float loz, float hix, float hiy, float hiz, f3 o, f3 inv, float& entry)
{
    const float inx = (hix - o.x) * inv.x, iny = (hiy - o.y) * inv.y, inz = (hiz - o.z) * inv.z;
    const float outx = (lox - o.x) * inv.x, outy = (loy - o.y) * inv.y, outz = (loz - o.z) * inv.z;
    return slabResult(inx, iny, inz, outx, outy, outz, entry);
}

// ---- the slab test LITERALLY, for the rays on which the two forms can differ ----
// glm::min(x, y) = (y < x) ? y : x and glm::max(x, y) = (x < y) ? y : x (bvh.cpp:238-242) hand a NaN on from their FIRST operand only; fminf / fmaxf (v_min / v_max)
// drop it from either side.  A NaN arises in the slab test of finite boxes in one way: 0 * inf — a direction component whose reciprocal is infinite (the component
// is zero or below 2.9e-39) on a ray whose origin lies EXACTLY on a plane of the box.  The reference then loses that axis's constraint AND, by the operand order of
// bvh.cpp:241-242, those of the axes nested inside the same min / max: it enters boxes the clean test rejects, and finds hits there (a camera on a vertex of a mesh
// at coordinates where the 0.001 pad is below one ulp: tools/fuzz_scenes.py found it).  Rays with such a direction (raySpecial) are therefore walked by
// traceClosestBvh2Glm(): the caller's BVH2 as bvh.cpp:146-175 walks it, this test, no culling.  Every other ray cannot produce a NaN from finite operands and keeps
// the fast forms, which agree with this one on all non-NaN inputs.
TRT_HD inline float glmMin(float x, float y) { return (y < x) ? y : x; }
TRT_HD inline float glmMax(float x, float y) { return (x < y) ? y : x; }
TRT_HD inline bool boxTestGlm(float lox, float loy, float loz, float hix, float hiy, float hiz, f3 o, f3 inv, float& entry)
{
    const float inx = (hix - o.x) * inv.x, iny = (hiy - o.y) * inv.y, inz = (hiz - o.z) * inv.z;
    const float outx = (lox - o.x) * inv.x, outy = (loy - o.y) * inv.y, outz = (loz - o.z) * inv.z;
    const float tmaxx = glmMax(inx, outx), tmaxy = glmMax(iny, outy), tmaxz = glmMax(inz, outz);
    const float tminx = glmMin(inx, outx), tminy = glmMin(iny, outy), tminz = glmMin(inz, outz);
    const float t1 = glmMin(tmaxx, glmMin(tmaxy, tmaxz));
    const float t0 = glmMax(tminx, glmMax(tminy, tminz));
    entry = t0;
    const float r = (t1 >= t0) ? ((t0 > 0.0f) ? t0 : t1) : -1.0f;  // bvh.cpp:244
    return r > 0.0f;                                                // bvh.cpp:162-166
}
// a reciprocal direction with a component that is infinite or NaN
TRT_HD inline bool raySpecial(f3 inv) { return !(fabsf(inv.x) <= 3.4028235e38f && fabsf(inv.y) <= 3.4028235e38f && fabsf(inv.z) <= 3.4028235e38f); }
// Even such a ray meets a NaN only if its origin lies EXACTLY on a plane of some box, on an axis whose reciprocal is infinite: (plane - o) * inf is NaN for plane == o
// alone.  A one-hash Bloom filter over all box planes of the tree (built by trt_create: planeFilterBuild, trt_wide.h) answers "certainly not" for almost every origin;
// only the rays it cannot clear pay for the unculled walk (which on a million-triangle soup visits tens of thousands of nodes).  No false "no": -0 and +0 share a key.
TRT_HD inline uint32_t planeKey(int axis, float x) { return (f2u(x + 0.0f) * 2654435761u) ^ ((uint32_t)(axis + 1) * 0x9E3779B9u); }
TRT_HD inline bool planeMaybe(const SceneDev& sc, int axis, float x)
{
    if (!sc.plane_bits) return true;
    const uint32_t h = (planeKey(axis, x) * 2246822519u) >> sc.plane_shift;
    return (sc.plane_bits[h >> 5] >> (h & 31u)) & 1u;
}
TRT_HD inline bool rayOnABoxPlane(const SceneDev& sc, f3 o, f3 inv)
{
    return (!(fabsf(inv.x) <= 3.4028235e38f) && planeMaybe(sc, 0, o.x)) || (!(fabsf(inv.y) <= 3.4028235e38f) && planeMaybe(sc, 1, o.y)) ||
           (!(fabsf(inv.z) <= 3.4028235e38f) && planeMaybe(sc, 2, o.z));
}

// ------------------------------------------------------ one inner-node step ----
// Children of wide node `cur` that the ray can still improve on: box passed (bvh.cpp:162-166) and entry
// not STRICTLY beyond the best hit.  Continues with the nearest (returns true, `cur` updated), the others
// go on the stack, nearest on top.  The visiting order is free (see traceClosest); only the set matters.
#define TRT_CSWAP(ka, ra, kb, rb)                                   \
    {                                                               \
        const bool sw_ = kb < ka;                                   \
        const float tk_ = sw_ ? kb : ka;                            \
        const uint32_t tr_ = sw_ ? rb : ra;                         \
        kb = sw_ ? ka : kb;                                         \
        rb = sw_ ? ra : rb;                                         \
        ka = tk_;                                                   \
        ra = tr_;                                                   \
    }
// Entry distance of the box of the leaf triangle `tri` lies in (the caller's box of that leaf, leaf_box).  A triangle hit counts
// only if it does not lie IN FRONT of its leaf's box by more than a tolerance: NOT t < trt_leaf_floor(entry, alpha) (trt_prims.h).
// For a ray within ~1e-4 rad of a triangle's plane the Moller-Trumbore distance tn / det can come out well in front of the
// triangle (its barycentrics are computed independently and still say "inside"), i.e. outside every box that contains it.  The
// reference never produces such a hit — its inside test is applied to the computed point P = o + d t (bvh.cpp:191-198) — and
// the rule is what makes culling exact: every hit that counts has t >= floor(entry of its leaf's box) >= floor(entry of any box
// above it) (nested boxes, monotone slab arithmetic, monotone floor), so skipping a node whose entry lies beyond
// trt_cull_bound(best hit) can never skip a hit that would have beaten or tied it.  The tolerance keeps the rule away from honest
// hits: a triangle ON a face of its leaf's box has entry and tn / det equal up to rounding, and a bare t < entry threw half of
// those away (unpadded foreign trees; coordinates of 4e4, where the reference's 0.001 pad is below one ulp).
TRT_HD inline float leafEntry(const SceneDev& sc, uint32_t tri, f3 o, f3 inv)
{
    const f4 a = sc.leaf_box[2 * (size_t)tri], b = sc.leaf_box[2 * (size_t)tri + 1];
    float e;
    (void)boxTest(a.x, a.y, a.z, a.w, b.x, b.y, o, inv, e);
    return e;
}
// the distance below which a hit on triangle `tri` does not count
TRT_HD inline float leafFloor(const SceneDev& sc, uint32_t tri, f3 o, f3 inv) { return trt_leaf_floor(leafEntry(sc, tri, o, inv), sc.leaf_alpha); }

// `cull_t` = trt_cull_bound(best hit so far): a child whose entry lies beyond it holds nothing that could still count.
template <class Stack>
TRT_HD inline bool innerStep(const SceneDev& sc, uint32_t& cur, int& sp, Stack& stk, f3 o, f3 inv, float cull_t)
{
    float e0, e1, e2, e3;
    bool h0, h1, h2, h3;
    uint32_t r0, r1, r2, r3;
    const f4* np4 = sc.wnodes[cur].q;
    const f4 lx = np4[0], ly = np4[1], lz = np4[2], hx = np4[3], hy = np4[4], hz = np4[5], rf = np4[6];
#if defined(__HIP_DEVICE_COMPILE__)
    // The 24 subtractions and 24 multiplications of four slab tests as packed-fp32 pairs (v_pk_add_f32 /
    // v_pk_mul_f32: two IEEE operations per lane and instruction, children (0,1) and (2,3) side by side);
    // every element is computed by the same operation as in boxTest().
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z}, ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
    const v2f inx_a = (v2f{hx.x, hx.y} - ox) * ix, inx_b = (v2f{hx.z, hx.w} - ox) * ix;
    const v2f iny_a = (v2f{hy.x, hy.y} - oy) * iy, iny_b = (v2f{hy.z, hy.w} - oy) * iy;
    const v2f inz_a = (v2f{hz.x, hz.y} - oz) * iz, inz_b = (v2f{hz.z, hz.w} - oz) * iz;
    const v2f outx_a = (v2f{lx.x, lx.y} - ox) * ix, outx_b = (v2f{lx.z, lx.w} - ox) * ix;
    const v2f outy_a = (v2f{ly.x, ly.y} - oy) * iy, outy_b = (v2f{ly.z, ly.w} - oy) * iy;
    const v2f outz_a = (v2f{lz.x, lz.y} - oz) * iz, outz_b = (v2f{lz.z, lz.w} - oz) * iz;
    h0 = slabResult(inx_a.x, iny_a.x, inz_a.x, outx_a.x, outy_a.x, outz_a.x, e0) && !(e0 > cull_t);
    h1 = slabResult(inx_a.y, iny_a.y, inz_a.y, outx_a.y, outy_a.y, outz_a.y, e1) && !(e1 > cull_t);
    h2 = slabResult(inx_b.x, iny_b.x, inz_b.x, outx_b.x, outy_b.x, outz_b.x, e2) && !(e2 > cull_t);
    h3 = slabResult(inx_b.y, iny_b.y, inz_b.y, outx_b.y, outy_b.y, outz_b.y, e3) && !(e3 > cull_t);
#else
    h0 = boxTest(lx.x, ly.x, lz.x, hx.x, hy.x, hz.x, o, inv, e0) && !(e0 > cull_t);
    h1 = boxTest(lx.y, ly.y, lz.y, hx.y, hy.y, hz.y, o, inv, e1) && !(e1 > cull_t);
    h2 = boxTest(lx.z, ly.z, lz.z, hx.z, hy.z, hz.z, o, inv, e2) && !(e2 > cull_t);
    h3 = boxTest(lx.w, ly.w, lz.w, hx.w, hy.w, hz.w, o, inv, e3) && !(e3 > cull_t);
#endif
    r0 = f2u(rf.x); r1 = f2u(rf.y); r2 = f2u(rf.z); r3 = f2u(rf.w);
    const int n = (int)h0 + (int)h1 + (int)h2 + (int)h3;
    if (n == 0) return false;
    // sort by entry distance; children that are not visited get a key no visited one can reach (a visited
    // entry is <= cull_t, a finite bound or +inf when nothing bounds the search: then nothing is skipped by distance and the
    // only keys of 3e38 belong to children whose box the ray misses) and so end up last
    const float skip = 3.0e38f;
    float k0 = h0 ? e0 : skip, k1 = h1 ? e1 : skip, k2 = h2 ? e2 : skip, k3 = h3 ? e3 : skip;
    TRT_CSWAP(k0, r0, k1, r1)
    TRT_CSWAP(k2, r2, k3, r3)
    TRT_CSWAP(k0, r0, k2, r2)
    TRT_CSWAP(k1, r1, k3, r3)
    TRT_CSWAP(k1, r1, k2, r2)
    if (n > 3) stk.push(sp++, r3);
    if (n > 2) stk.push(sp++, r2);
    if (n > 1) stk.push(sp++, r1);
    cur = r0;
    return true;
}

// ------------------------------------------------------ traverseBVH (a3) ----
// Closest hit with the reference's result semantics (bvh.cpp:146-175,
// 211-229) on an explicit stack: children are visited nearest-first and a
// subtree is skipped only when its box entry is STRICTLY beyond the best hit,
// so every equal-distance candidate is still seen.  Tie rules (Q10):
//   inside a leaf   — index order; replace when nearer, or equal and emissive
//                     (bvh.cpp:219);
//   between leaves  — the reference's sibling merge "r1 if r1 emissive else r2"
//                     (bvh.cpp:168-172), with r1 the lower-index subtree, is
//                     order independent as: leftmost emissive wins, otherwise the
//                     rightmost candidate wins.
// Stack: push(sp, ref) / pop(sp) with sp < scene depth.
// RULE: every triangle hit is tested against leafEntry() on the spot (the exact, slow form).  The kernels run RULE = false and
// validate the RESULT instead (traceClosest() below): a hit in front of its leaf's box can only matter if it is the one the
// traversal ends with — while it was the best hit it culled nothing a valid nearer hit lies in (such a hit's boxes all start
// before it), and once a valid nearer hit replaces it the result is the one the exact form finds.  One box test per ray instead
// of one per triangle hit; the exact form reruns the (one in ~10^7) rays whose result fails it.
template <class Stack, bool COUNT, int NK = 0, bool RULE = false>
// `t_init` bounds the search (only hits STRICTLY nearer count) and `any` stops at the first leaf that yields
// one: together they make the occlusion test of TRT_FLAG_FIXED_NEE (tri >= 0 <=> something lies in front of t_init).
// `redo`: t_init is only a hint — a hit in front of it is the closest hit of the whole scene (anything nearer than the
// bound beats everything beyond it, ties included), and when there is none the search runs again without the bound.
TRT_HD inline Hit traceClosestPass(const SceneDev& sc, f3 o, f3 d, Stack& stk, uint32_t& n_inner, uint32_t& n_tri, float t_init = TRT_INF, bool any = false,
                                   bool redo = false)
{
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    Hit best;
    best.t = t_init; best.tri = -1; best.u = 0.f; best.v = 0.f; best.flags = 0u;
    float best_det = 1.0f;  // best.u / best.v hold the numerators un, vn until the end
  for (;;) {  // once; twice when `redo` and nothing lies in front of t_init
    int sp = 0;
    uint32_t cur = 0;  // nodes[0] is always an inner node
    for (;;) {
        if (cur & TRT_LEAF_BIT) {
            const uint32_t first = TRT_LEAF_FIRST(cur), count = TRT_LEAF_COUNT(cur);
            float lt = TRT_INF, lun = 0.f, lvn = 0.f, ldet = 1.0f;
            int32_t li = -1;
            uint32_t lflags = 0u;
            if (count) {
#if TRT_PREFETCH
                TriIsect T = sc.tri_isect[first];
#endif
                for (uint32_t k = 0; k < count; ++k) {
                    const uint32_t i = first + k;
#if TRT_PREFETCH
                    // fetch the next record of the leaf while this one is tested
                    const TriIsect Tn = sc.tri_isect[k + 1 < count ? i + 1 : i];
#else
                    const TriIsect T = sc.tri_isect[i];
#endif
                    if (COUNT) n_tri++;
                    float t, un, vn, det;
                    if (triTest(T, o, d, t, un, vn, det) && !(RULE && t < leafFloor(sc, i, o, inv))) {
                        const uint32_t fl = f2u(T.c.z);
                        if ((t == lt && (fl & 1u)) || t < lt) { lt = t; li = (int32_t)i; lun = un; lvn = vn; ldet = det; lflags = fl; }
                    }
#if TRT_PREFETCH
                    T = Tn;
#endif
                }
            }
            if (li >= 0) {
                bool take = lt < best.t;
                if (lt == best.t && best.tri >= 0) {
                    const bool lem = (lflags & 1u) != 0, bem = (best.flags & 1u) != 0;
                    take = lem ? (!bem || li < best.tri) : (!bem && li > best.tri);
                }
                if (take) { best.t = lt; best.tri = li; best.u = lun; best.v = lvn; best_det = ldet; best.flags = lflags; }
            }
            if (sp == 0 || (any && best.tri >= 0)) break;
            cur = stk.pop(--sp);
            continue;
        }
        if (COUNT) n_inner++;
        if (!innerStep(sc, cur, sp, stk, o, inv, trt_cull_bound(best.t, sc.cull_alpha))) {
            if (sp == 0) break;
            cur = stk.pop(--sp);
        }
    }
    if (!redo || best.tri >= 0 || !(best.t < TRT_INF)) break;
    best.t = TRT_INF;
  }
    if (best.tri >= 0) {  // barycentric weights of v1, v2 (the values findBaryCor feeds bvh.cpp:224)
        best.u = best.u / best_det;
        best.v = best.v / best_det;
    }
    return best;
}

// traverseBVH on the caller's BVH2 with the literal slab test (boxTestGlm), unculled, for the rays of raySpecial(): every box the ray passes is entered (bvh.cpp:162-166),
// every triangle of a leaf so reached is tested and held to the leaf-box rule with the entry distance of THIS test (the oracle's leafScan), leaves are merged by the
// order-independent form of bvh.cpp:168-172.  `any`: the occlusion test of TRT_FLAG_FIXED_NEE (a hit nearer than t_init ends the search).  Stack: <= BVH2 depth entries.
template <class Stack, bool COUNT>
TRT_HD inline Hit traceClosestBvh2Glm(const SceneDev& sc, f3 o, f3 d, Stack& stk, uint32_t& n_inner, uint32_t& n_tri, float t_init, bool any)
{
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    Hit best;
    best.t = any ? t_init : TRT_INF; best.tri = -1; best.u = 0.f; best.v = 0.f; best.flags = 0u;
    float best_det = 1.0f;
    int sp = 0;
    uint32_t cur = 0;  // nodes[0] is always an inner node
    for (;;) {
        if (cur & TRT_LEAF_BIT) {
            const uint32_t first = TRT_LEAF_FIRST(cur), count = TRT_LEAF_COUNT(cur);
            float lt = TRT_INF, lun = 0.f, lvn = 0.f, ldet = 1.0f;
            int32_t li = -1;
            uint32_t lflags = 0u;
            if (count) {
                const f4 ba = sc.leaf_box[2 * (size_t)first], bb = sc.leaf_box[2 * (size_t)first + 1];  // the box this leaf was entered through
                float e;
                (void)boxTestGlm(ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, o, inv, e);
                const float floor_t = trt_leaf_floor(e, sc.leaf_alpha);
                for (uint32_t k = 0; k < count; ++k) {
                    const uint32_t i = first + k;
                    const TriIsect T = sc.tri_isect[i];
                    if (COUNT) n_tri++;
                    float t, un, vn, det;
                    if (triTest(T, o, d, t, un, vn, det) && !(t < floor_t)) {
                        const uint32_t fl = f2u(T.c.z);
                        if ((t == lt && (fl & 1u)) || t < lt) { lt = t; li = (int32_t)i; lun = un; lvn = vn; ldet = det; lflags = fl; }
                    }
                }
            }
            if (li >= 0) {
                bool take = lt < best.t;
                if (lt == best.t && best.tri >= 0) {
                    const bool lem = (lflags & 1u) != 0, bem = (best.flags & 1u) != 0;
                    take = lem ? (!bem || li < best.tri) : (!bem && li > best.tri);
                }
                if (take) { best.t = lt; best.tri = li; best.u = lun; best.v = lvn; best_det = ldet; best.flags = lflags; }
            }
            if (sp == 0 || (any && best.tri >= 0)) break;
            cur = stk.pop(--sp);
            continue;
        }
        if (COUNT) n_inner++;
        const trt_bvh_node& nd = sc.nodes[cur];
        float e0, e1;
        const bool h0 = boxTestGlm(nd.lo0[0], nd.lo0[1], nd.lo0[2], nd.hi0[0], nd.hi0[1], nd.hi0[2], o, inv, e0);
        const bool h1 = boxTestGlm(nd.lo1[0], nd.lo1[1], nd.lo1[2], nd.hi1[0], nd.hi1[1], nd.hi1[2], o, inv, e1);
        const uint32_t c0 = nd.child0, c1 = nd.child1;
        if (h0 && h1) { stk.push(sp++, c1); cur = c0; }
        else if (h0) cur = c0;
        else if (h1) cur = c1;
        else {
            if (sp == 0) break;
            cur = stk.pop(--sp);
        }
    }
    if (best.tri >= 0) {
        best.u = best.u / best_det;
        best.v = best.v / best_det;
    }
    return best;
}

// does the hit (t, tri) lie in front of the box of tri's leaf by more than the tolerance?  (leafFloor(): such a hit does not count)
TRT_HD inline bool hitInFrontOfItsLeaf(const SceneDev& sc, float t, int32_t tri, f3 o, f3 inv)
{
    return tri >= 0 && t < leafFloor(sc, (uint32_t)tri, o, inv);
}
template <class Stack, bool COUNT, int NK = 0>
TRT_HD inline Hit traceClosest(const SceneDev& sc, f3 o, f3 d, Stack& stk, uint32_t& n_inner, uint32_t& n_tri, float t_init = TRT_INF, bool any = false,
                               bool redo = false)
{
    {
        const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        if (raySpecial(inv) && rayOnABoxPlane(sc, o, inv)) return traceClosestBvh2Glm<Stack, COUNT>(sc, o, d, stk, n_inner, n_tri, t_init, any);
    }
    const Hit h = traceClosestPass<Stack, COUNT, NK, false>(sc, o, d, stk, n_inner, n_tri, t_init, any, redo);
    if (!hitInFrontOfItsLeaf(sc, h.t, h.tri, o, mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z))) return h;
    return traceClosestPass<Stack, COUNT, NK, true>(sc, o, d, stk, n_inner, n_tri, t_init, any, redo);
}

// --------------------------------------------------------------- RNG stream ----
struct Stream {
    trt_rng_key key;
    uint32_t ctr;
    TRT_HD float next() { return trt_rng_uniform(key, ctr++); }
};

// ------------------------------------- primary ray (main.cpp:88-95, a1/a2) ----
// `fixed` (TRT_FLAG_FIXED_PIXELS): pixel (i, j) covers [j/W, (j+1)/W) x [(H-1-i)/H, (H-i)/H) of the image plane and the
// jitter is uniform inside it, instead of Q1 (rows shifted by one, pitch 1/(H-1)) and Q2 (jitter of 1/W on a 1/(W-1) grid).
// `grid` (kernels only): the correctly rounded reciprocals 1 / (W - 1), 1 / (H - 1), 1 / W, 1 / H, formed once per render on the
// host; with them the four binary64 divisions of the pixel grid are three instructions each (trt_div_by: the same bits, checked
// for every operand pair the grid can produce).  Null: plain divisions (images beyond 65536 pixels a side, W or H of 1, hosts).
TRT_HD inline void cameraRay(const trt_camera& cam, int W, int H, int i, int j, float u1, float u2, f3& o, f3& d, bool fixed = false, const double* grid = nullptr)
{
    double x, y;
    if (fixed) {
        x = (double(j) + (double)u1) / double(W);
        y = (double(H - 1 - i) + (double)u2) / double(H);
    } else if (grid) {
        x = trt_div_by(double(j), double(W - 1.0), grid[0]);
        y = trt_div_by(double(H - i), double(H - 1.0), grid[1]);  // Q1
        x += trt_div_by((double)u1 - 0.5, double(W), grid[2]);      // Q2
        y += trt_div_by((double)u2 - 0.5, double(H), grid[3]);
    } else {
        x = double(j) / double(W - 1.0);
        y = double(H - i) / double(H - 1.0);  // Q1
        x += ((double)u1 - 0.5) / double(W);  // Q2
        y += ((double)u2 - 0.5) / double(H);
    }
    const float s = (float)x, t = (float)y;
    const f3 llc = ld3(cam.lower_left_corner), hor = ld3(cam.horizontal), ver = ld3(cam.vertical), eye = ld3(cam.eye);
    o = eye;
    d = normalize(((llc + hor * s) + ver * t) - eye);  // camera.cpp:19-28
}

// ----------------------------------------------------------- Sample (a12) ----
// pathTracing.cpp:111-145 (phi drawn first).  asin(sqrt(u)) and acos(u^(1/(Ns+1)))
// enter only through their sine and cosine.
TRT_HD inline f3 sampleDir(f3 a, int ray_type, float Ns, float u_phi, float u_theta)
{
    float c, s;
    trt_sincos2pi(u_phi, &c, &s);
    float sin_t, cos_t;
    if (ray_type == TRT_RAY_DIFFUSE) {
        sin_t = trt_sqrt(u_theta);
        cos_t = trt_sqrt(1.0f - u_theta);
    } else {
        cos_t = trt_pow01(u_theta, 1.0f / (Ns + 1.0f));
        const float s2 = 1.0f - cos_t * cos_t;
        sin_t = trt_sqrt(s2 > 0.0f ? s2 : 0.0f);
    }
    const f3 local = mk3(sin_t * c, cos_t, sin_t * s);
    f3 front;
    if (fabsf(a.x) > fabsf(a.y)) front = normalize(mk3(a.z, 0.0f, -a.x));
    else front = normalize(mk3(0.0f, -a.z, a.y));
    const f3 right = cross(a, front);
    return normalize((right * local.x + a * local.y) + front * local.z);
}

// ---------------------------------------------------------- nextRay (a11) ----
// pathTracing.cpp:147-209; I = incoming direction.  In two halves, so that a kernel can put a queue reservation between
// them: nextRayDecide makes every random draw and every branch decision that tells WHETHER a ray leaves the vertex (cheap);
// nextRayFinish builds its direction (refract / reflect / Sample: the expensive part) and its type.  nextRay = both.
struct NextPlan {
    int kind;  // 0: INVALID (no ray), 1: Fresnel transmission branch, 2: diffuse lobe, 3: specular lobe
    float u_phi, u_theta;
};
TRT_HD inline NextPlan nextRayDecide(const MaterialDev& m, f3 pn, f3 I, Stream& rng)
{
    NextPlan pl;
    pl.kind = 0; pl.u_phi = 0.0f; pl.u_theta = 0.0f;
    if (m.Ni > 1.0f) {
        const float cos_in = dot(I, pn);
        const float rf0 = m.rf0;  // ((n1 - n2) / (n1 + n2))^2, makeMaterialDev
        const float x = 1.0f - fabsf(cos_in);
        const float x2 = x * x;
        const float fresnel = rf0 + (1.0f - rf0) * ((x2 * x2) * x);
        if (fresnel < rng.next()) { pl.kind = 1; return pl; }
    }
    const float kd = m.sel_kd, kdks = m.sel_kdks;  // |Kd| / (|Kd| + |Ks|) and that + |Ks| / (|Kd| + |Ks|), makeMaterialDev
    // the lobe draw p lies in [0, 1): a purely diffuse material (kd == 1) takes the diffuse lobe whatever p is, so the
    // draw is only counted there, not generated
    float p = 0.0f;
    if (kd >= 1.0f) rng.ctr++;
    else p = rng.next();
    if (p < kd) pl.kind = 2;
    else if (m.Ns > 1.0f && p < kdks) pl.kind = 3;
    else return pl;
    pl.u_phi = rng.next();
    pl.u_theta = rng.next();
    return pl;
}
TRT_HD inline int nextRayFinish(const MaterialDev& m, f3 pn, f3 I, const NextPlan& pl, f3& out)
{
    if (pl.kind == 1) {
        const float cos_in = dot(I, pn);
        f3 n;
        float eta;  // n1 / n2
        if (cos_in > 0.0f) { n = -pn; eta = m.Ni; }         // n1 = Ni, n2 = 1:  Ni / 1 == Ni
        else { n = pn; eta = m.Ni_inv; }                    // n1 = 1, n2 = Ni
        const f3 T = refract(I, n, eta);
        if (T.x != 0.0f || T.y != 0.0f || T.z != 0.0f) { out = T; return TRT_RAY_TRANSMISSION; }
        out = reflect(I, n);
        return TRT_RAY_SPECULAR;
    }
    if (pl.kind == 2) {
        out = sampleDir(pn, TRT_RAY_DIFFUSE, m.Ns, pl.u_phi, pl.u_theta);
        return TRT_RAY_DIFFUSE;
    }
    if (pl.kind == 3) {
        out = sampleDir(reflect(I, pn), TRT_RAY_SPECULAR, m.Ns, pl.u_phi, pl.u_theta);
        return TRT_RAY_SPECULAR;
    }
    out = mk3(0.f, 0.f, 0.f);
    return TRT_RAY_INVALID;
}
TRT_HD inline int nextRay(const MaterialDev& m, f3 pn, f3 I, Stream& rng, f3& out)
{
    const NextPlan pl = nextRayDecide(m, pn, I, rng);
    return nextRayFinish(m, pn, I, pl, out);
}

// ------------------------------------------------------ path vertex (a5/a8) ----
struct Vertex {
    f3 P, pn, wi, Kd;
    int32_t mat;
};

// Hit point (bvh.cpp:191), shading normal from the hit's barycentrics
// (bvh.cpp:223-224; closed form instead of the QR solve of triangle.cpp:12-29),
// albedo or nearest texel (pathTracing.cpp:15-30).
TRT_HD inline Vertex makeVertex(const SceneDev& sc, const Hit& h, f3 o, f3 d, const TriShade& ts, const MaterialDev& m)
{
    Vertex vx;
    vx.P = mk3(o.x + d.x * h.t, o.y + d.y * h.t, o.z + d.z * h.t);
    vx.wi = -d;
    vx.mat = ts.mat;
    const float b0 = (1.0f - h.u) - h.v, b1 = h.u, b2 = h.v;
    vx.pn = normalize((ld3(ts.vn) * b0 + ld3(ts.vn + 3) * b1) + ld3(ts.vn + 6) * b2);
    if (m.tex >= 0) {
        const TextureDev tx = sc.textures[m.tex];
        const float colf = (ts.vt[0] * b0 + ts.vt[2] * b1) + ts.vt[4] * b2;
        const float rowf = (ts.vt[1] * b0 + ts.vt[3] * b1) + ts.vt[5] * b2;
        const double col = colf, row = rowf;
        const double irow = row - floor(row), icol = col - floor(col);
        int r = (int)(irow * tx.height), c = (int)(icol * tx.width);
        if (r > tx.height - 1) r = tx.height - 1;
        if (c > tx.width - 1) c = tx.width - 1;
        if (r < 0) r = 0;
        if (c < 0) c = 0;
        const uint8_t* px = sc.tex_bytes + tx.offset + ((size_t)r * tx.width + c) * 3;
        vx.Kd = mk3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
    } else {
        vx.Kd = ld3(m.Kd);
    }
    return vx;
}

// One light of shade()'s direct-illumination loop up to, but excluding, the
// shadow ray (pathTracing.cpp:34-53).  Returns true when a shadow ray must be
// traced; `wo` is its direction, `contrib` what it adds to L_dir if the closest
// hit carries the light's material (pathTracing.cpp:55-70).
// `fixed` (TRT_FLAG_FIXED_NEE): the CDF draw spans this light's own area (not Q3), the point is uniform on the
// triangle (not Q4), and `t_max` = 0.999 |x' - x| bounds the occlusion test that replaces the closest-hit +
// material comparison (not Q5).  In parity mode t_max = 1.001 |x' - x| is only a search hint for the closest-hit
// query (traceClosest's `redo`): the light's own triangle lies inside it, so the bounded search almost always
// finds the closest hit at once, having culled everything beyond the light.
TRT_HD inline bool lightSample(const SceneDev& sc, const Vertex& vx, const MaterialDev& m, uint32_t li, Stream& rng, f3& wo, f3& contrib, bool fixed, float& t_max)
{
    const LightDev L = sc.lights[li];
    const float rnd = rng.next() * (fixed ? L.area : sc.light0_area);  // Q3
    // first triangle whose cumulative area exceeds rnd (the linear scan of pathTracing.cpp:38-42); on a
    // non-decreasing CDF a bisection finds the same index
    uint32_t pick = L.tri_count;
    if (sc.light_cum) {
        uint32_t lo = 0, hi = L.tri_count;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (rnd < sc.light_cum[L.tri_first + mid]) hi = mid; else lo = mid + 1;
        }
        pick = lo;
    } else {
        for (uint32_t k = 0; k < L.tri_count; ++k)
            if (rnd < sc.light_tris[L.tri_first + k].cum_area) { pick = k; break; }
    }
    if (pick >= L.tri_count) return false;
    const LightTriDev T = sc.light_tris[L.tri_first + pick];  // whole record: five 16-B loads
    const LightTriDev* lt = &T;
    const float r1 = rng.next(), r2 = rng.next(), r3 = rng.next();
    float p1, p2, p3;
    if (fixed) {  // uniform on the triangle; r3 is drawn (same stream layout) but unused
        const float su = trt_sqrt(r1);
        p1 = 1.0f - su; p2 = su * (1.0f - r2); p3 = su * r2;
    } else {
        const float rs = (r1 + r2) + r3;
        p1 = r1 / rs; p2 = r2 / rs; p3 = r3 / rs;  // Q4
    }
    const f3 light_p = (ld3(lt->v[0]) * p1 + ld3(lt->v[1]) * p2) + ld3(lt->v[2]) * p3;
    const f3 light_n = normalize((ld3(lt->vn[0]) * p1 + ld3(lt->vn[1]) * p2) + ld3(lt->vn[2]) * p3);
    const f3 diff = light_p - vx.P;
    float diff_len, diff_rlen;  // length(diff) and the factor of normalize(diff): one square root serves both
    trt_sqrt_rsqrt2(dot(diff, diff), &diff_len, &diff_rlen);
    wo = diff * diff_rlen;
    const float cos_s = dot(wo, vx.pn);
    if (!(cos_s > 0.0f)) return false;  // pathTracing.cpp:60: such a sample never contributes
    t_max = fminf((fixed ? 0.999f : 1.001f) * diff_len, TRT_INF);  // Q7: nothing beyond 114514 is ever a hit (bvh.h:5, bvh.cpp:219), hint and occlusion range included
    const float pdf_light = L.pdf;  // 1 / area, makeLightDev
    const float cos_theta_p = fabsf(dot(wo, light_n));
    const float cos_theta = fabsf(cos_s / length(vx.pn));
    const f3 radiance = ld3(L.radiance);
    const f3 intensity = (((radiance * cos_theta_p) * cos_theta) / dot(diff, diff)) / pdf_light;
    const f3 kd_pi = m.tex >= 0 ? vx.Kd / TRT_PI : ld3(m.Kd_pi);
    f3 brdf;
    if (m.no_spec) {
        // Ks = 0 and 0 < Ns < inf: cos_alpha is in [0, 1] (never NaN), pow01 there is finite, so the Phong term
        // ((Ks (Ns+2)) pw) / 2pi is exactly +-0 and kd_pi + it is kd_pi (makeMaterialDev rules out the one case
        // -0 + +0 where it would not be): the half vector, the power and three divisions are not evaluated at all.
        brdf = kd_pi;
    } else {
        const f3 hv = normalize((vx.wi + wo) * 0.5f);
        const float ca = dot(vx.pn, hv);
        const float cos_alpha = ca > 0.0f ? ca : 0.0f;
        const float pw = trt_pow01(cos_alpha, m.Ns);
        const f3 spec = ((ld3(m.Ks) * (m.Ns + 2.0f)) * pw) / (2.0f * TRT_PI);
        brdf = kd_pi + spec;
    }
    contrib = intensity * brdf;
    return true;
}

// Path-state word carried with every queued ray.
//   bits 0..15  next RNG draw index        bits 16..17 type of the ray being traced
//   (3 = camera ray)                       bits 20..31 path depth (vertex index)
TRT_HD inline uint32_t packMeta(uint32_t ctr, uint32_t type, uint32_t depth) { return (ctr & 0xFFFFu) | ((type & 3u) << 16) | (depth << 20); }
TRT_HD inline uint32_t metaCtr(uint32_t m) { return m & 0xFFFFu; }
TRT_HD inline uint32_t metaType(uint32_t m) { return (m >> 16) & 3u; }
TRT_HD inline uint32_t metaDepth(uint32_t m) { return m >> 20; }
#define TRT_META_CAMERA 3u
#define TRT_MAX_PATH_DEPTH 4000u

// ------------------------------------------------ one path vertex: shade() ----
// Which image rows/pixels a render call covers, and the sample numbering.
struct TileDesc {
    const int32_t* rows;  // image row of each packed output row
    int32_t tile_w, x0, width, height;
    uint32_t fixed_nee;   // TRT_FLAG_FIXED_NEE
    uint32_t fixed_pixels;  // TRT_FLAG_FIXED_PIXELS
    uint32_t ray_offset;    // TRT_FLAG_RAY_OFFSET
    uint32_t specular_ks;   // TRT_FLAG_SPECULAR_KS
    uint32_t npix;        // rows * tile_w
    uint32_t seed, spp;
    uint32_t npix_magic, tile_w_magic;  // magicOf(npix), magicOf(tile_w)
    uint32_t grid_ok;     // grid_rcp may be used: 2 <= width, height <= 65536 (the operand range trt_div_by is checked for)
    double grid_rcp[4];   // 1 / (W - 1), 1 / (H - 1), 1 / W, 1 / H, correctly rounded (host)
};

// n / d for a divisor whose magic number m = min(floor(2^32 / d), 2^32 - 1) was formed on the host (TileDesc): the high word of
// n m is floor(n / d) or one less for every n < 2^32 (n m / 2^32 > n / d - n / 2^32 > n / d - 1), so one correction gives the
// quotient: two multiplications instead of the twenty-instruction general division (two of them per path vertex: the RNG key).
TRT_HD inline uint32_t divMagic(uint32_t n, uint32_t d, uint32_t m)
{
    uint32_t q = (uint32_t)(((uint64_t)n * m) >> 32);
    if (n - q * d >= d) q++;
    return q;
}
TRT_HD inline uint32_t magicOf(uint32_t d) { return d <= 1u ? 0xFFFFFFFFu : (uint32_t)(0x100000000ull / d); }

// Image row of packed row r of the tile: the table in global memory.  k_shade passes a look-up of its own (an LDS copy of
// the table: a dependent global load per vertex otherwise).
struct RowsGlobal {
    TRT_HD int operator()(const TileDesc& td, uint32_t r) const { return td.rows[r]; }
};

// path id -> (pixel index y*W+x, sample index): id = s_local * npix + pixel-in-tile
template <class Rows = RowsGlobal>
TRT_HD inline trt_rng_key pathKey(const TileDesc& td, uint32_t s0, uint32_t pid, Rows rows = Rows())
{
    const uint32_t s_local = divMagic(pid, td.npix, td.npix_magic), pl = pid - s_local * td.npix;
    const uint32_t r = divMagic(pl, (uint32_t)td.tile_w, td.tile_w_magic), c = pl - r * (uint32_t)td.tile_w;
    const uint32_t pixel = (uint32_t)rows(td, r) * (uint32_t)td.width + (uint32_t)(td.x0 + (int)c);
    return trt_rng_make_key(td.seed, pixel, s0 + s_local);
}

// The camera ray of path `pid` (main.cpp:88-95, camera.cpp:19-28) as the queue record (ra, rb): a pure
// function of the path id, so bounce 0 never goes through HBM — the traversal kernel generates it and
// k_shade generates the same bits again.
template <class Rows = RowsGlobal>
TRT_HD inline void primaryRay(const SceneDev& sc, const TileDesc& td, uint32_t s0, uint32_t pid, f4& ra, f4& rb, Rows rows = Rows())
{
    const uint32_t s_local = divMagic(pid, td.npix, td.npix_magic), pl = pid - s_local * td.npix;
    const uint32_t r = divMagic(pl, (uint32_t)td.tile_w, td.tile_w_magic), c = pl - r * (uint32_t)td.tile_w;
    const int y = rows(td, r), x = td.x0 + (int)c;
    Stream rng;
    rng.key = trt_rng_make_key(td.seed, (uint32_t)y * (uint32_t)td.width + (uint32_t)x, s0 + s_local);
    rng.ctr = 0;
    const float u1 = rng.next(), u2 = rng.next();  // jitter x, then y (main.cpp:92-93)
    f3 o, d;
    cameraRay(sc.cam, td.width, td.height, y, x, u1, u2, o, d, td.fixed_pixels != 0u, td.grid_ok ? td.grid_rcp : nullptr);
    ra = mk4(o.x, o.y, o.z, d.x);
    rb = mk4(d.y, d.z, u2f(pid), u2f(packMeta(rng.ctr, TRT_META_CAMERA, 0)));
}

struct ShadeCtx {
    bool had_hit;   // the traced ray hit something
    bool shade_ok;  // ... a non-emissive surface: NEE + continuation follow
    bool add_L;     // ... an emissive surface whose radiance is kept: Lacc[pid] += addL
    f3 addL;
    f3 d, beta;
    uint32_t pid, depth;
    Vertex vx;
    const MaterialDev* m;  // into the scene's table (LDS copy inside k_shade): fields are fetched where they are used, not held in registers
    Stream rng;
    bool spec_ks;   // TRT_FLAG_SPECULAR_KS: a SPECULAR bounce is weighted by the material's Ks instead of the texel Kd
    bool use_off;   // TRT_FLAG_RAY_OFFSET
    f3 off;         // eps * Ng of the hit triangle (only with use_off)
};

// eps * geometric normal of triangle `tri` at hit point P (TRT_FLAG_RAY_OFFSET; the same expression in the oracle)
TRT_HD inline f3 offsetVector(const TriIsect& T, f3 P)
{
    const f3 e1 = mk3(T.a.w, T.b.x, T.b.y), e2 = mk3(T.b.z, T.b.w, T.c.x);
    const f3 ng = normalize(cross(e1, e2));
    const float eps = TRT_OFFSET_EPS * fmaxf(1.0f, fmaxf(fabsf(P.x), fmaxf(fabsf(P.y), fabsf(P.z))));
    return ng * eps;
}
// where a ray that leaves the vertex in direction w starts: the hit point (Q6), or, with TRT_FLAG_RAY_OFFSET, eps off the surface
// on w's side
TRT_HD inline f3 rayOrigin(const ShadeCtx& c, f3 w)
{
    if (!c.use_off) return c.vx.P;
    return dot(c.off, w) >= 0.0f ? c.vx.P + c.off : c.vx.P - c.off;
}

// First part of shade() (pathTracing.cpp:9-30) for the ray (ra, rb, bt) and its hit record.
template <class Rows = RowsGlobal>
TRT_HD inline void shadeBegin(const SceneDev& sc, const TileDesc& td, uint32_t s0, const f4& ra, const f4& rb, const f4& bt, const f4& hit4, ShadeCtx& c, Rows rows = Rows())
{
    c.had_hit = c.shade_ok = c.add_L = false;
    c.addL = mk3(0, 0, 0);
    c.d = mk3(0, 0, 0);
    c.beta = mk3(0, 0, 0);
    c.pid = 0; c.depth = 0;
    c.vx.P = c.vx.pn = c.vx.wi = c.vx.Kd = mk3(0, 0, 0);
    c.vx.mat = 0;
    c.m = sc.materials;  // any valid record: only dereferenced under had_hit
    c.rng.key.k0 = c.rng.key.k1 = 0; c.rng.ctr = 0;
    c.use_off = td.ray_offset != 0u;
    c.spec_ks = td.specular_ks != 0u;
    c.off = mk3(0, 0, 0);
    Hit h;
    h.t = hit4.x; h.tri = (int32_t)f2u(hit4.y); h.u = hit4.z; h.v = hit4.w; h.flags = 0;
    if (h.tri < 0) return;
    c.had_hit = true;
    const f3 o = mk3(ra.x, ra.y, ra.z);
    c.d = mk3(ra.w, rb.x, rb.y);
    c.beta = mk3(bt.x, bt.y, bt.z);
    c.pid = f2u(rb.z);
    const uint32_t meta = f2u(rb.w);
    c.depth = metaDepth(meta);
    const TriShade ts = sc.tri_shade[h.tri];
    c.m = sc.materials + ts.mat;
    if (c.m->is_emissive) {
        // pathTracing.cpp:9-12 returns the radiance; the callers keep it for the camera ray
        // (main.cpp:101) and for TRANSMISSION (pathTracing.cpp:95-96), not after DIFFUSE/SPECULAR (Q9)
        const uint32_t type = metaType(meta);
        if (c.depth == 0 || type == TRT_RAY_TRANSMISSION) {
            const f3 rad = ld3(c.m->radiance);
            c.add_L = true;
            c.addL = c.depth == 0 ? rad : c.beta * rad;
        }
        return;
    }
    c.shade_ok = true;
    c.vx = makeVertex(sc, h, o, c.d, ts, *c.m);
    if (c.use_off) c.off = offsetVector(sc.tri_isect[h.tri], c.vx.P);
    c.rng.key = pathKey(td, s0, c.pid, rows);
    c.rng.ctr = metaCtr(meta);
}

// Last part of shade() (pathTracing.cpp:78-99): RR(0.8), nextRay, beta update — in the two halves of nextRay.
// shadeNextDecide returns true when an extension ray leaves the vertex (INVALID rays are not traced); shadeNextFinish
// then builds its queue record.
TRT_HD inline bool shadeNextDecide(ShadeCtx& c, int max_depth, NextPlan& pl)
{
    pl.kind = 0; pl.u_phi = 0.0f; pl.u_theta = 0.0f;
    if (!c.shade_ok) return false;
    const bool last = (max_depth > 0 && (int)c.depth + 1 >= max_depth) || c.depth + 1 >= TRT_MAX_PATH_DEPTH;
    if (last || !(c.rng.next() < TRT_P_RR)) return false;  // RR, pathTracing.cpp:104-109
    pl = nextRayDecide(*c.m, c.vx.pn, c.d, c.rng);
    return pl.kind != 0;
}
TRT_HD inline void shadeNextFinish(const ShadeCtx& c, const NextPlan& pl, f4& ra, f4& rb, f4& bt)
{
    f3 nd;
    const int type = nextRayFinish(*c.m, c.vx.pn, c.d, pl, nd);
    const f3 w = (type == TRT_RAY_TRANSMISSION) ? ld3(c.m->Tr) : ((c.spec_ks && type == TRT_RAY_SPECULAR) ? ld3(c.m->Ks) : c.vx.Kd);  // Q8: SPECULAR is weighted by Kd too (by Ks with TRT_FLAG_SPECULAR_KS)
    const f3 nb = (c.beta * w) / TRT_P_RR;
    const f3 org = rayOrigin(c, nd);  // Q6: the hit point itself unless TRT_FLAG_RAY_OFFSET
    ra = mk4(org.x, org.y, org.z, nd.x);
    rb = mk4(nd.y, nd.z, u2f(c.pid), u2f(packMeta(c.rng.ctr, (uint32_t)type, c.depth + 1)));
    bt = mk4(nb.x, nb.y, nb.z, 0.0f);
}
TRT_HD inline bool shadeNext(ShadeCtx& c, int max_depth, f4& ra, f4& rb, f4& bt)
{
    NextPlan pl;
    if (!shadeNextDecide(c, max_depth, pl)) return false;
    shadeNextFinish(c, pl, ra, rb, bt);
    return true;
}

}  // namespace trtd
