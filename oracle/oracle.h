/*
 * oracle.h — C interface of the CPU oracle (oracle/liboracle.so).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under tinyraytracing_amd/ may include,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.  See oracle.cpp for what it restates and how it is
 * (not) pinned.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include "trt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_stats {
    uint64_t rays_camera, rays_shadow, rays_indirect, shaded_hits;
    uint64_t inner_visits[2]; /* [closest, shadow] */
    uint64_t tri_tests[2];
    uint32_t max_bounces;
    int32_t threads;
    double seconds;           /* wall time of the render loop only */
} oracle_stats;

#define ORACLE_MODE_ITERATIVE 0 /* beta-weighted loop: the bit-parity target of the HIP path */
#define ORACLE_MODE_RECURSIVE 1 /* literal recursion of shade() (pathTracing.cpp:3-102) */
#define ORACLE_MODE_EXPERIMENT_NO_RR_DIV 0x100 /* or-ed into `mode` (iterative form): the estimator WITHOUT the 1 / P_RR of pathTracing.cpp:84 — an
                                                * experiment of tests/test_ref_png.py against the reference's older `back` snapshots, never the parity path.
                                                * An explicit argument: no environment variable can change what this library computes. */
#define ORACLE_MODE_EXPERIMENT_SPECULAR_KS 0x200 /* or-ed into `mode` (iterative form): a SPECULAR bounce weighted by the material's Ks instead of the texel Kd that
                                                  * pathTracing.cpp:91-93 multiplies by (Q8) — an experiment against the reference's staircase snapshots, whose teal
                                                  * Metal strip (Kd 0.2, Ks 0 0.8 0.8) the committed weighting cannot produce; never the parity path. */

/* The three glass hypotheses VERDICT r03 listed for the staircase residual, each an explicit bit (iterative form; tests/test_ref_png.py measures what each explains —
 * nothing: the residual is the weight of SPECULAR bounces, above): */
#define ORACLE_MODE_EXPERIMENT_GLASS_MIRROR 0x400       /* Ni > 1 and the Fresnel draw says "reflect": a mirror reflection instead of the fall-through to the opaque lobes (pathTracing.cpp:173-194) */
#define ORACLE_MODE_EXPERIMENT_NO_TR_ON_EMITTER 0x800   /* an emitter reached through a TRANSMISSION bounce is not weighted by that bounce's Tr (pathTracing.cpp:95-96 weights it) */
#define ORACLE_MODE_EXPERIMENT_NO_NEE_ON_GLASS 0x1000   /* no next-event estimation at vertices on Ni > 1 surfaces (pathTracing.cpp:34-74 samples the lights there too) */

/* main.cpp:80-113 restated.  Same trt_params semantics as trt_render (tile,
 * row interleave, packed float output).  threads <= 0 -> all cores. */
int oracle_render(const trt_scene* scene, const trt_params* p, float* out_rgb, oracle_stats* stats,
                  int threads, int mode);

/* ORACLE_MODE_LITERAL: the reference's own arithmetic (oracle_literal.cpp): plane + edge-cross triangle test with the
 * stored unit normal (bvh.cpp:177-209), double least-squares barycentrics (triangle.cpp:12-29), double scalars and libm
 * where the reference has them, shade() as a recursion; same counter RNG and draw order as the modes above.  Parity
 * mode only (max_depth 0, no TRT_FLAG_FIXED_*).  Used to measure and freeze the stated tolerance between the
 * reference's formulation and the one the HIP kernels share with oracle_render(). */
int oracle_render_literal(const trt_scene* scene, const trt_params* p, float* out_rgb, oracle_stats* stats, int threads);
/* Experiments on what the parity path does NOT take from the reference — its unsynchronised accumulation (main.cpp:79-81,103-108) and
 * its five shared random engines, three of them seeded alike (main.cpp:57, pathTracing.cpp:33,106,113,149) — to measure how much of the
 * brightness gap to the reference's saved renders they account for (tests/test_ref_png.py).  Whole image only. */
#define ORACLE_EXP_RACY_ACCUM 1
#define ORACLE_EXP_SHARED_ENGINES 2
#define ORACLE_EXP_INDEPENDENT_ENGINES 3
int oracle_render_literal_experiment(const trt_scene* scene, const trt_params* p, float* out_rgb, int threads, int experiment);
/* interactTriangle + findBaryCor literally on one triangle: returns 1 on hit, out = {t, b0, b1, b2}. */
int oracle_tri_test_literal(const float v[9], const float o[3], const float d[3], float out[4]);
/* traverseBVH with the literal triangle test on a ray batch (uv = barycentric weights of v1, v2 from findBaryCor). */
int oracle_trace_literal(const trt_scene* scene, uint64_t n, const float* org, const float* dir,
                         float* t, int32_t* tri, float* uv);

#define ORACLE_TRACE_REFERENCE 0 /* recursive, both children, no culling (bvh.cpp:146-175) */
#define ORACLE_TRACE_BRUTE 1     /* every triangle in index order through the leaf rule */
int oracle_trace(const trt_scene* scene, uint64_t n, const float* org, const float* dir,
                 float* t, int32_t* tri, float* uv, int mode, oracle_stats* stats);

/* buildBVH (bvh.cpp:16-144) restated, including its Cost = INF start value.
 * perm[i] = input index of the triangle that ends up at position i;
 * nodes must hold 2*n entries. */
int oracle_build_bvh(uint32_t n, const float* tri_v, int leaf_num, uint32_t* perm,
                     trt_bvh_node* nodes, uint32_t* n_nodes, uint32_t* depth);

/* ---- known-answer entry points for unit tests -------------------------------- */
/* interactTriangle (bvh.cpp:177-209): returns 1 on hit, out = {t, u, v}. */
int oracle_tri_test(const float v[9], const float o[3], const float d[3], float out[3]);
/* interactAABB (bvh.cpp:231-245). */
float oracle_aabb(const float lo[3], const float hi[3], const float o[3], const float d[3]);
/* Sample (pathTracing.cpp:111-145) with explicit uniforms (phi draw first). */
void oracle_sample(const float axis[3], int ray_type, float Ns, float u_phi, float u_theta, float out[3]);
void oracle_reflect(const float I[3], const float N[3], float out[3]);
void oracle_refract(const float I[3], const float N[3], float eta, float out[3]);
/* main.cpp:88-95 + camera.cpp:19-28 for pixel row i, column j with jitter draws u1,u2. */
void oracle_camera_ray(const trt_camera* cam, int width, int height, int i, int j, float u1, float u2,
                       float org[3], float dir[3]);
/* the same with TRT_FLAG_FIXED_PIXELS' mapping when fixed != 0 */
void oracle_camera_ray_mode(const trt_camera* cam, int width, int height, int i, int j, float u1, float u2, int fixed,
                            float org[3], float dir[3]);
/* nextRay (pathTracing.cpp:147-209) on explicit inputs; draws come from (seed,pixel,sample) at
 * counter *ctr (advanced).  Returns the ray type; out_dir is the new direction. */
int oracle_next_ray(const trt_material* m, const float pn[3], const float incoming[3], uint32_t seed,
                    uint32_t pixel, uint32_t sample, uint32_t* ctr, float out_dir[3]);
/* the shared primitives, exported so tests can check them against libm */
void oracle_prims_sincos2pi(float u, float* c, float* s);
float oracle_prims_pow01(float x, float y);
float oracle_prims_uniform(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t i);

/* One path, vertex by vertex, for debugging parity failures: for each path
 * vertex writes 8 floats {t, tri, u, v, L.r, L.g, L.b, ray_type_out}. Returns the
 * number of vertices written (<= max_vertices). */
int oracle_debug_path(const trt_scene* scene, const trt_params* p, int x, int y, int sample,
                      float* out, int max_vertices);

#ifdef __cplusplus
}
#endif
#endif
