/*
 * trt_prims.h — build-defined numeric primitives with NO analogue in the
 * reference, shared verbatim by the HIP kernels, the host code and the CPU
 * oracle so that all of them consume bit-identical random numbers and
 * transcendental values.
 *
 * Why they exist: the reference draws from five shared
 * `static std::default_random_engine`s (main.cpp:57-58, pathTracing.cpp:33,
 * 106,113,149) — there is no per-pixel stream to reproduce — and evaluates
 * sin/cos/asin/acos/pow through the platform libm in double.  A GPU path needs
 * (a) a counter-based stream keyed by (seed, pixel, sample) and (b) fp32
 * transcendental functions whose results do not depend on which libm is
 * linked.  Everything here is a fixed sequence of IEEE-754 binary32
 * add/mul/fma/div/sqrt operations and integer operations, so gcc on x86-64
 * (-ffp-contract=off, hardware FMA) and hipcc on gfx950 (-ffp-contract=off,
 * correctly-rounded divide/sqrt, the HIP default) produce identical bits.
 *
 * Nothing in this file restates reference arithmetic; everything that does is
 * written separately in oracle/ (CPU) and tinyraytracing_amd/csrc (HIP).
 */
#ifndef TRT_PRIMS_H
#define TRT_PRIMS_H

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define TRT_HD __host__ __device__
#else
#define TRT_HD
#endif

/* The ONE place where the sources shared by the HIP kernels and their CPU build (tests/hostsim, the oracle) differ by compiler: loop-unroll hints and
 * three integer bit operations whose device spelling is a HIP intrinsic.  Same results on both sides by definition of the operations (x != 0 where noted). */
#if defined(__HIPCC__)
#define TRT_UNROLL _Pragma("unroll")
#else
#define TRT_UNROLL
#endif
#if defined(__HIP_DEVICE_COMPILE__)
static inline __device__ uint32_t trt_clz32(uint32_t x) { return (uint32_t)__clz((int)x); }       /* x != 0 */
static inline __device__ uint32_t trt_ctz32(uint32_t x) { return (uint32_t)__ffs((int)x) - 1u; }  /* x != 0 */
static inline __device__ uint32_t trt_popc32(uint32_t x) { return (uint32_t)__popc(x); }
#else
static inline uint32_t trt_clz32(uint32_t x) { return (uint32_t)__builtin_clz(x); }
static inline uint32_t trt_ctz32(uint32_t x) { return (uint32_t)__builtin_ctz(x); }
static inline uint32_t trt_popc32(uint32_t x) { return (uint32_t)__builtin_popcount(x); }
#endif

static inline TRT_HD uint32_t trt_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
static inline TRT_HD float trt_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

/* ---- counter-based RNG ------------------------------------------------------ */

/* 32-bit finaliser (full-avalanche integer hash). */
static inline TRT_HD uint32_t trt_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

typedef struct trt_rng_key { uint32_t k0, k1; } trt_rng_key;

/* 64-bit stream key of one camera sample: (seed, pixel index y*W+x, sample k). */
static inline TRT_HD trt_rng_key trt_rng_make_key(uint32_t seed, uint32_t pixel, uint32_t sample)
{
    trt_rng_key k;
    uint32_t a = trt_mix32(seed ^ 0x9E3779B9u);
    uint32_t b = trt_mix32(pixel + 0x85EBCA6Bu);
    uint32_t c = trt_mix32(sample + 0xC2B2AE35u);
    k.k0 = trt_mix32(a ^ (b + 0x27D4EB2Fu) ^ (c << 1 | c >> 31));
    k.k1 = trt_mix32((a << 7 | a >> 25) + b * 0x9E3779B1u + (c ^ 0x165667B1u));
    return k;
}

/* i-th 32-bit word of the stream. */
static inline TRT_HD uint32_t trt_rng_u32(trt_rng_key k, uint32_t i)
{
    uint32_t x = trt_mix32(k.k1 + i * 0x9E3779B9u);
    return trt_mix32(x ^ k.k0);
}

/* i-th uniform in [0,1): 24 random mantissa bits, exactly representable. */
static inline TRT_HD float trt_rng_uniform(trt_rng_key k, uint32_t i)
{
    return (float)(trt_rng_u32(k, i) >> 8) * 5.9604644775390625e-8f;
}

/* ---- fp32 transcendental functions as fixed FMA sequences -------------------- */

/* (cos, sin) of 2*pi*u for u in [0,1).  Quadrant reduction is exact; the
 * residual angle in [-pi/4, pi/4] goes through the classic single-precision
 * minimax polynomials (abs error < 1.2e-7). */
static inline TRT_HD void trt_sincos2pi(float u, float* c_out, float* s_out)
{
    float x = 4.0f * u;                 /* exact */
    int k = (int)(x + 0.5f);            /* nearest quadrant 0..4 */
    float r = x - (float)k;             /* exact, in [-0.5, 0.5] */
    float a = r * 1.57079637050628662109375f;
    float z = a * a;
    float sp = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float s = fmaf(sp * z, a, a);
    float cp = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float c = fmaf(cp * z, z, fmaf(-0.5f, z, 1.0f));
    switch (k & 3) {
    case 0: *c_out = c;  *s_out = s;  break;
    case 1: *c_out = -s; *s_out = c;  break;
    case 2: *c_out = -c; *s_out = -s; break;
    default: *c_out = s; *s_out = -c; break;
    }
}

/* natural log of a positive normal float (abs/rel error ~1e-7, relative near 1). */
static inline TRT_HD float trt_logf(float x)
{
    uint32_t b = trt_f2u(x);
    int e = (int)((b >> 23) & 0xffu) - 126;
    float m = trt_u2f((b & 0x007fffffu) | 0x3f000000u);   /* [0.5, 1) */
    float f;
    if (m < 0.707106781186547524f) { e -= 1; f = (m + m) - 1.0f; }
    else { f = m - 1.0f; }
    float z = f * f;
    float p = 7.0376836292e-2f;
    p = fmaf(p, f, -1.1514610310e-1f);
    p = fmaf(p, f, 1.1676998740e-1f);
    p = fmaf(p, f, -1.2420140846e-1f);
    p = fmaf(p, f, 1.4249322787e-1f);
    p = fmaf(p, f, -1.6668057665e-1f);
    p = fmaf(p, f, 2.0000714765e-1f);
    p = fmaf(p, f, -2.4999993993e-1f);
    p = fmaf(p, f, 3.3333331174e-1f);
    float y = (p * f) * z;
    float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    return fmaf(fe, 0.693359375f, f + y);
}

/* e^x for x <= 0 (the only range the path needs); flushes to 0 below 2^-126. */
static inline TRT_HD float trt_expf_neg(float x)
{
    if (!(x > -87.0f)) return 0.0f;
    float fn = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(fn, -0.693359375f, x);
    r = fmaf(fn, 2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p, z, r) + 1.0f;
    int n = (int)fn;
    if (n < -125) return 0.0f;
    return trt_u2f(trt_f2u(y) + ((uint32_t)n << 23));    /* y in [0.5,2): exponent stays normal */
}

/* x^y for x in [0,1], y > 0: Blinn-Phong lobe pow(cos_alpha, Ns)
 * (pathTracing.cpp:69) and the lobe inverse CDF pow(u, 1/(Ns+1))
 * (pathTracing.cpp:125).  y == 1 is exact. */
static inline TRT_HD float trt_pow01(float x, float y)
{
    if (y == 1.0f) return x;
    if (!(x > 1.17549435e-38f)) return 0.0f;
    if (x >= 1.0f) return 1.0f;
    return trt_expf_neg(y * trt_logf(x));
}

/* ---- the leaf-box rule with a tolerance (DESIGN.md "Formulation") ----------------------------------------------------
 * A triangle hit at distance t counts iff NOT t < trt_leaf_floor(e, alpha), e = the slab entry distance of the caller's box
 * of the leaf the triangle lies in.  The rule exists to make ordered, culled traversal return the unculled traversal's
 * hit for EVERY input (for rays within ~1e-4 rad of a triangle's plane Moller-Trumbore's tn/det can come out far in
 * front of the triangle and of every box around it); it is not meant to judge hits, so the floor lies well below the
 * entry: 2^-16 relative plus alpha = 2^-17 of the scene's largest coordinate.  That absorbs the rounding differences
 * between the two computations of one distance — a triangle lying ON a face of its (unpadded) leaf box, coordinates of
 * 4e4 where one ulp exceeds the reference's 0.001 pad — which a bare `t < e` rejected about half of the time.
 * Monotone under rounding: e -> e * k is non-decreasing for a fixed k > 0 (negative products stay <= 0 <= positive
 * ones), x -> x - alpha is non-decreasing; so for nested boxes floor(entry of the leaf) >= floor(entry of any box above). */
#define TRT_LEAF_KPOS 0.9999847412109375f  /* 1 - 2^-16 */
#define TRT_LEAF_KNEG 1.0000152587890625f  /* 1 + 2^-16 */
static inline TRT_HD float trt_leaf_floor(float e, float alpha)
{
    return e * (e < 0.0f ? TRT_LEAF_KNEG : TRT_LEAF_KPOS) - alpha;
}
/* alpha of a scene whose root boxes reach |coordinate| <= m (exact: a power-of-two scaling). */
static inline TRT_HD float trt_leaf_alpha(float m) { return m * 7.62939453125e-6f; /* 2^-17 */ }
/* Culling bound that goes with the rule: a node whose entry distance e satisfies e > trt_cull_bound(b, alpha) holds no hit
 * that counts and is <= b, for b >= 0: then e > 0, and trt_leaf_floor(e) >= (e (1 - 2^-16)(1 - u) - alpha)(1 - u) with
 * e >= (b + 2 alpha)(1 + 2^-14)(1 - u)^2 (u = 2^-24), i.e. > b (1 + 2.9 * 2^-16) + alpha > b; every hit below the node
 * that counts has t >= floor(entry of its leaf) >= floor(e) > b.  (b = +inf: never culls.) */
static inline TRT_HD float trt_cull_bound(float b, float alpha)
{
    return (b + (alpha + alpha)) * 1.00006103515625f;  /* 1 + 2^-14 */
}

#endif /* TRT_PRIMS_H */
