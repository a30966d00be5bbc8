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

#endif /* TRT_PRIMS_H */
