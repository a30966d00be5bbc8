/*
 * trt_exact.h — correctly rounded fp32 square root and reciprocal square root, and binary64 division by a divisor whose
 * reciprocal is known, in fewer instructions than the general-purpose expansions hipcc emits for `sqrtf(x)`, `1.0f / sqrtf(x)`
 * and `a / b`.
 *
 * The kernels are bound by VALU issue, and a path vertex normalises six vectors: glm::normalize is
 * v * (1 / sqrt(dot(v, v))) — a correctly rounded square root (15 instructions on gfx950) followed by a correctly rounded
 * division (11 more, v_rcp_f32 and two v_div_scale among them).  Both are UNARY functions of one binary32 value, so a
 * shorter sequence can be PROVEN to return the same bits by trying all 2^32 inputs: tools/exact_unary_check.hip does that on
 * the GPU (tests/test_gpu_parity.py runs it), against hipcc's own expansions.  Inputs outside the range where the short
 * sequence is exact (zeros, denormals, infinities, NaNs, the ends of the exponent range) take the plain expression.
 *
 * On the host (oracle, hostsim, loaders) these are the plain expressions: the results are bit-identical by the proof above,
 * so nothing outside the kernels can tell which form ran.
 */
#ifndef TRT_EXACT_H
#define TRT_EXACT_H

#include <math.h>
#include <stdint.h>

#include "trt_prims.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define TRT_EXACT_DEVICE 1
#else
#define TRT_EXACT_DEVICE 0
#endif

/* x is positive with an exponent field in [32, 254], i.e. 2^-95 <= x < 2^128 (an add and an unsigned compare on the bits):
 * the inputs the short sequences are proven for.  Below 2^-102 the residual x - s1^2 (about 2^-24 x) becomes denormal and the
 * last step no longer rounds correctly; zeros, denormals, negative numbers, infinities and NaNs are excluded as well. */
static inline TRT_HD bool trt_exact_domain(float x)
{
    return trt_f2u(x) - (32u << 23) < ((255u - 32u) << 23);
}

/* sqrt(x), correctly rounded, from one v_rsq_f32 (1 ulp) and a coupled Newton step on the root and on half its reciprocal;
 * the last fma adds the residual x - s1^2 (exact in an fma) times h1 ~ 1 / (2 sqrt x). */
static inline TRT_HD float trt_sqrt_fast(float x)
{
#if TRT_EXACT_DEVICE
    const float r0 = __builtin_amdgcn_rsqf(x);
    const float s0 = x * r0;
    const float h0 = 0.5f * r0;
    const float e0 = __builtin_fmaf(-s0, h0, 0.5f);
    const float s1 = __builtin_fmaf(s0, e0, s0);
    const float h1 = __builtin_fmaf(h0, e0, h0);
    const float d1 = __builtin_fmaf(-s1, s1, x);
    return __builtin_fmaf(d1, h1, s1);
#else
    return sqrtf(x);
#endif
}

/* s = sqrt(x) and r = 1 / s (the reciprocal of the ROUNDED root: two roundings, as glm::normalize has them).  The reciprocal
 * starts from 2 h1 and takes two Newton steps.  One family of inputs defeats any such iteration: a root whose significand is
 * all ones, 1 / s = (1 + 2^-24 + ...) * 2^k, where the last fma hits an exact tie and rounds to even, one ulp low
 * (Markstein's exceptional case); the low bit is added back for exactly those s. */
static inline TRT_HD void trt_sqrt_rsqrt2_fast(float x, float* s_out, float* r_out)
{
#if TRT_EXACT_DEVICE
    const float r0 = __builtin_amdgcn_rsqf(x);
    const float s0 = x * r0;
    const float h0 = 0.5f * r0;
    const float e0 = __builtin_fmaf(-s0, h0, 0.5f);
    const float s1 = __builtin_fmaf(s0, e0, s0);
    const float h1 = __builtin_fmaf(h0, e0, h0);
    const float d1 = __builtin_fmaf(-s1, s1, x);
    const float s = __builtin_fmaf(d1, h1, s1);
    float r = h1 + h1;
    float e = __builtin_fmaf(-s, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-s, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    *s_out = s;
    *r_out = trt_u2f(trt_f2u(r) + ((trt_f2u(s) & 0x7FFFFFu) == 0x7FFFFFu ? 1u : 0u));
#else
    *s_out = sqrtf(x);
    *r_out = 1.0f / *s_out;
#endif
}

static inline TRT_HD float trt_sqrt(float x)
{
#if TRT_EXACT_DEVICE
    if (trt_exact_domain(x)) return trt_sqrt_fast(x);
#endif
    return sqrtf(x);
}
static inline TRT_HD void trt_sqrt_rsqrt2(float x, float* s, float* r)
{
#if TRT_EXACT_DEVICE
    if (trt_exact_domain(x)) { trt_sqrt_rsqrt2_fast(x, s, r); return; }
#endif
    *s = sqrtf(x);
    *r = 1.0f / *s;
}
static inline TRT_HD float trt_rsqrt2(float x)
{
    float s, r;
    trt_sqrt_rsqrt2(x, &s, &r);
    return r;
}

/* a / b in binary64 when the correctly rounded reciprocal rb = 1.0 / b of the divisor is at hand (the pixel grid of
 * main.cpp:88-93 divides by W - 1, H - 1, W and H: wave-uniform, formed once per render on the host): one product, the exact
 * residual a - q0 b in an fma, one correction — Markstein's theorem makes q the correctly rounded quotient whenever rb is the
 * correctly rounded reciprocal and b's significand is not all ones; for the operands the camera-ray set-up feeds it
 * (b an integer in [1, 65536]; a an integer in [0, 65536] or m 2^-24 with |m| <= 2^23) tools/exact_unary_check.hip tries every
 * pair.  Three binary64 instructions instead of the eleven of the general division. */
static inline TRT_HD double trt_div_by(double a, double b, double rb)
{
#if TRT_EXACT_DEVICE
    const double q0 = a * rb;
    const double r = __builtin_fma(-q0, b, a);
    return __builtin_fma(r, rb, q0);
#else
    (void)rb;
    return a / b;
#endif
}

#endif /* TRT_EXACT_H */
