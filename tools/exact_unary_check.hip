// exact_unary_check.hip — exhaustive proof, over all 2^32 binary32 inputs, that the short correction sequences of
// include/trt_exact.h return the bits of the IEEE-754 correctly rounded operations they stand for on gfx950:
//   trt_sqrt(x)            == sqrtf(x)
//   trt_sqrt_rsqrt2(x)     == (sqrtf(x), 1.0f / sqrtf(x))        (two roundings, as glm::normalize does it)
//   trt_div_by(a, b, 1/b)  == a / b in binary64, for every (a, b) the pixel grid of cameraRay() can form (1.1e12 pairs)
// The reference side is what hipcc emits for the plain expressions (correctly rounded division and square root, the HIP
// default).  A unary fp32 function has 2^32 inputs: the check is a proof, not a sample.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Iinclude -o tools/exact_unary_check tools/exact_unary_check.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>

#include "trt_exact.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ inline uint32_t bitsOf(float f) { return __float_as_uint(f); }
__device__ inline bool sameBits(float a, float b) { return bitsOf(a) == bitsOf(b) || (a != a && b != b); }  // any NaN equals any NaN

// which: 0 trt_sqrt, 1 trt_rsqrt2, 2 / 3 the two results of trt_sqrt_rsqrt2; raw != 0: the fast paths without their input guard
__global__ __launch_bounds__(256) void k_check(int which, int raw, unsigned long long* n_bad, unsigned long long* bad_by_exp, uint32_t* first_bad)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned long long bad = 0, seen = 0;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t it = 0; it < (uint32_t)(0x100000000ull / stride); ++it, i += stride) {
        const float x = __uint_as_float(i);
        float ref, got;
        seen++;
        if (which == 0) { ref = sqrtf(x); got = raw ? trt_sqrt_fast(x) : trt_sqrt(x); }
        else if (which == 1) { ref = 1.0f / sqrtf(x); got = raw ? (trt_sqrt_rsqrt2_fast(x, &ref, &got), got) : trt_rsqrt2(x); ref = 1.0f / sqrtf(x); }
        else {
            float s, r;
            if (raw) trt_sqrt_rsqrt2_fast(x, &s, &r); else trt_sqrt_rsqrt2(x, &s, &r);
            ref = which == 2 ? sqrtf(x) : 1.0f / sqrtf(x);
            got = which == 2 ? s : r;
        }
        if (!sameBits(ref, got)) {
            bad++;
            atomicAdd(&bad_by_exp[(i >> 23) & 0x1FF], 1ull);
            atomicMin(first_bad, i);
        }
    }
    if (bad) atomicAdd(n_bad, bad);
    atomicAdd(n_bad + 1, seen);
}


// trt_div_by(a, b, 1 / b) == a / b in binary64 for every operand pair the pixel grid of cameraRay() can form:
// b = blockIdx.x + 1 in [1, 65536]; a = an integer in [0, 65536] (pixel column / row counts) or m 2^-24, m in [-2^23, 2^23)
// ((u - 0.5) for a 24-bit uniform u).  1.1e12 pairs.
__global__ __launch_bounds__(256) void k_check_div(unsigned long long* n_bad, unsigned long long* n_seen, unsigned long long* first_bad)
{
    const double b = (double)(blockIdx.x + 1u);
    const double rb = 1.0 / b;
    unsigned long long bad = 0, seen = 0;
    for (uint32_t k = threadIdx.x; k < 65537u + (1u << 24); k += 256u) {
        const double a = k <= 65536u ? (double)k : ((double)(int)(k - 65537u) - 8388608.0) * 5.9604644775390625e-8;
        const double ref = a / b, got = trt_div_by(a, b, rb);
        seen++;
        if (__double_as_longlong(ref) != __double_as_longlong(got)) {
            bad++;
            atomicMin(first_bad, ((unsigned long long)(blockIdx.x + 1u) << 32) | k);
        }
    }
    if (bad) atomicAdd(n_bad, bad);
    atomicAdd(n_seen, seen);
}

int main(int argc, char** argv)
{
    const int raw = argc > 1 && std::atoi(argv[1]) != 0;
    CK(hipSetDevice(0));
    unsigned long long *d_bad, *d_hist;
    uint32_t* d_first;
    CK(hipMalloc(&d_bad, 16));
    CK(hipMalloc(&d_hist, 512 * 8));
    CK(hipMalloc(&d_first, 4));
    const char* names[] = {"trt_sqrt == sqrtf(x)", "trt_rsqrt2 == 1.0f / sqrtf(x)", "trt_sqrt_rsqrt2: sqrt", "trt_sqrt_rsqrt2: 1/sqrt"};
    int rc = 0;
    for (int which = 0; which < 4; ++which) {
        CK(hipMemset(d_bad, 0, 16));
        CK(hipMemset(d_hist, 0, 512 * 8));
        CK(hipMemset(d_first, 0xFF, 4));
        hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, which, raw, d_bad, d_hist, d_first);  // 2^20 threads x 4096 iterations = 2^32 inputs
        CK(hipDeviceSynchronize());
        unsigned long long bad2[2] = {0, 0}, hist[512];
        uint32_t first = 0;
        CK(hipMemcpy(bad2, d_bad, 16, hipMemcpyDeviceToHost));
        const unsigned long long bad = bad2[0];
        if (bad2[1] != 0x100000000ull) { std::printf("internal error: %llu inputs visited\n", bad2[1]); return 2; }
        CK(hipMemcpy(hist, d_hist, sizeof hist, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost));
        std::printf("%-32s %s: %llu of 4294967296 inputs differ", names[which], raw ? "(fast path, unguarded)" : "(as shipped)", bad);
        if (bad) {
            std::printf("; first 0x%08x; by sign|exponent field:", first);
            int shown = 0;
            for (int e = 0; e < 512 && shown < 24; ++e)
                if (hist[e]) { std::printf(" %s%d:%llu", e >= 256 ? "-" : "", e & 255, hist[e]); shown++; }
        }
        std::printf("\n");
        if (bad && !raw) rc = 1;
    }
    {
        unsigned long long* d3;
        CK(hipMalloc(&d3, 24));
        CK(hipMemset(d3, 0, 16));
        CK(hipMemset(d3 + 2, 0xFF, 8));
        hipLaunchKernelGGL(k_check_div, dim3(65536), dim3(256), 0, 0, d3, d3 + 1, d3 + 2);
        CK(hipDeviceSynchronize());
        unsigned long long h3[3];
        CK(hipMemcpy(h3, d3, 24, hipMemcpyDeviceToHost));
        const unsigned long long want = 65536ull * (65537ull + (1ull << 24));
        if (h3[1] != want) { std::printf("internal error: %llu operand pairs visited, not %llu\n", h3[1], want); return 2; }
        std::printf("%-32s (as shipped): %llu of %llu operand pairs differ", "trt_div_by == a / b (binary64)", h3[0], want);
        if (h3[0]) { std::printf("; first b=%llu k=%llu", h3[2] >> 32, h3[2] & 0xFFFFFFFFull); rc = 1; }
        std::printf("\n");
    }
    return rc;
}
