// valu_probe.hip — issue rate of the VALU instruction kinds the traversal and shading kernels are made of, per SIMD of gfx950:
// plain v_fma_f32, packed v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, v_cndmask / v_cmp, v_max3, v_rcp, v_mul_lo_u32.
// Inline asm so that the compiler neither packs nor folds anything.  8 independent chains per lane; waves/SIMD = 1, 2, 4, 8.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/valu_probe tools/valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k_issue(uint32_t iters, float* __restrict__ sink)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 1.0000001f, c = 1e-9f;
    const v2f pm = {m, m}, pc = {c, c};
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\nv_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\nv_pk_fma_f32 %3, %3, %8, %9\nv_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\nv_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));
            if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %8\nv_pk_mul_f32 %1, %1, %8\nv_pk_mul_f32 %2, %2, %8\nv_pk_mul_f32 %3, %3, %8\nv_pk_mul_f32 %4, %4, %8\nv_pk_mul_f32 %5, %5, %8\nv_pk_mul_f32 %6, %6, %8\nv_pk_mul_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));
            if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %8\nv_pk_add_f32 %1, %1, %8\nv_pk_add_f32 %2, %2, %8\nv_pk_add_f32 %3, %3, %8\nv_pk_add_f32 %4, %4, %8\nv_pk_add_f32 %5, %5, %8\nv_pk_add_f32 %6, %6, %8\nv_pk_add_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
            if (KIND == 4) asm volatile("v_cmp_lt_f32 vcc, %0, %8\nv_cndmask_b32 %0, %0, %9, vcc\nv_cmp_lt_f32 vcc, %1, %8\nv_cndmask_b32 %1, %1, %9, vcc\nv_cmp_lt_f32 vcc, %2, %8\nv_cndmask_b32 %2, %2, %9, vcc\nv_cmp_lt_f32 vcc, %3, %8\nv_cndmask_b32 %3, %3, %9, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc");
            if (KIND == 5) asm volatile("v_max3_f32 %0, %0, %8, %9\nv_max3_f32 %1, %1, %8, %9\nv_max3_f32 %2, %2, %8, %9\nv_max3_f32 %3, %3, %8, %9\nv_max3_f32 %4, %4, %8, %9\nv_max3_f32 %5, %5, %8, %9\nv_max3_f32 %6, %6, %8, %9\nv_max3_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            if (KIND == 6) asm volatile("v_rcp_f32 %0, %0\nv_rcp_f32 %1, %1\nv_rcp_f32 %2, %2\nv_rcp_f32 %3, %3\nv_rcp_f32 %4, %4\nv_rcp_f32 %5, %5\nv_rcp_f32 %6, %6\nv_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            if (KIND == 7) asm volatile("v_mul_f32 %0, %0, %8\nv_mul_f32 %1, %1, %8\nv_mul_f32 %2, %2, %8\nv_mul_f32 %3, %3, %8\nv_mul_f32 %4, %4, %8\nv_mul_f32 %5, %5, %8\nv_mul_f32 %6, %6, %8\nv_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            if (KIND == 8) asm volatile("v_xor_b32 %0, %0, %8\nv_xor_b32 %1, %1, %8\nv_xor_b32 %2, %2, %8\nv_xor_b32 %3, %3, %8\nv_xor_b32 %4, %4, %8\nv_xor_b32 %5, %5, %8\nv_xor_b32 %6, %6, %8\nv_xor_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            if (KIND == 9) asm volatile("v_fmac_f32_e32 %0, %8, %9\nv_fmac_f32_e32 %1, %8, %9\nv_fmac_f32_e32 %2, %8, %9\nv_fmac_f32_e32 %3, %8, %9\nv_fmac_f32_e32 %4, %8, %9\nv_fmac_f32_e32 %5, %8, %9\nv_fmac_f32_e32 %6, %8, %9\nv_fmac_f32_e32 %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            if (KIND == 10) asm volatile("v_mul_f32_e64 %0, %0, -%8\nv_mul_f32_e64 %1, %1, -%8\nv_mul_f32_e64 %2, %2, -%8\nv_mul_f32_e64 %3, %3, -%8\nv_mul_f32_e64 %4, %4, -%8\nv_mul_f32_e64 %5, %5, -%8\nv_mul_f32_e64 %6, %6, -%8\nv_mul_f32_e64 %7, %7, -%8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            if (KIND == 11) asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\nv_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m), "v"(c));
            if (KIND == 12) asm volatile("v_fmac_f32_e32 %0, %8, %9\nv_fmac_f32_e32 %1, %8, %9\nv_fmac_f32_e32 %2, %8, %9\nv_fmac_f32_e32 %3, %8, %9\nv_fmac_f32_e32 %4, %8, %9\nv_fmac_f32_e32 %5, %8, %9\nv_fmac_f32_e32 %6, %8, %9\nv_fmac_f32_e32 %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m), "v"(c));
            if (KIND == 13) asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\nv_cndmask_b32_e32 %1, %1, %8, vcc\nv_cndmask_b32_e32 %2, %2, %8, vcc\nv_cndmask_b32_e32 %3, %3, %8, vcc\nv_cndmask_b32_e32 %4, %4, %8, vcc\nv_cndmask_b32_e32 %5, %5, %8, vcc\nv_cndmask_b32_e32 %6, %6, %8, vcc\nv_cndmask_b32_e32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
            if (KIND == 14) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %8\nv_cmp_lt_f32_e32 vcc, %1, %8\nv_cmp_lt_f32_e32 vcc, %2, %8\nv_cmp_lt_f32_e32 vcc, %3, %8\nv_cmp_lt_f32_e32 vcc, %4, %8\nv_cmp_lt_f32_e32 vcc, %5, %8\nv_cmp_lt_f32_e32 vcc, %6, %8\nv_cmp_lt_f32_e32 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
            if (KIND == 15) asm volatile("v_min_f32_e32 %0, %0, %8\nv_max_f32_e32 %1, %1, %8\nv_min_f32_e32 %2, %2, %8\nv_max_f32_e32 %3, %3, %8\nv_min_f32_e32 %4, %4, %8\nv_max_f32_e32 %5, %5, %8\nv_min_f32_e32 %6, %6, %8\nv_max_f32_e32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        }
    }
    const float s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)) + (p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y);
    if (s == 123.456f) sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    CK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    std::printf("# device %s, %d CUs, nominal %.0f MHz\n", prop.gcnArchName, cus, clk / 1e6);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float* sink = nullptr;
    CK(hipMalloc(&sink, (size_t)1 << 24));
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_cmp+v_cndmask", "v_max3_f32", "v_rcp_f32", "v_mul_f32", "v_xor_b32", "v_fmac_f32_e32", "v_mul_f32_e64 (neg)", "v_fma_f32 (sgpr src)", "v_fmac_e32 (sgpr src)", "v_cndmask_e32", "v_cmp_e32", "v_min/max_f32_e32"};
    const uint32_t iters = 2048;
    for (int kind = 0; kind < 16; ++kind)
        for (int bpc : {2, 8}) {
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                const dim3 g((uint32_t)cus * bpc), b(256);
                switch (kind) {
                    case 0: hipLaunchKernelGGL(k_issue<0>, g, b, 0, 0, iters, sink); break;
                    case 1: hipLaunchKernelGGL(k_issue<1>, g, b, 0, 0, iters, sink); break;
                    case 2: hipLaunchKernelGGL(k_issue<2>, g, b, 0, 0, iters, sink); break;
                    case 3: hipLaunchKernelGGL(k_issue<3>, g, b, 0, 0, iters, sink); break;
                    case 4: hipLaunchKernelGGL(k_issue<4>, g, b, 0, 0, iters, sink); break;
                    case 5: hipLaunchKernelGGL(k_issue<5>, g, b, 0, 0, iters, sink); break;
                    case 6: hipLaunchKernelGGL(k_issue<6>, g, b, 0, 0, iters, sink); break;
                    case 7: hipLaunchKernelGGL(k_issue<7>, g, b, 0, 0, iters, sink); break;
                    case 8: hipLaunchKernelGGL(k_issue<8>, g, b, 0, 0, iters, sink); break;
                    case 9: hipLaunchKernelGGL(k_issue<9>, g, b, 0, 0, iters, sink); break;
                    case 10: hipLaunchKernelGGL(k_issue<10>, g, b, 0, 0, iters, sink); break;
                    case 11: hipLaunchKernelGGL(k_issue<11>, g, b, 0, 0, iters, sink); break;
                    case 12: hipLaunchKernelGGL(k_issue<12>, g, b, 0, 0, iters, sink); break;
                    case 13: hipLaunchKernelGGL(k_issue<13>, g, b, 0, 0, iters, sink); break;
                    case 14: hipLaunchKernelGGL(k_issue<14>, g, b, 0, 0, iters, sink); break;
                    default: hipLaunchKernelGGL(k_issue<15>, g, b, 0, 0, iters, sink); break;
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double instr = (double)cus * bpc * 4 * iters * 64.0;  // wave-level instructions
            std::printf("%-18s waves/SIMD=%d  %8.3f ms  %6.3f wave-instr/clk/SIMD = one per %.2f clocks (nominal clock)\n", names[kind], bpc, ms, instr / (ms * 1e-3 * clk) / (cus * 4.0),
                        (ms * 1e-3 * clk) * (cus * 4.0) / instr);
        }
    return 0;
}
