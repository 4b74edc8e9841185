// ubench2.hip -- how the abs-accumulate adds should be grouped around
// v_mfma_f32_16x16x4_f32: M MFMAs back to back, then 4*M v_add_f32 |.|.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define MF(D) "v_mfma_f32_16x16x4_f32 " D ", %8, %9, 0\n"
#define AD4(a, D) "v_add_f32 %" #a ", |" D "|, %" #a "\n"
// plain (no abs, e32 4-byte) adds for comparison
#define ADP(a) "v_add_f32_e32 %" #a ", %10, %" #a "\n"

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b)
{
    f32x4 d0 = {0,0,0,0}, d1 = d0, d2 = d0, d3 = d0;
    float av = a + threadIdx.x, bv = b + threadIdx.x, e = a;
    float c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)      // M=1: MFMA, 4 adds (independent source e)
            asm volatile(MF("%0") ADP(4) ADP(5) ADP(6) ADP(7) MF("%1") ADP(4) ADP(5) ADP(6) ADP(7) MF("%2") ADP(4) ADP(5) ADP(6) ADP(7) MF("%3") ADP(4) ADP(5) ADP(6) ADP(7)
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv), "v"(e));
        if (MODE == 1)      // M=2
            asm volatile(MF("%0") MF("%1") ADP(4) ADP(5) ADP(6) ADP(7) ADP(4) ADP(5) ADP(6) ADP(7) MF("%2") MF("%3") ADP(4) ADP(5) ADP(6) ADP(7) ADP(4) ADP(5) ADP(6) ADP(7)
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv), "v"(e));
        if (MODE == 2)      // M=4
            asm volatile(MF("%0") MF("%1") MF("%2") MF("%3") ADP(4) ADP(5) ADP(6) ADP(7) ADP(4) ADP(5) ADP(6) ADP(7) ADP(4) ADP(5) ADP(6) ADP(7) ADP(4) ADP(5) ADP(6) ADP(7)
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv), "v"(e));
        if (MODE == 3)      // 16 plain e32 adds only
            asm volatile(ADP(4) ADP(5) ADP(6) ADP(7) ADP(4) ADP(5) ADP(6) ADP(7) ADP(4) ADP(5) ADP(6) ADP(7) ADP(4) ADP(5) ADP(6) ADP(7)
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv), "v"(e));
        if (MODE == 4)      // 16 fmac e32 only (8 independent accumulators would need more regs; 4 here)
            asm volatile("v_fmac_f32_e32 %4, %10, %8\n v_fmac_f32_e32 %5, %10, %9\n v_fmac_f32_e32 %6, %10, %8\n v_fmac_f32_e32 %7, %10, %9\n"
                         "v_fmac_f32_e32 %4, %10, %8\n v_fmac_f32_e32 %5, %10, %9\n v_fmac_f32_e32 %6, %10, %8\n v_fmac_f32_e32 %7, %10, %9\n"
                         "v_fmac_f32_e32 %4, %10, %8\n v_fmac_f32_e32 %5, %10, %9\n v_fmac_f32_e32 %6, %10, %8\n v_fmac_f32_e32 %7, %10, %9\n"
                         "v_fmac_f32_e32 %4, %10, %8\n v_fmac_f32_e32 %5, %10, %9\n v_fmac_f32_e32 %6, %10, %8\n v_fmac_f32_e32 %7, %10, %9\n"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv), "v"(e));
        if (MODE == 5)      // M=4 with VOP3 abs adds (8-byte encodings) on an independent source
            asm volatile(MF("%0") MF("%1") MF("%2") MF("%3") AD4(4, "%10") AD4(5, "%10") AD4(6, "%10") AD4(7, "%10") AD4(4, "%10") AD4(5, "%10") AD4(6, "%10") AD4(7, "%10")
                         AD4(4, "%10") AD4(5, "%10") AD4(6, "%10") AD4(7, "%10") AD4(4, "%10") AD4(5, "%10") AD4(6, "%10") AD4(7, "%10")
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv), "v"(e));
        if (MODE == 6)      // 8 independent e32 adds chains? (uses d regs as accumulators too)
            asm volatile("v_add_f32_e32 %4, %10, %4\n v_add_f32_e32 %5, %10, %5\n v_add_f32_e32 %6, %10, %6\n v_add_f32_e32 %7, %10, %7\n"
                         "v_add_f32_e32 %8, %10, %8\n v_add_f32_e32 %9, %10, %9\n v_add_f32_e32 %4, %10, %4\n v_add_f32_e32 %5, %10, %5\n"
                         "v_add_f32_e32 %6, %10, %6\n v_add_f32_e32 %7, %10, %7\n v_add_f32_e32 %8, %10, %8\n v_add_f32_e32 %9, %10, %9\n"
                         "v_add_f32_e32 %4, %10, %4\n v_add_f32_e32 %5, %10, %5\n v_add_f32_e32 %6, %10, %6\n v_add_f32_e32 %7, %10, %7\n"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(av), "+v"(bv) : "v"(e));
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0[0] + d1[1] + d2[2] + d3[3] + c0 + c1 + c2 + c3 + av + bv;
}

template <int MODE> double run(float* out, int blocks, int iters)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    float* out; CHECK(hipMalloc(&out, 1 << 24));
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = 256 * wps;
        auto cyc = [&](double ms) { return ms * 1e-3 * 2.4e9 / ((double)iters * wps); };  // cycles per loop trip per SIMD
        printf("wps %d  per 4 MFMA+16 adds: M=1 %.1f  M=2 %.1f  M=4 %.1f | 16 add_e32 %.1f  16 fmac_e32 %.1f | M=4 with abs(VOP3) adds %.1f | 16 add 6-chain %.1f\n", wps,
               cyc(run<0>(out, blocks, iters)), cyc(run<1>(out, blocks, iters)), cyc(run<2>(out, blocks, iters)),
               cyc(run<3>(out, blocks, iters)), cyc(run<4>(out, blocks, iters)), cyc(run<5>(out, blocks, iters)), cyc(run<6>(out, blocks, iters)));
    }
    return 0;
}
