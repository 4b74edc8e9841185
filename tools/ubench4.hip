// ubench4.hip -- v_mfma_f32_16x16x32_bf16 issue rate alone and with independent VALU adds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define MF(D) "v_mfma_f32_16x16x32_bf16 " D ", %8, %9, 0\n"
#define AD(a) "v_add_f32 %" #a ", |%10|, %" #a "\n"
template <int NADD>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b)
{
    f32x4 d0 = {0,0,0,0}, d1 = d0, d2 = d0, d3 = d0;
    f32x4 av = {a + threadIdx.x, a, b, a}, bv = {b + threadIdx.x, b, a, b};
    float e = a, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int i = 0; i < iters; ++i) {
        if (NADD == 0) asm volatile(MF("%0") MF("%1") MF("%2") MF("%3") MF("%0") MF("%1") MF("%2") MF("%3")
            : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv), "v"(e));
        if (NADD == 2) asm volatile(MF("%0") AD(4) AD(5) MF("%1") AD(6) AD(7) MF("%2") AD(4) AD(5) MF("%3") AD(6) AD(7) MF("%0") AD(4) AD(5) MF("%1") AD(6) AD(7) MF("%2") AD(4) AD(5) MF("%3") AD(6) AD(7)
            : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv), "v"(e));
        if (NADD == 4) asm volatile(MF("%0") AD(4) AD(5) AD(6) AD(7) MF("%1") AD(4) AD(5) AD(6) AD(7) MF("%2") AD(4) AD(5) AD(6) AD(7) MF("%3") AD(4) AD(5) AD(6) AD(7) MF("%0") AD(4) AD(5) AD(6) AD(7) MF("%1") AD(4) AD(5) AD(6) AD(7) MF("%2") AD(4) AD(5) AD(6) AD(7) MF("%3") AD(4) AD(5) AD(6) AD(7)
            : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv), "v"(e));
        if (NADD == 5) asm volatile(MF("%0") AD(4) AD(5) AD(6) AD(7) AD(4) MF("%1") AD(5) AD(6) AD(7) AD(4) AD(5) MF("%2") AD(6) AD(7) AD(4) AD(5) AD(6) MF("%3") AD(7) AD(4) AD(5) AD(6) AD(7) MF("%0") AD(4) AD(5) AD(6) AD(7) AD(4) MF("%1") AD(5) AD(6) AD(7) AD(4) AD(5) MF("%2") AD(6) AD(7) AD(4) AD(5) AD(6) MF("%3") AD(7) AD(4) AD(5) AD(6) AD(7)
            : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv), "v"(e));
        if (NADD == 44) // bursts: 4 MFMA then 16 adds
            asm volatile(MF("%0") MF("%1") MF("%2") MF("%3") AD(4) AD(5) AD(6) AD(7) AD(4) AD(5) AD(6) AD(7) AD(4) AD(5) AD(6) AD(7) AD(4) AD(5) AD(6) AD(7) MF("%0") MF("%1") MF("%2") MF("%3") AD(4) AD(5) AD(6) AD(7) AD(4) AD(5) AD(6) AD(7) AD(4) AD(5) AD(6) AD(7) AD(4) AD(5) AD(6) AD(7)
            : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv), "v"(e));
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0[0] + d1[1] + d2[2] + d3[3] + c0 + c1 + c2 + c3;
}
template <int N> double run(float* out, int blocks, int iters)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < 3; ++r) { CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    return best;
}
int main()
{
    float* out; CHECK(hipMalloc(&out, 1 << 24));
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = 256 * wps;
        auto cyc = [&](double ms) { return ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wps); };
        printf("wps %d bf16 16x16x32 cyc/MFMA/SIMD: alone %.1f  +2 adds %.1f  +4 adds %.1f  +5 adds %.1f  burst(4 MFMA,16 adds) %.1f\n", wps,
               cyc(run<0>(out, blocks, iters)), cyc(run<2>(out, blocks, iters)), cyc(run<4>(out, blocks, iters)), cyc(run<5>(out, blocks, iters)), cyc(run<44>(out, blocks, iters)));
    }
    return 0;
}
