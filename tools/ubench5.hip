// ubench5.hip -- v_mfma_f32_32x32x16_bf16 chained pairs (K = 32) with the 16 |d| adds per pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define MF0(D) "v_mfma_f32_32x32x16_bf16 " D ", %6, %7, 0\n"
#define MF1(D) "v_mfma_f32_32x32x16_bf16 " D ", %6, %7, " D "\n"
#define AD(a) "v_add_f32 %" #a ", |%8|, %" #a "\n"
#define AD16 AD(2) AD(3) AD(4) AD(5) AD(2) AD(3) AD(4) AD(5) AD(2) AD(3) AD(4) AD(5) AD(2) AD(3) AD(4) AD(5)
#define AD8 AD(2) AD(3) AD(4) AD(5) AD(2) AD(3) AD(4) AD(5)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b)
{
    f32x16 d0, d1;
    for (int i = 0; i < 16; ++i) { d0[i] = 0; d1[i] = 0; }
    f32x4 av = {a + threadIdx.x, a, b, a}, bv = {b + threadIdx.x, b, a, b};
    float e = a, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(MF0("%0") MF1("%0") MF0("%1") MF1("%1") MF0("%0") MF1("%0") MF0("%1") MF1("%1")
            : "+v"(d0), "+v"(d1), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv), "v"(e));
        if (MODE == 1) asm volatile(MF0("%0") MF1("%0") AD16 MF0("%1") MF1("%1") AD16 MF0("%0") MF1("%0") AD16 MF0("%1") MF1("%1") AD16
            : "+v"(d0), "+v"(d1), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv), "v"(e));
        if (MODE == 2) asm volatile(MF0("%0") AD8 MF1("%0") AD8 MF0("%1") AD8 MF1("%1") AD8 MF0("%0") AD8 MF1("%0") AD8 MF0("%1") AD8 MF1("%1") AD8
            : "+v"(d0), "+v"(d1), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv), "v"(e));
        if (MODE == 3) asm volatile(MF0("%0") MF1("%0") MF0("%1") MF1("%1") AD16 AD16 MF0("%0") MF1("%0") MF0("%1") MF1("%1") AD16 AD16
            : "+v"(d0), "+v"(d1), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv), "v"(e));
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0[0] + d1[1] + c0 + c1 + c2 + c3;
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
    const int iters = 10000;
    for (int wps = 1; wps <= 4; ++wps) {
        int blocks = 256 * wps;
        // cycles per 256 residuals per SIMD: each trip = 4 pairs = 4096 residuals
        auto cyc = [&](double ms) { return ms * 1e-3 * 2.4e9 / ((double)iters * 16 * wps); };
        printf("wps %d 32x32x16 pairs, cyc per 256 residuals/SIMD: alone %.1f | pair+16 adds %.1f | (mfma,8 adds)x2 %.1f | 2 pairs+32 adds %.1f\n", wps,
               cyc(run<0>(out, blocks, iters)), cyc(run<1>(out, blocks, iters)), cyc(run<2>(out, blocks, iters)), cyc(run<3>(out, blocks, iters)));
    }
    return 0;
}
