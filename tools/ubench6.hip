// ubench6.hip -- issue rate of the FP64 / 64-bit integer VALU instructions the exact path uses.
// Each mode runs 8 independent dependency chains per wave; cycles per wave instruction per SIMD
// are printed for 1..4 waves per SIMD (a 2.4 GHz clock is assumed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define ADD64(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define MUL64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define FMA64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define ADDU64(i) "v_lshl_add_u64 %" #i ", %" #i ", 0, %8\n"
#define ADD32(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define CVTDF(i) "v_cvt_f32_f64 %" #i ", %8\n"
#define CVTFD(i) "v_cvt_f64_f32 %" #i ", %9\n"
#define CMP64(i) "v_cmp_lt_f64 vcc, %" #i ", %8\n"
#define MAX64(i) "v_max_f64 %" #i ", %" #i ", %8\n"
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a, double b)
{
    double c0 = a + threadIdx.x, c1 = a, c2 = b, c3 = a + 1, c4 = a + 2, c5 = a + 3, c6 = a + 4, c7 = a + 5;
    double e = b, f = a;
    float g = (float)a;
    for (int i = 0; i < iters; ++i) {
#define OPERANDS : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(e), "v"(f)
        if (MODE == 0) asm volatile(R8(ADD64) R8(ADD64) R8(ADD64) R8(ADD64) OPERANDS);
        if (MODE == 1) asm volatile(R8(MUL64) R8(MUL64) R8(MUL64) R8(MUL64) OPERANDS);
        if (MODE == 2) asm volatile(R8(FMA64) R8(FMA64) R8(FMA64) R8(FMA64) OPERANDS);
        if (MODE == 3) asm volatile(R8(ADDU64) R8(ADDU64) R8(ADDU64) R8(ADDU64) OPERANDS);
        if (MODE == 4) asm volatile(R8(MAX64) R8(MAX64) R8(MAX64) R8(MAX64) OPERANDS);
        if (MODE == 5) asm volatile(R8(CMP64) R8(CMP64) R8(CMP64) R8(CMP64) OPERANDS : "vcc");
        if (MODE == 6) {
            float h0, h1, h2, h3, h4, h5, h6, h7;
            asm volatile(R8(CVTDF) R8(CVTDF) R8(CVTDF) R8(CVTDF)
                         : "=v"(h0), "=v"(h1), "=v"(h2), "=v"(h3), "=v"(h4), "=v"(h5), "=v"(h6), "=v"(h7) : "v"(e), "v"(g));
            g += h0 + h7;
        }
        if (MODE == 7) asm volatile(R8(CVTFD) R8(CVTFD) R8(CVTFD) R8(CVTFD)
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(e), "v"(g));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + g;
}
template <int N> double run(double* out, int blocks, int iters)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < 3; ++r) { CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    return best;
}
int main()
{
    double* out; CHECK(hipMalloc(&out, 1 << 24));
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = 256 * wps;
        auto cyc = [&](double ms) { return ms * 1e-3 * 2.4e9 / ((double)iters * 32 * wps); };
        printf("wps %d cycles/wave-instr/SIMD: add_f64 %.2f mul_f64 %.2f fma_f64 %.2f lshl_add_u64 %.2f max_f64 %.2f cmp_f64 %.2f cvt_f32_f64 %.2f cvt_f64_f32 %.2f\n", wps,
               cyc(run<0>(out, blocks, iters)), cyc(run<1>(out, blocks, iters)), cyc(run<2>(out, blocks, iters)), cyc(run<3>(out, blocks, iters)),
               cyc(run<4>(out, blocks, iters)), cyc(run<5>(out, blocks, iters)), cyc(run<6>(out, blocks, iters)), cyc(run<7>(out, blocks, iters)));
    }
    return 0;
}
