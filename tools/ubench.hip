// ubench.hip -- issue-rate microbenchmarks behind the residual kernel's design:
// does v_mfma_f32_16x16x4_f32 overlap with VALU adds on the same SIMD, and what
// does a pure-VALU K=4 residual stream sustain?  Build: see tools/ubench.sh.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// A: MFMA only, 4 independent destinations per loop trip (16 MFMAs per trip)
__global__ __launch_bounds__(256) void k_mfma_only(float* out, int iters, float a, float b)
{
    f32x4 d0, d1, d2, d3;
    float av = a + threadIdx.x, bv = b + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "v_mfma_f32_16x16x4_f32 %0, %4, %5, 0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, 0\n"
            "v_mfma_f32_16x16x4_f32 %2, %4, %5, 0\n v_mfma_f32_16x16x4_f32 %3, %4, %5, 0\n"
            "v_mfma_f32_16x16x4_f32 %0, %4, %5, 0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, 0\n"
            "v_mfma_f32_16x16x4_f32 %2, %4, %5, 0\n v_mfma_f32_16x16x4_f32 %3, %4, %5, 0\n"
            "v_mfma_f32_16x16x4_f32 %0, %4, %5, 0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, 0\n"
            "v_mfma_f32_16x16x4_f32 %2, %4, %5, 0\n v_mfma_f32_16x16x4_f32 %3, %4, %5, 0\n"
            "v_mfma_f32_16x16x4_f32 %0, %4, %5, 0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, 0\n"
            "v_mfma_f32_16x16x4_f32 %2, %4, %5, 0\n v_mfma_f32_16x16x4_f32 %3, %4, %5, 0\n"
            : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(av), "v"(bv));
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0[0] + d1[1] + d2[2] + d3[3];
}

// B: MFMA + NADD independent v_add_f32 (abs modifier) per MFMA
template <int NADD>
__global__ __launch_bounds__(256) void k_mfma_adds(float* out, int iters, float a, float b)
{
    f32x4 d0, d1, d2, d3;
    float av = a + threadIdx.x, bv = b + threadIdx.x;
    float c0 = 0, c1 = 0, c2 = 0, c3 = 0, e0 = a, e1 = b, e2 = a, e3 = b;
    for (int i = 0; i < iters; ++i) {
#define ADDS4 "v_add_f32 %6, |%10|, %6\n v_add_f32 %7, |%11|, %7\n v_add_f32 %8, |%12|, %8\n v_add_f32 %9, |%13|, %9\n"
#define ADDS2 "v_add_f32 %6, |%10|, %6\n v_add_f32 %7, |%11|, %7\n"
#define ADDS1 "v_add_f32 %6, |%10|, %6\n"
#define ADDS6 ADDS4 ADDS2
#define ADDS8 ADDS4 ADDS4
#define ADDS(N) ADDS##N
#define STEP(D, N) "v_mfma_f32_16x16x4_f32 " D ", %4, %5, 0\n" N
        if (NADD == 1) { asm volatile(STEP("%0", ADDS1) STEP("%1", ADDS1) STEP("%2", ADDS1) STEP("%3", ADDS1) STEP("%0", ADDS1) STEP("%1", ADDS1) STEP("%2", ADDS1) STEP("%3", ADDS1)
            : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) , "+v"(av), "+v"(bv), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(e0), "v"(e1), "v"(e2), "v"(e3)); }
        if (NADD == 2) { asm volatile(STEP("%0", ADDS2) STEP("%1", ADDS2) STEP("%2", ADDS2) STEP("%3", ADDS2) STEP("%0", ADDS2) STEP("%1", ADDS2) STEP("%2", ADDS2) STEP("%3", ADDS2)
            : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) , "+v"(av), "+v"(bv), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(e0), "v"(e1), "v"(e2), "v"(e3)); }
        if (NADD == 4) { asm volatile(STEP("%0", ADDS4) STEP("%1", ADDS4) STEP("%2", ADDS4) STEP("%3", ADDS4) STEP("%0", ADDS4) STEP("%1", ADDS4) STEP("%2", ADDS4) STEP("%3", ADDS4)
            : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) , "+v"(av), "+v"(bv), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(e0), "v"(e1), "v"(e2), "v"(e3)); }
        if (NADD == 6) { asm volatile(STEP("%0", ADDS6) STEP("%1", ADDS6) STEP("%2", ADDS6) STEP("%3", ADDS6) STEP("%0", ADDS6) STEP("%1", ADDS6) STEP("%2", ADDS6) STEP("%3", ADDS6)
            : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) , "+v"(av), "+v"(bv), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(e0), "v"(e1), "v"(e2), "v"(e3)); }
        if (NADD == 8) { asm volatile(STEP("%0", ADDS8) STEP("%1", ADDS8) STEP("%2", ADDS8) STEP("%3", ADDS8) STEP("%0", ADDS8) STEP("%1", ADDS8) STEP("%2", ADDS8) STEP("%3", ADDS8)
            : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) , "+v"(av), "+v"(bv), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(e0), "v"(e1), "v"(e2), "v"(e3)); }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0[0] + d1[1] + d2[2] + d3[3] + c0 + c1 + c2 + c3;
}

// C: adds only (8 per trip-step, same count as B<4> without the MFMAs)
__global__ __launch_bounds__(256) void k_adds_only(float* out, int iters, float a, float b)
{
    float c0 = 0, c1 = 0, c2 = 0, c3 = 0, e0 = a + threadIdx.x, e1 = b, e2 = a, e3 = b;
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "v_add_f32 %0, |%4|, %0\n v_add_f32 %1, |%5|, %1\n v_add_f32 %2, |%6|, %2\n v_add_f32 %3, |%7|, %3\n"
            "v_add_f32 %0, |%4|, %0\n v_add_f32 %1, |%5|, %1\n v_add_f32 %2, |%6|, %2\n v_add_f32 %3, |%7|, %3\n"
            "v_add_f32 %0, |%4|, %0\n v_add_f32 %1, |%5|, %1\n v_add_f32 %2, |%6|, %2\n v_add_f32 %3, |%7|, %3\n"
            "v_add_f32 %0, |%4|, %0\n v_add_f32 %1, |%5|, %1\n v_add_f32 %2, |%6|, %2\n v_add_f32 %3, |%7|, %3\n"
            "v_add_f32 %0, |%4|, %0\n v_add_f32 %1, |%5|, %1\n v_add_f32 %2, |%6|, %2\n v_add_f32 %3, |%7|, %3\n"
            "v_add_f32 %0, |%4|, %0\n v_add_f32 %1, |%5|, %1\n v_add_f32 %2, |%6|, %2\n v_add_f32 %3, |%7|, %3\n"
            "v_add_f32 %0, |%4|, %0\n v_add_f32 %1, |%5|, %1\n v_add_f32 %2, |%6|, %2\n v_add_f32 %3, |%7|, %3\n"
            "v_add_f32 %0, |%4|, %0\n v_add_f32 %1, |%5|, %1\n v_add_f32 %2, |%6|, %2\n v_add_f32 %3, |%7|, %3\n"
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(e0), "v"(e1), "v"(e2), "v"(e3));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
}

// D: pure-VALU residual stream: 8 points per lane in registers, candidates' w in
// SGPRs; per (candidate, point): 3 v_fma + 1 v_add|.|; one v_mov per candidate.
__global__ __launch_bounds__(256) void k_valu_residual(const float4* __restrict__ w, float* out,
                                                        int n_cand, float seed)
{
    float x[8], y[8], z[8], acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) { x[p] = seed + threadIdx.x + p; y[p] = seed * 2 + p; z[p] = seed * 3 - p; acc[p] = 0; }
    for (int c = 0; c < n_cand; c += 4) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const float4 wc = w[c + cc];   // uniform address -> scalar load
            float t[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                t[p] = __builtin_fmaf(wc.x, x[p], wc.w);
                t[p] = __builtin_fmaf(wc.y, y[p], t[p]);
                t[p] = __builtin_fmaf(wc.z, z[p], t[p]);
                acc[(p + cc) & 7] += __builtin_fabsf(t[p]);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) s += acc[p];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// E: packed variant: v_pk_fma_f32 on point pairs
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_valu_residual_pk(const float4* __restrict__ w, float* out,
                                                           int n_cand, float seed)
{
    f32x2 x[4], y[4], z[4], acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        x[p] = f32x2{seed + threadIdx.x + p, seed - p}; y[p] = f32x2{seed * 2 + p, seed * 5 + p};
        z[p] = f32x2{seed * 3 - p, seed * 7 - p}; acc[p] = f32x2{0, 0};
    }
    for (int c = 0; c < n_cand; c += 4) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const float4 wc = w[c + cc];
            const f32x2 w0 = {wc.x, wc.x}, w1 = {wc.y, wc.y}, w2 = {wc.z, wc.z}, w3 = {wc.w, wc.w};
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                f32x2 t = __builtin_elementwise_fma(w0, x[p], w3);
                t = __builtin_elementwise_fma(w1, y[p], t);
                t = __builtin_elementwise_fma(w2, z[p], t);
                acc[p] += __builtin_elementwise_abs(t);
            }
        }
    }
    f32x2 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1];
}

template <typename F>
double time_ms(F launch, int reps = 5)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    float* out; CHECK(hipMalloc(&out, 1 << 24));
    const int ncand = 4096;
    std::vector<float4> hw(ncand);
    for (int i = 0; i < ncand; ++i) hw[i] = make_float4(0.1f + i * 1e-5f, 0.2f, 0.9f, -0.3f);
    float4* w; CHECK(hipMalloc(&w, ncand * sizeof(float4)));
    CHECK(hipMemcpy(w, hw.data(), ncand * sizeof(float4), hipMemcpyHostToDevice));
    const int iters = 20000;
    for (int wps = 1; wps <= 4; ++wps) {          // waves per SIMD (blocks of 4 waves per CU)
        const int blocks = cus * wps;
        const double simds = cus * 4.0;
        auto cyc_per = [&](double ms, double n_per_wave) { return ms * 1e-3 * 2.4e9 / (n_per_wave * wps); };
        double a = time_ms([&] { hipLaunchKernelGGL(k_mfma_only, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
        printf("wps %d  mfma_only        %.3f ms  %.1f cyc/MFMA/SIMD (at 2.4GHz)\n", wps, a, cyc_per(a, iters * 16.0));
        double b1 = time_ms([&] { hipLaunchKernelGGL(k_mfma_adds<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
        double b2 = time_ms([&] { hipLaunchKernelGGL(k_mfma_adds<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
        double b4 = time_ms([&] { hipLaunchKernelGGL(k_mfma_adds<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
        double b6 = time_ms([&] { hipLaunchKernelGGL(k_mfma_adds<6>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
        double b8 = time_ms([&] { hipLaunchKernelGGL(k_mfma_adds<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
        printf("wps %d  mfma+1/2/4/6/8 adds  %.1f %.1f %.1f %.1f %.1f cyc/MFMA/SIMD\n", wps, cyc_per(b1, iters * 8.0), cyc_per(b2, iters * 8.0),
               cyc_per(b4, iters * 8.0), cyc_per(b6, iters * 8.0), cyc_per(b8, iters * 8.0));
        double c = time_ms([&] { hipLaunchKernelGGL(k_adds_only, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
        printf("wps %d  adds_only        %.3f ms  %.2f cyc/add/SIMD\n", wps, c, cyc_per(c, iters * 32.0));
        double d = time_ms([&] { hipLaunchKernelGGL(k_valu_residual, dim3(blocks), dim3(256), 0, 0, w, out, ncand, 0.5f); });
        double res = (double)blocks * 256 * 8 * ncand;
        printf("wps %d  valu_residual    %.3f ms  %.3e res/s  frac %.3f\n", wps, d, res / d * 1e3, res * 8 / d * 1e3 / 157.3e12);
        double e = time_ms([&] { hipLaunchKernelGGL(k_valu_residual_pk, dim3(blocks), dim3(256), 0, 0, w, out, ncand, 0.5f); });
        printf("wps %d  valu_residual_pk %.3f ms  %.3e res/s  frac %.3f\n", wps, e, res / e * 1e3, res * 8 / e * 1e3 / 157.3e12);
        (void)simds;
    }
    return 0;
}
