// ubench3.hip -- can the K=4 FP32 contraction run on the bf16 matrix pipe at FP32 accuracy?
// Each f32 operand is split exactly into three bf16 pieces (8+8+8 significand bits); the eight
// significant cross products per coordinate occupy the 32 K-slots of v_mfma_f32_16x16x32_bf16:
//   slots of lane group g<3 (coordinate g): A = [wh wh wm wh wm wl wm wl], B = [ph pm ph pl pm ph pl pm]
//   lane group 3 (constant term, p = 1):    A = [w3h 0 w3m 0 0 w3l 0 0],  B from the same split of 1.0
// Measures cycles per MFMA with the |d| accumulate adds, and the error against FP64.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ inline void split3(float v, unsigned& h, unsigned& m, unsigned& l)
{
    h = __float_as_uint(v) & 0xffff0000u;
    const float r1 = v - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(m);
    l = __float_as_uint(r2) & 0xffff0000u;
}

// packs two bf16 (given as f32 bit patterns with zero low halves): lo16 = a, hi16 = b
__device__ inline unsigned pk(unsigned a, unsigned b) { return (a >> 16) | b; }

__device__ inline bf16x8 make_b(float v)
{
    unsigned h, m, l;
    split3(v, h, m, l);
    u32x4 r = {pk(h, m), pk(h, l), pk(m, h), pk(l, m)};
    return __builtin_bit_cast(bf16x8, r);
}

__device__ inline bf16x8 make_a(float w, bool constant_group)
{
    unsigned h, m, l;
    split3(w, h, m, l);
    u32x4 r;
    if (constant_group) r = u32x4{pk(h, 0), pk(m, 0), pk(0, l), 0u};
    else r = u32x4{pk(h, h), pk(m, h), pk(m, l), pk(m, l)};
    return __builtin_bit_cast(bf16x8, r);
}

constexpr int NA = 8;

// W: [NA tiles][64] f32 (k*16+cand), groups: [n_groups][64 lanes] float4 as in the product layout.
template <int MODE>   // 0 = bf16x3 MFMA, 1 = f32 MFMA (reference structure)
__global__ __launch_bounds__(256) void k_resid(const float* __restrict__ W, const f32x4* __restrict__ groups,
                                               int n_groups, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const f32x4 zero = {0, 0, 0, 0};
    float wv[NA];
    bf16x8 A[NA];
    for (int a = 0; a < NA; ++a) {
        wv[a] = W[((blockIdx.x * 4 + wave) * NA + a) * 64 + lane];
        A[a] = make_a(wv[a], lane >= 48);
    }
    float acc[NA][4];
    for (int a = 0; a < NA; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
    const f32x4* gp = groups + lane;
    f32x4 b = gp[0];
    if (MODE == 0) {
        // two-deep software pipeline: adds of MFMA k are issued behind MFMAs k+1 and k+2; the
        // bf16 split of the next point tile is spread between the MFMAs of the current one
        bf16x8 Bc = make_b(b[0]);
        f32x4 c0 = zero, c1 = zero;
        for (int g = 0; g < n_groups; ++g) {
            f32x4 bn = b;
            if (g + 1 < n_groups) bn = gp[64];
            gp += 64;
            f32x4 d[4 * NA];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bf16x8 Bn = make_b(u < 3 ? b[u + 1] : bn[0]);
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const int idx = u * NA + a;
                    d[idx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[a], Bc, zero, 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    const f32x4 done = idx == 0 ? c0 : (idx == 1 ? c1 : d[idx >= 2 ? idx - 2 : 0]);
                    const int aa = (idx + 4 * NA - 2) % NA;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[aa][r] += __builtin_fabsf(done[r]);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                }
                Bc = Bn;
            }
            c0 = d[4 * NA - 2]; c1 = d[4 * NA - 1];
            b = bn;
        }
        for (int r = 0; r < 4; ++r) { acc[NA - 2][r] += __builtin_fabsf(c0[r]); acc[NA - 1][r] += __builtin_fabsf(c1[r]); }
    } else {
        for (int g = 0; g < n_groups; ++g) {
            f32x4 bn = b;
            if (g + 1 < n_groups) bn = gp[64];
            gp += 64;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[a], b[u], zero, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[a][r] += __builtin_fabsf(d[r]);
                }
            }
            b = bn;
        }
    }
    // per-candidate sums: reduce over the 16 lanes of a row
    for (int a = 0; a < NA; ++a)
        for (int r = 0; r < 4; ++r) {
            float v = acc[a][r];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if ((lane & 15) == 0) out[(((blockIdx.x * 4 + wave) * NA + a) * 16) + (lane >> 4) * 4 + r] = v;
        }
}

// single-residual probe: D for one tile pair, to compare individual residuals with FP64
template <int MODE>
__global__ void k_single(const float* __restrict__ W, const f32x4* __restrict__ groups, float* __restrict__ out)
{
    const int lane = threadIdx.x;
    const f32x4 zero = {0, 0, 0, 0};
    const float wv = W[lane];
    const f32x4 b = groups[lane];
    f32x4 d;
    if (MODE == 0) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(make_a(wv, lane >= 48), make_b(b[0]), zero, 0, 0, 0);
    else d = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, b[0], zero, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = d[r];   // [cand][point]
}

int main()
{
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1, 1);
    std::normal_distribution<double> N(0, 1);
    const int cus = 256;
    const int n_groups = 2048;                    // 131072 points
    const int blocks = cus * 3;                   // 3 blocks of 4 waves per CU
    const int n_tiles = blocks * 4 * NA;
    // candidates: unit-ish normal + small offset, like folded plane vectors
    std::vector<float> W((size_t)n_tiles * 64);
    std::vector<double> Wd((size_t)n_tiles * 16 * 4);
    for (int t = 0; t < n_tiles; ++t)
        for (int c = 0; c < 16; ++c) {
            double w[4] = {0.05 * U(rng), 0.05 * U(rng), 1.0, 0.002 * U(rng)};
            double nn = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
            for (int k = 0; k < 3; ++k) w[k] /= nn;
            for (int k = 0; k < 4; ++k) { W[(size_t)t * 64 + k * 16 + c] = (float)w[k]; Wd[((size_t)t * 16 + c) * 4 + k] = (double)(float)w[k]; }
        }
    // points near the plane z = 0 (centred cloud): x,y in +-0.15, z noise 0.5 mm + tilt
    std::vector<float> G((size_t)n_groups * 256);
    std::vector<double> P((size_t)n_groups * 64 * 3);
    for (int g = 0; g < n_groups; ++g)
        for (int u = 0; u < 4; ++u)
            for (int col = 0; col < 16; ++col) {
                float x = (float)(0.15 * U(rng)), y = (float)(0.15 * U(rng)), z = (float)(0.03 * x + 0.0005 * N(rng));
                float* gp = &G[(size_t)g * 256];
                gp[(0 * 16 + col) * 4 + u] = x; gp[(1 * 16 + col) * 4 + u] = y; gp[(2 * 16 + col) * 4 + u] = z; gp[(3 * 16 + col) * 4 + u] = 1.f;
                double* pp = &P[((size_t)g * 64 + u * 16 + col) * 3];
                pp[0] = x; pp[1] = y; pp[2] = z;
            }
    float *dW, *dout; f32x4* dG;
    CHECK(hipMalloc(&dW, W.size() * 4)); CHECK(hipMalloc(&dG, G.size() * 4)); CHECK(hipMalloc(&dout, (size_t)n_tiles * 16 * 4));
    CHECK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dG, G.data(), G.size() * 4, hipMemcpyHostToDevice));

    // --- single residual accuracy -------------------------------------------------------
    float* ds; CHECK(hipMalloc(&ds, 256 * 4));
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(k_single<0>, dim3(1), dim3(64), 0, 0, dW, dG, ds);
        else hipLaunchKernelGGL(k_single<1>, dim3(1), dim3(64), 0, 0, dW, dG, ds);
        std::vector<float> hs(256); CHECK(hipMemcpy(hs.data(), ds, 256 * 4, hipMemcpyDeviceToHost));
        double maxabs = 0, maxres = 0;
        for (int c = 0; c < 16; ++c) for (int p = 0; p < 16; ++p) {
            const double* w = &Wd[(size_t)c * 4]; const double* q = &P[(size_t)p * 3];
            double ref = w[0] * q[0] + w[1] * q[1] + w[2] * q[2] + w[3];
            maxabs = std::max(maxabs, std::fabs(hs[c * 16 + p] - ref)); maxres = std::max(maxres, std::fabs(ref));
        }
        printf("%s single-tile: max |d - ref| = %.3e (max |ref| %.3e)\n", mode == 0 ? "bf16x3" : "f32   ", maxabs, maxres);
    }

    // --- sums accuracy + timing ---------------------------------------------------------
    std::vector<double> ref(64 * 16, 0.0);      // first 64 tiles only on the host
    for (int t = 0; t < 64; ++t) for (int c = 0; c < 16; ++c) {
        const double* w = &Wd[((size_t)t * 16 + c) * 4]; double s = 0;
        for (size_t p = 0; p < (size_t)n_groups * 64; ++p) s += std::fabs(w[0] * P[p * 3] + w[1] * P[p * 3 + 1] + w[2] * P[p * 3 + 2] + w[3]);
        ref[t * 16 + c] = s;
    }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        double best = 1e30;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_resid<0>, dim3(blocks), dim3(256), 0, 0, dW, dG, n_groups, dout);
            else hipLaunchKernelGGL(k_resid<1>, dim3(blocks), dim3(256), 0, 0, dW, dG, n_groups, dout);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
        }
        std::vector<float> ho((size_t)64 * 16); CHECK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
        double maxrel = 0;
        for (size_t i = 0; i < ho.size(); ++i) maxrel = std::max(maxrel, std::fabs(ho[i] - ref[i]) / ref[i]);
        double res = (double)n_tiles * 16 * n_groups * 64;
        double mfma_per_simd = (double)n_groups * 4 * NA * 3;     // per SIMD: 3 waves
        printf("%s: %.3f ms  %.3e res/s  frac(157.3TF) %.3f  cyc/MFMA/SIMD %.1f  max rel err of sums %.2e (fp32 accumulate over %d adds)\n",
               mode == 0 ? "bf16x3" : "f32   ", best, res / best * 1e3, res * 8 / best * 1e3 / 157.3e12, best * 1e-3 * 2.4e9 / mfma_per_simd, maxrel, n_groups * 4);
    }
    return 0;
}
