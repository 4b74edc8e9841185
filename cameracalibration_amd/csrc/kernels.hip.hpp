// kernels.hip.hpp -- gfx950 kernels of the calibration hot path.
//
// Two arithmetic families:
//   * "fast": FP32, the point-to-plane residual folded into a 4-vector per
//     (candidate, cloud) and contracted with homogeneous points on the matrix
//     cores: v_mfma_f32_16x16x4_f32 is exactly a K=4 contraction of a
//     16-candidate tile with a 16-point tile, bit-identical to an fmaf chain.
//     |.| and the running sum are VALU adds on the MFMA result.
//   * "exact": the reference's own arithmetic (FP64 SE3 apply in the SSE2
//     summation order, point rounded to float, FP64 distance with a true
//     division, distances of a cloud added one by one in point order), used for
//     single poses, finite-difference stencils, the error report and for
//     re-scoring near-minimum grid candidates.  Its results are bit-identical
//     to the CPU evaluation of the same lines.
//
// Included once by ccal_api.hip; everything here is internal.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pose_math.hpp"

#pragma clang fp contract(off)

namespace ccal {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// Layouts
// ---------------------------------------------------------------------------
// Fast path points: "groups" of 64 points = 4 MFMA tiles of 16 points.  A group
// is 64 lanes x float4 (1 KiB), so one global_load_dwordx4 per lane fetches the
// four B operands of a wave: component u of lane l is element l of tile u,
// element l = k*16 + col holds coordinate k (x,y,z,one) of point u*16 + col.
// Points are stored relative to their cloud's centre; padding points are all
// zero including `one`, so they contribute |0| to every sum.
//
// Fast path candidates: W[cloud][ctile][k*16 + cand] (64 floats per tile), one
// coalesced dword per lane is the A operand of a 16-candidate tile.
constexpr int kGroupPoints = 64;
constexpr int kTilePoints = 16;
constexpr int kFlushGroups = 32;  // FP32 partial sums are promoted every 128 adds

// ---------------------------------------------------------------------------
// Candidate sources for the W generator
// ---------------------------------------------------------------------------
// Grid: per axis a table of translations (axes 0..2) or of sin/cos pairs
// (axes 3..5), built on the host in FP64 with the reference's formulas.
struct GridTables {
    const double* t[3];    // t[k][i] = base[k] + value/1000.0
    const double* sn[3];   // sn[k][i] = sin(base[3+k] + value*3.141592/180)
    const double* cs[3];
    int count[6];
    int64_t first;         // linear index of chunk candidate 0
};

// List: explicit records of 9 doubles: t[3], sin[3], cos[3].
struct PoseList {
    const double* rec;
};

struct FoldConsts {
    double n[3];
    double d;
    double inv_norm;
};

__device__ inline void fetch_pose(const GridTables& g, int64_t c, double t[3], double s[3],
                                  double cs[3])
{
    int64_t idx = g.first + c;
    int i5 = (int)(idx % g.count[5]); idx /= g.count[5];
    int i4 = (int)(idx % g.count[4]); idx /= g.count[4];
    int i3 = (int)(idx % g.count[3]); idx /= g.count[3];
    int i2 = (int)(idx % g.count[2]); idx /= g.count[2];
    int i1 = (int)(idx % g.count[1]); idx /= g.count[1];
    int i0 = (int)idx;
    t[0] = g.t[0][i0]; t[1] = g.t[1][i1]; t[2] = g.t[2][i2];
    s[0] = g.sn[0][i3]; s[1] = g.sn[1][i4]; s[2] = g.sn[2][i5];
    cs[0] = g.cs[0][i3]; cs[1] = g.cs[1][i4]; cs[2] = g.cs[2][i5];
}

__device__ inline void fetch_pose(const PoseList& p, int64_t c, double t[3], double s[3],
                                  double cs[3])
{
    const double* r = p.rec + c * 9;
    for (int k = 0; k < 3; ++k) { t[k] = r[k]; s[k] = r[3 + k]; cs[k] = r[6 + k]; }
}

// One thread per candidate; loops over clouds.  Candidates >= n_valid (padding
// up to the launch tile) get w = 0.
template <typename Source>
__global__ __launch_bounds__(256) void wgen_kernel(Source src, int64_t n_valid, int n_ctiles,
                                                    int n_clouds,
                                                    const double* __restrict__ base_T_flange,
                                                    const double* __restrict__ centres,
                                                    FoldConsts fc, float* __restrict__ W)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)n_ctiles * kTilePoints) return;
    const bool valid = c < n_valid;
    double R[9], t[3] = {0, 0, 0};
    if (valid) {
        double s[3], cs[3];
        fetch_pose(src, c, t, s, cs);
        rotation_zyx(s, cs, R);
    }
    const int64_t tile = c >> 4;
    const int col = (int)(c & 15);
    for (int j = 0; j < n_clouds; ++j) {
        double w[4] = {0, 0, 0, 0};
        if (valid) {
            double M[12];
            compose_rows(base_T_flange + 16 * j, R, t, M);
            fold_plane(M, fc.n, fc.d, fc.inv_norm, centres + 3 * j, w);
        }
        float* dst = W + ((int64_t)j * n_ctiles + tile) * 64 + col;
        dst[0] = (float)w[0];
        dst[16] = (float)w[1];
        dst[32] = (float)w[2];
        dst[48] = (float)w[3];
    }
}

// ---------------------------------------------------------------------------
// Fast residual kernel
// ---------------------------------------------------------------------------
// grid = (candidate blocks, point splits); block = 4 waves; wave w of block bx
// owns candidate tiles [(bx*4+w)*NA, +NA).  All four waves stream the same
// point groups (their loads coalesce in L1/L2), split `by` of every cloud.
// Output: partial[by][candidate] = sum_j scale_j * sum_{i in split} |w.p|.
template <int NA>
__global__ __launch_bounds__(256, 3) void residual_mfma_kernel(
    const float* __restrict__ W, const f32x4* __restrict__ groups,
    const int* __restrict__ group_off, const double* __restrict__ scale,
    double* __restrict__ partial, int n_clouds, int n_ctiles, int n_splits)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ctile0 = (blockIdx.x * 4 + wave) * NA;
    const int split = blockIdx.y;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    double accD[NA][4];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) accD[a][r] = 0.0;

    for (int j = 0; j < n_clouds; ++j) {
        const int g0 = group_off[j];
        const int G = group_off[j + 1] - g0;
        const int gb = g0 + (int)(((int64_t)G * split) / n_splits);
        const int ge = g0 + (int)(((int64_t)G * (split + 1)) / n_splits);
        if (gb >= ge) continue;
        float A[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a)
            A[a] = W[((int64_t)j * n_ctiles + ctile0 + a) * 64 + lane];
        const double sc = scale[j];

        float acc[NA][4];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;

        const f32x4* gp = groups + (int64_t)gb * 64 + lane;
        f32x4 b = gp[0];
        // Retire the A and first-B loads here, so that inside the loop the only
        // outstanding load is the next group's prefetch (vmcnt(0), others max).
        __builtin_amdgcn_s_waitcnt(0x0F70);
        int since = 0;
        // The f32 MFMA and the VALU adds do not overlap on a SIMD, and every
        // switch between them costs a few cycles, so they are issued in bursts:
        // kBurst MFMAs back to back, then the 4*kBurst |d| adds of the PREVIOUS
        // burst (whose results have long landed, so no wait states are needed).
        // The last burst of a group is carried into the next loop trip; the
        // carry starts as zero, and |0| adds nothing.
        constexpr int kBurst = 4;
        static_assert(NA % kBurst == 0, "candidate tiles per wave must be a multiple of the burst");
        f32x4 carry[kBurst];
#pragma unroll
        for (int q = 0; q < kBurst; ++q) carry[q] = zero;
        for (int g = gb; g < ge; ++g) {
            // register double buffer: fetch the next group before using this one
            f32x4 bn = b;
            if (g + 1 < ge) bn = gp[64];
            gp += 64;
            f32x4 d[4 * NA];
#pragma unroll
            for (int i0 = 0; i0 < 4 * NA; i0 += kBurst) {
#pragma unroll
                for (int q = 0; q < kBurst; ++q)
                    d[i0 + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(i0 + q) % NA], b[(i0 + q) / NA], zero, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, kBurst, 0);
#pragma unroll
                for (int q = 0; q < kBurst; ++q) {
                    const f32x4 done = i0 == 0 ? carry[q] : d[i0 == 0 ? 0 : i0 - kBurst + q];
                    const int a = (i0 + 4 * NA - kBurst + q) % NA;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[a][r] += __builtin_fabsf(done[r]);
                }
                __builtin_amdgcn_sched_group_barrier(0x002, 4 * kBurst, 0);
            }
#pragma unroll
            for (int q = 0; q < kBurst; ++q) carry[q] = d[4 * NA - kBurst + q];
            b = bn;
            if (++since == kFlushGroups) {
                since = 0;
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        accD[a][r] += (double)acc[a][r] * sc;
                        acc[a][r] = 0.f;
                    }
            }
        }
#pragma unroll
        for (int q = 0; q < kBurst; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[NA - kBurst + q][r] += __builtin_fabsf(carry[q][r]);
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) accD[a][r] += (double)acc[a][r] * sc;
    }

    // D layout: lane l, register r holds candidate row (l>>4)*4 + r and point
    // column l&15, so the 16 lanes of a DPP row share their four candidates.
    double* out = partial + (int64_t)split * n_ctiles * 16;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double v = accD[a][r];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            if ((lane & 15) == 0)
                out[(int64_t)(ctile0 + a) * 16 + (lane >> 4) * 4 + r] = v;
        }
}

// ---------------------------------------------------------------------------
// Fast residual kernel on the bf16 matrix pipe, FP32-equivalent operands
// ---------------------------------------------------------------------------
// The FP32-input MFMA above shares the SIMD with the VALU adds (they serialise),
// so it tops out near 3/4 of the FP32 peak.  v_mfma_f32_16x16x32_bf16 runs on the
// dense matrix pipe at twice that rate, and its K = 32 is wide enough to carry an
// exact three-way bf16 split of both FP32 operands:
//     v = v_h + v_m + v_l   (8 + 8 + 8 significand bits, exact for any float)
// The eight significant cross products of a coordinate fill one lane group's 8
// K-slots (the dropped v_l*w_l term is below 2^-32 relative), products of bf16
// pairs are exact in FP32, and the pipe accumulates them in FP32:
//   lane group g < 3 (coordinate g):  A = [wh wh wm wh wm wl wm wl]
//                                     B = [ph pm ph pl pm ph pl pm]
//   lane group 3 (the constant, p=1): A = [w3h 0 w3m 0 0 w3l 0 0], B = same split of 1.0
// Measured error per residual 1.5e-9 (FP32 fmaf chain: 0.8e-9) on calib_1-like data.
// The splits are done in registers from the same FP32 layouts the FP32 kernel
// reads (9 VALU ops per 16-point tile and wave, shared by NA MFMAs).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ inline void split_bf16x3(float v, unsigned& h, unsigned& m, unsigned& l)
{
    h = __float_as_uint(v) & 0xffff0000u;             // truncation keeps every piece exact
    const float r1 = v - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(m);
    l = __float_as_uint(r2) & 0xffff0000u;
}

// two bf16 given as FP32 bit patterns with zero low halves: low 16 bits = a, high = b
__device__ inline unsigned pack_bf16(unsigned a, unsigned b) { return (a >> 16) | b; }

__device__ inline bf16x8 split_point_operand(float v)
{
    unsigned h, m, l;
    split_bf16x3(v, h, m, l);
    const u32x4 r = {pack_bf16(h, m), pack_bf16(h, l), pack_bf16(m, h), pack_bf16(l, m)};
    return __builtin_bit_cast(bf16x8, r);
}

__device__ inline bf16x8 split_candidate_operand(float w, bool constant_group)
{
    unsigned h, m, l;
    split_bf16x3(w, h, m, l);
    u32x4 r;
    if (constant_group) r = u32x4{pack_bf16(h, 0u), pack_bf16(m, 0u), pack_bf16(0u, l), 0u};
    else r = u32x4{pack_bf16(h, h), pack_bf16(m, h), pack_bf16(m, l), pack_bf16(m, l)};
    return __builtin_bit_cast(bf16x8, r);
}

// Promotes the FP32 lane accumulators of a wave into two FP64 totals per lane.
// acc[a][r] of lane l belongs to candidate (l>>4)*4 + r of tile a and point column
// l&15; the 16 lanes of a row are reduce-scattered (4 exchange steps), after which
// lane column c owns the row totals of pairs 2c and 2c+1, pair = a*4 + r.
template <int NA>
__device__ inline void promote_rows(float (&acc)[NA][4], double scale, double& tot0, double& tot1)
{
    static_assert(NA == 8, "the reduce-scatter below is written for 32 values per lane");
    const int col = threadIdx.x & 15;
    float v[32];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[a * 4 + r] = acc[a][r]; acc[a][r] = 0.f; }
    // step with partner col^8: lanes with bit 3 clear keep values 0..15, others 16..31
    float w16[16];
    {
        const bool hi = (col & 8) != 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float keep = hi ? v[16 + i] : v[i];
            const float give = hi ? v[i] : v[16 + i];
            w16[i] = keep + __shfl_xor(give, 8, 64);
        }
    }
    float w8[8];
    {
        const bool hi = (col & 4) != 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float keep = hi ? w16[8 + i] : w16[i];
            const float give = hi ? w16[i] : w16[8 + i];
            w8[i] = keep + __shfl_xor(give, 4, 64);
        }
    }
    float w4[4];
    {
        const bool hi = (col & 2) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float keep = hi ? w8[4 + i] : w8[i];
            const float give = hi ? w8[i] : w8[4 + i];
            w4[i] = keep + __shfl_xor(give, 2, 64);
        }
    }
    {
        const bool hi = (col & 1) != 0;
        const float k0 = hi ? w4[2] : w4[0], g0 = hi ? w4[0] : w4[2];
        const float k1 = hi ? w4[3] : w4[1], g1 = hi ? w4[1] : w4[3];
        tot0 += (double)(k0 + __shfl_xor(g0, 1, 64)) * scale;
        tot1 += (double)(k1 + __shfl_xor(g1, 1, 64)) * scale;
    }
}

// Pair index owned by lane column c after promote_rows: bit 3 of c selects the
// upper 16 values, bit 2 the upper 8 of those, bit 1 the upper 4, bit 0 the upper 2;
// the two totals are consecutive values of that quad-pair.
__device__ inline int promoted_pair(int col, int which)
{
    return ((col & 8) ? 16 : 0) + ((col & 4) ? 8 : 0) + ((col & 2) ? 4 : 0) + ((col & 1) ? 2 : 0) + which;
}

template <int NA>
__global__ __launch_bounds__(256, 3) void residual_bf16x3_kernel(
    const float* __restrict__ W, const f32x4* __restrict__ groups,
    const int* __restrict__ group_off, const double* __restrict__ scale,
    double* __restrict__ partial, int n_clouds, int n_ctiles, int n_splits)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ctile0 = (blockIdx.x * 4 + wave) * NA;
    const int split = blockIdx.y;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    double tot0 = 0.0, tot1 = 0.0;

    for (int j = 0; j < n_clouds; ++j) {
        const int g0 = group_off[j];
        const int G = group_off[j + 1] - g0;
        const int gb = g0 + (int)(((int64_t)G * split) / n_splits);
        const int ge = g0 + (int)(((int64_t)G * (split + 1)) / n_splits);
        if (gb >= ge) continue;
        bf16x8 A[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a)
            A[a] = split_candidate_operand(W[((int64_t)j * n_ctiles + ctile0 + a) * 64 + lane], lane >= 48);
        const double sc = scale[j];

        float acc[NA][4];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;

        const f32x4* gp = groups + (int64_t)gb * 64 + lane;
        f32x4 b = gp[0];
        __builtin_amdgcn_s_waitcnt(0x0F70);   // A and first-B loads retired before the loop
        int since = 0;
        // Bursts: kBurst MFMAs back to back, then the |d| adds of the previous
        // burst; the bf16 split of the next point tile rides along with the adds.
        constexpr int kBurst = 4;
        static_assert(NA % kBurst == 0, "candidate tiles per wave must be a multiple of the burst");
        bf16x8 Bc = split_point_operand(b[0]);
        f32x4 carry[kBurst];
#pragma unroll
        for (int q = 0; q < kBurst; ++q) carry[q] = zero;
        for (int g = gb; g < ge; ++g) {
            f32x4 bn = b;
            if (g + 1 < ge) bn = gp[64];
            gp += 64;
            f32x4 d[4 * NA];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bf16x8 Bn = split_point_operand(u < 3 ? b[u + 1] : bn[0]);
#pragma unroll
                for (int i0 = u * NA; i0 < (u + 1) * NA; i0 += kBurst) {
#pragma unroll
                    for (int q = 0; q < kBurst; ++q)
                        d[i0 + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[(i0 + q) % NA], Bc, zero, 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, kBurst, 0);
#pragma unroll
                    for (int q = 0; q < kBurst; ++q) {
                        const f32x4 done = i0 == 0 ? carry[q] : d[i0 == 0 ? 0 : i0 - kBurst + q];
                        const int a = (i0 + 4 * NA - kBurst + q) % NA;
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[a][r] += __builtin_fabsf(done[r]);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x002, 4 * kBurst + 5, 0);
                }
                Bc = Bn;
            }
#pragma unroll
            for (int q = 0; q < kBurst; ++q) carry[q] = d[4 * NA - kBurst + q];
            b = bn;
            if (++since == kFlushGroups) {
                since = 0;
                promote_rows<NA>(acc, sc, tot0, tot1);
            }
        }
#pragma unroll
        for (int q = 0; q < kBurst; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[NA - kBurst + q][r] += __builtin_fabsf(carry[q][r]);
        promote_rows<NA>(acc, sc, tot0, tot1);
    }

    // lane (row, col) owns pairs p0, p0+1 of its row: candidate = tile a, row*4 + r
    double* out = partial + (int64_t)split * n_ctiles * 16;
    const int row = lane >> 4, col = lane & 15;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const int pair = promoted_pair(col, which);
        const int a = pair >> 2, r = pair & 3;
        out[(int64_t)(ctile0 + a) * 16 + row * 4 + r] = which ? tot1 : tot0;
    }
}

// Same contraction, counting |w.p| < thr instead of summing (RANSAC scoring).
// Padding points have residual 0 and are counted; the host subtracts them.
template <int NA>
__global__ __launch_bounds__(256) void inlier_mfma_kernel(
    const float* __restrict__ W, const f32x4* __restrict__ groups,
    const int* __restrict__ group_off, unsigned long long* __restrict__ partial,
    float thr, int n_clouds, int n_ctiles, int n_splits)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ctile0 = (blockIdx.x * 4 + wave) * NA;
    const int split = blockIdx.y;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    unsigned int cnt[NA][4];
    unsigned long long tot[NA][4];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) { cnt[a][r] = 0; tot[a][r] = 0; }

    for (int j = 0; j < n_clouds; ++j) {
        const int g0 = group_off[j];
        const int G = group_off[j + 1] - g0;
        const int gb = g0 + (int)(((int64_t)G * split) / n_splits);
        const int ge = g0 + (int)(((int64_t)G * (split + 1)) / n_splits);
        if (gb >= ge) continue;
        float A[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a)
            A[a] = W[((int64_t)j * n_ctiles + ctile0 + a) * 64 + lane];
        const f32x4* gp = groups + (int64_t)gb * 64 + lane;
        f32x4 b = gp[0];
        for (int g = gb; g < ge; ++g) {
            f32x4 bn = b;
            if (g + 1 < ge) bn = gp[64];
            gp += 64;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(A[a], b[u], zero, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) cnt[a][r] += (__builtin_fabsf(d[r]) < thr) ? 1u : 0u;
                }
            }
            b = bn;
        }
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) { tot[a][r] += cnt[a][r]; cnt[a][r] = 0; }
    }
    unsigned long long* out = partial + (int64_t)split * n_ctiles * 16;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            unsigned long long v = tot[a][r];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            if ((lane & 15) == 0)
                out[(int64_t)(ctile0 + a) * 16 + (lane >> 4) * 4 + r] = v;
        }
}

// cost[c] = sum over splits (fixed order, deterministic)
__global__ __launch_bounds__(256) void reduce_splits_kernel(const double* __restrict__ partial,
                                                            int64_t n_padded, int n_splits,
                                                            int64_t n_valid,
                                                            float* __restrict__ cost)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_valid) return;
    double s = 0.0;
    for (int k = 0; k < n_splits; ++k) s += partial[(int64_t)k * n_padded + c];
    cost[c] = (float)s;
}

__global__ __launch_bounds__(256) void reduce_splits_u64_kernel(
    const unsigned long long* __restrict__ partial, int64_t n_padded, int n_splits,
    int64_t n_valid, unsigned long long* __restrict__ count)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_valid) return;
    unsigned long long s = 0;
    for (int k = 0; k < n_splits; ++k) s += partial[(int64_t)k * n_padded + c];
    count[c] = s;
}

// ---------------------------------------------------------------------------
// arg-min over a cost array: lowest cost, lowest index on ties
// ---------------------------------------------------------------------------
struct MinRec {
    float cost;
    int pad;
    int64_t index;
};

__device__ inline bool rec_less(float ca, int64_t ia, float cb, int64_t ib)
{
    return ca < cb || (ca == cb && ia < ib);
}

__device__ inline void block_min(float& c, int64_t& i, float* sc, int64_t* si)
{
    for (int off = 32; off > 0; off >>= 1) {
        float oc = __shfl_xor(c, off, 64);
        int64_t oi = __shfl_xor(i, off, 64);
        if (rec_less(oc, oi, c, i)) { c = oc; i = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sc[wave] = c; si[wave] = i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (rec_less(sc[w], si[w], c, i)) { c = sc[w]; i = si[w]; }
    }
}

__global__ __launch_bounds__(256) void argmin_stage1_kernel(const float* __restrict__ cost,
                                                            int64_t n, MinRec* __restrict__ out)
{
    __shared__ float sc[4];
    __shared__ int64_t si[4];
    float c = __builtin_inff();
    int64_t i = INT64_MAX;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
         k += (int64_t)gridDim.x * blockDim.x) {
        float v = cost[k];
        if (rec_less(v, k, c, i)) { c = v; i = k; }
    }
    block_min(c, i, sc, si);
    if (threadIdx.x == 0) { out[blockIdx.x].cost = c; out[blockIdx.x].index = i; }
}

__global__ __launch_bounds__(256) void argmin_stage2_kernel(const MinRec* __restrict__ in, int n,
                                                            MinRec* __restrict__ out)
{
    __shared__ float sc[4];
    __shared__ int64_t si[4];
    float c = __builtin_inff();
    int64_t i = INT64_MAX;
    for (int k = threadIdx.x; k < n; k += blockDim.x)
        if (rec_less(in[k].cost, in[k].index, c, i)) { c = in[k].cost; i = in[k].index; }
    block_min(c, i, sc, si);
    if (threadIdx.x == 0) { out->cost = c; out->index = i; }
}

// Appends every candidate with cost <= min*(1+band) to `list` (order arbitrary;
// the host sorts).  count may exceed cap: entries past cap are dropped and the
// host reports the overflow.
__global__ __launch_bounds__(256) void select_band_kernel(const float* __restrict__ cost,
                                                          int64_t n, const MinRec* __restrict__ mn,
                                                          float band, int64_t* __restrict__ list,
                                                          unsigned int cap,
                                                          unsigned int* __restrict__ count)
{
    const float limit = mn->cost * (1.0f + band);
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
         k += (int64_t)gridDim.x * blockDim.x) {
        if (cost[k] <= limit) {
            unsigned int slot = atomicAdd(count, 1u);
            if (slot < cap) list[slot] = k;
        }
    }
}

// ---------------------------------------------------------------------------
// Exact path
// ---------------------------------------------------------------------------
// A block covers a contiguous run of points of one cloud.
struct BlockDesc {
    int cloud;
    int first_of_cloud;  // index of the cloud's first block (for the finaliser)
    int64_t begin;
    int64_t end;
};

// Per (pose, cloud): rows 0..2 of M = base_T_flange_j * F(pose), 12 doubles.
// Per pose: plane a,b,c,d and sqrt(a^2+b^2+c^2), 5 doubles (padded to 8).
constexpr int kPlaneStride = 8;

__device__ inline double exact_distance(const double* M, const double* pl, float x, float y,
                                        float z)
{
    // src/helpers.hpp:139-151: p = col3; p = v_i*col_i + p (i = 0,1,2); -> float
    const double xd = (double)x, yd = (double)y, zd = (double)z;
    float q[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double p = M[r * 4 + 3];
        p = xd * M[r * 4 + 0] + p;
        p = yd * M[r * 4 + 1] + p;
        p = zd * M[r * 4 + 2] + p;
        q[r] = (float)p;
    }
    // src/camera_calibration_optimization.cpp:121-129
    const double num = pl[0] * (double)q[0] + pl[1] * (double)q[1] + pl[2] * (double)q[2] + pl[3];
    return __builtin_fabs(num) / pl[4];
}

// The viewer's plane-mode colouring (src/algorithms.cpp:379-385) transforms the point with
// the generic FP64 matrix product (k = 0..3 accumulated from zero) and does NOT round it to
// float before the distance (pointToPlaneDistanceD, :255-258).
__device__ inline double viewer_distance(const double* M, const double* pl, float x, float y, float z)
{
    const double xd = (double)x, yd = (double)y, zd = (double)z;
    double q[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double p = M[r * 4 + 0] * xd;
        p = M[r * 4 + 1] * yd + p;
        p = M[r * 4 + 2] * zd + p;
        p = M[r * 4 + 3] * 1.0 + p;
        q[r] = p;
    }
    const double num = pl[0] * q[0] + pl[1] * q[1] + pl[2] * q[2] + pl[3];
    return __builtin_fabs(num) / pl[4];
}

// One pose: distance of every point (either arithmetic) and its error class
// (src/algorithms.cpp:390-402: < 2.5 mm, < 5 mm, < 10 mm, else).
__global__ __launch_bounds__(256) void point_error_kernel(
    const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
    const BlockDesc* __restrict__ blocks, const double* __restrict__ mats,
    const double* __restrict__ plane, int viewer_arithmetic, double* __restrict__ dist,
    unsigned char* __restrict__ cls)
{
    __shared__ double sM[12];
    __shared__ double sP[kPlaneStride];
    const BlockDesc bd = blocks[blockIdx.x];
    if (threadIdx.x < 12) sM[threadIdx.x] = mats[(int64_t)bd.cloud * 12 + threadIdx.x];
    if (threadIdx.x < kPlaneStride) sP[threadIdx.x] = plane[threadIdx.x];
    __syncthreads();
    for (int64_t i = bd.begin + threadIdx.x; i < bd.end; i += blockDim.x) {
        const double d = viewer_arithmetic ? viewer_distance(sM, sP, xs[i], ys[i], zs[i])
                                           : exact_distance(sM, sP, xs[i], ys[i], zs[i]);
        dist[i] = d;
        const double mm = d * 1000.0;
        cls[i] = mm < 2.5 ? 0 : (mm < 5.0 ? 1 : (mm < 10.0 ? 2 : 3));
    }
}

// Distances of every point for n_poses poses: dist[k][i] (pose-major, i = global
// point index).  The sums are NOT formed here: the reference adds the distances
// of a cloud one by one in point order
// (src/camera_calibration_optimization.cpp:305), and floating-point addition is
// not associative, so the ordered sum (and the maximum) is the separate chain
// pass below.
constexpr int kMaxPosesPerPass = 64;

__global__ __launch_bounds__(256) void exact_dist_kernel(
    const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
    const BlockDesc* __restrict__ blocks, const double* __restrict__ mats,
    const double* __restrict__ planes, int n_poses, int n_clouds, int64_t total_pts,
    double* __restrict__ dist)
{
    __shared__ double sM[kMaxPosesPerPass * 12];
    __shared__ double sP[kMaxPosesPerPass * kPlaneStride];
    const BlockDesc bd = blocks[blockIdx.x];
    for (int i = threadIdx.x; i < n_poses * 12; i += blockDim.x) {
        int k = i / 12, e = i % 12;
        sM[i] = mats[((int64_t)k * n_clouds + bd.cloud) * 12 + e];
    }
    for (int i = threadIdx.x; i < n_poses * kPlaneStride; i += blockDim.x) sP[i] = planes[i];
    __syncthreads();
    for (int64_t i0 = bd.begin + threadIdx.x; i0 < bd.end; i0 += 4 * blockDim.x) {
        float x[4], y[4], z[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int64_t i = i0 + (int64_t)p * blockDim.x;
            const bool in = i < bd.end;
            x[p] = in ? xs[i] : 0.f; y[p] = in ? ys[i] : 0.f; z[p] = in ? zs[i] : 0.f;
        }
        for (int k = 0; k < n_poses; ++k) {
            const double* M = sM + k * 12;
            const double* pl = sP + k * kPlaneStride;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int64_t i = i0 + (int64_t)p * blockDim.x;
                if (i < bd.end) dist[(int64_t)k * total_pts + i] = exact_distance(M, pl, x[p], y[p], z[p]);
            }
        }
    }
}

// Ordered sum of one (pose k, cloud j) chain: average = ((0 + d_0) + d_1) + ... in
// point order with a rounding after every add, exactly the reference's
// accumulation -- computed by one workgroup per chain as an exact parallel
// emulation of the sequential sum:
//
//   While the running sum s stays inside one binade [2^e, 2^(e+1)) every add
//   rounds to a multiple of u = 2^(e-52).  With S = s/u (an integer in
//   [2^52, 2^53)) and t = d/u = q + r (q integer, 0 <= r < 1), round-to-nearest-
//   even gives  S' = S + q + [r > 1/2]  and, only for an exact tie r == 1/2,
//   S' = S + q + ((S + q) odd).  So an element is a map  parity(S) -> increment,
//   and maps compose associatively:  (A then B)[p] = A[p] + B[(p + A[p]) & 1].
//   Each lane folds 8 consecutive elements into its pair (inc[0], inc[1]), a
//   wave composes its 64 pairs in lane order with a shuffle tree (one 512-element
//   segment), and thread 0 applies the 16 segment maps of the workgroup in order
//   to the actual S -- a segment is accepted iff S + total < 2^53, which (all
//   increments being >= 0) holds exactly when no add inside it left the binade.
//   The first rejected segment (about once per binade, ~20 times per chain) is
//   replayed with plain one-by-one adds and the pass restarts behind it with the
//   new binade; the first kChainWarm elements, where binades change every few
//   adds, are summed one by one up front.
//
// Distances are >= 0 by construction.
constexpr int kChainRun = 8;                       // consecutive elements per lane
constexpr int kChainSeg = 64 * kChainRun;          // elements per wave segment
constexpr int kChainWaves = 16;                    // segments per pass
constexpr int kChainPass = kChainSeg * kChainWaves;

__device__ inline void chain_fold(const double v[kChainRun], double scale, unsigned long long& inc0,
                                  unsigned long long& inc1, bool& ok)
{
    const double two53 = 9007199254740992.0;
    inc0 = 0; inc1 = 0; ok = true;
#pragma unroll
    for (int t = 0; t < kChainRun; ++t) {
        const double tt = v[t] * scale;             // exact (power of two)
        ok = ok && (tt < two53);                    // also false for NaN
        const double tc = tt < two53 ? tt : 0.0;
        const unsigned long long q = (unsigned long long)tc;
        const double r = tc - (double)q;            // exact
        const unsigned long long up = r > 0.5 ? 1ull : 0ull;
        const bool tie = r == 0.5;
        inc0 += q + (tie ? ((inc0 + q) & 1ull) : up);
        inc1 += q + (tie ? ((1ull + inc1 + q) & 1ull) : up);
    }
    // compose the 64 lane maps in lane order (left operand = lower lane)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long r0 = __shfl_down(inc0, off, 64);
        const unsigned long long r1 = __shfl_down(inc1, off, 64);
        inc0 += (inc0 & 1ull) ? r1 : r0;
        inc1 += ((1ull + inc1) & 1ull) ? r1 : r0;
    }
    ok = __all(ok ? 1 : 0);
}

// Also returns the chain's maximum (:306-307; order-free).
__global__ __launch_bounds__(64 * kChainWaves) void ordered_sum_kernel(
    const double* __restrict__ dist, const int64_t* __restrict__ cloud_pt_off, int n_chains,
    int n_clouds, int64_t total_pts, double* __restrict__ out_sum, double* __restrict__ out_max)
{
    __shared__ double sh_vals[kChainWaves][kChainSeg];   // this pass's elements (64 KiB)
    __shared__ unsigned long long sh_inc[kChainWaves][2];
    __shared__ int sh_ok[kChainWaves];
    __shared__ double sh_red[kChainWaves];
    __shared__ double sh_s;
    __shared__ int sh_first;
    const int c = blockIdx.x;
    if (c >= n_chains) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = c / n_clouds, j = c % n_clouds;
    const int64_t b = cloud_pt_off[j], e = cloud_pt_off[j + 1];
    const double* row = dist + (int64_t)k * total_pts;
    const int my = wave * kChainSeg + lane * kChainRun;   // this lane's offset inside a pass

    double s = 0.0, mx = -1e9;                            // :288 error_mx = -1e9
    double v[kChainRun], vn[kChainRun];
#pragma unroll
    for (int t = 0; t < kChainRun; ++t) v[t] = b + my + t < e ? row[b + my + t] : 0.0;   // a zero adds nothing

    for (int64_t i = b; i < e; i += kChainPass) {
        const int64_t left = e - i;
        const int nseg = left >= kChainPass ? kChainWaves : (int)((left + kChainSeg - 1) / kChainSeg);
        // next pass in flight while this one is folded
#pragma unroll
        for (int t = 0; t < kChainRun; ++t) {
            const int64_t idx = i + kChainPass + my + t;
            vn[t] = idx < e ? row[idx] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < kChainRun; ++t) {
            sh_vals[wave][lane * kChainRun + t] = v[t];
            if (i + my + t < e) mx = v[t] > mx ? v[t] : mx;
        }
        int first = 0;
        for (;;) {
            const unsigned long long sbits = (unsigned long long)__double_as_longlong(s);
            const int ebiased = (int)((sbits >> 52) & 0x7ff);
            const bool usable = ebiased >= 64 && ebiased <= 2000;
            // scale = 1/u = 2^(52 - e): exponent field 1023 + 52 - (ebiased - 1023)
            const double scale =
                usable ? __longlong_as_double((long long)((unsigned long long)(2098 - ebiased) << 52)) : 1.0;
            if (usable && wave >= first && wave < nseg) {
                unsigned long long inc0, inc1;
                bool ok;
                chain_fold(v, scale, inc0, inc1, ok);
                if (lane == 0) { sh_inc[wave][0] = inc0; sh_inc[wave][1] = inc1; sh_ok[wave] = ok ? 1 : 0; }
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int adv = first;
                double sn = s;
                if (usable) {
                    unsigned long long S = (unsigned long long)(s * scale);   // exact integer in [2^52, 2^53)
                    for (; adv < nseg; ++adv) {
                        const unsigned long long total = sh_inc[adv][S & 1ull];
                        if (!sh_ok[adv] || total >= (1ull << 53) || S + total >= (1ull << 53)) break;
                        S += total;
                    }
                    sn = (double)S / scale;                                  // exact
                }
                if (adv < nseg) {
                    // the segment that left the binade (or s not yet usable): one by one
                    for (int t = 0; t < kChainSeg; ++t) sn += sh_vals[adv][t];
                    ++adv;
                }
                sh_s = sn;
                sh_first = adv;
            }
            __syncthreads();
            s = sh_s;
            first = sh_first;
            if (first >= nseg) break;
        }
#pragma unroll
        for (int t = 0; t < kChainRun; ++t) v[t] = vn[t];
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double om = __shfl_xor(mx, off, 64);
        mx = om > mx ? om : mx;
    }
    if (lane == 0) sh_red[wave] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kChainWaves; ++w) mx = sh_red[w] > mx ? sh_red[w] : mx;
        out_sum[c] = s;
        out_max[c] = mx;
    }
}

// Moments of the base-frame points q = float(M p): for the z-regression plane fit
// (src/camera_calibration_optimization.cpp:100-119 on the cloud of :240-272) and,
// with an inlier mask, for the RANSAC refit.  Accumulated about a fixed origin
// `org` (an approximate centroid) so the normal equations / covariance stay well
// conditioned in FP64.  Order of the 10 sums: 1, x, y, z, xx, xy, yy, xz, yz, zz.
// mask: plane (a,b,c,d,|n|) and threshold; thr < 0 keeps every point.
constexpr int kMoments = 10;

struct InlierMask {
    double pl[5];
    double thr;
};

__global__ __launch_bounds__(256) void moments_kernel(
    const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
    const BlockDesc* __restrict__ blocks, const double* __restrict__ mats, double ox, double oy,
    double oz, InlierMask mask, double* __restrict__ part)
{
    __shared__ double sM[12];
    __shared__ double red[4][kMoments];
    const BlockDesc bd = blocks[blockIdx.x];
    if (threadIdx.x < 12) sM[threadIdx.x] = mats[(int64_t)bd.cloud * 12 + threadIdx.x];
    __syncthreads();
    double m[kMoments];
#pragma unroll
    for (int k = 0; k < kMoments; ++k) m[k] = 0.0;
    for (int64_t i = bd.begin + threadIdx.x; i < bd.end; i += blockDim.x) {
        const double xd = (double)xs[i], yd = (double)ys[i], zd = (double)zs[i];
        float q[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double p = sM[r * 4 + 3];
            p = xd * sM[r * 4 + 0] + p;
            p = yd * sM[r * 4 + 1] + p;
            p = zd * sM[r * 4 + 2] + p;
            q[r] = (float)p;
        }
        if (mask.thr >= 0.0) {
            const double num = mask.pl[0] * (double)q[0] + mask.pl[1] * (double)q[1] +
                               mask.pl[2] * (double)q[2] + mask.pl[3];
            if (!(__builtin_fabs(num) / mask.pl[4] < mask.thr)) continue;
        }
        const double x = (double)q[0] - ox, y = (double)q[1] - oy, z = (double)q[2] - oz;
        m[0] += 1.0; m[1] += x; m[2] += y; m[3] += z;
        m[4] += x * x; m[5] += x * y; m[6] += y * y; m[7] += x * z; m[8] += y * z; m[9] += z * z;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kMoments; ++k) {
        double s = m[k];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < kMoments) {
        const int k = threadIdx.x;
        part[(int64_t)blockIdx.x * kMoments + k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    }
}

__global__ __launch_bounds__(64) void moments_finalize_kernel(const double* __restrict__ part,
                                                              int n_blocks,
                                                              double* __restrict__ out)
{
    const int k = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += 64) s += part[(int64_t)b * kMoments + k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) out[k] = s;
}

// ---------------------------------------------------------------------------
// RANSAC plane hypotheses
// ---------------------------------------------------------------------------
// Counter-based generator: splitmix64 of (seed + counter).
__host__ __device__ inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// Hypothesis h samples three distinct global point indices
//   i_k = splitmix64(seed + 3h + k) mod N   (a repeat is bumped to the next index),
// carries them to the base frame with the reference's transform arithmetic and
// spans the plane: n = (q1-q0) x (q2-q0) normalised with n.z >= 0, d = -n.q0.
// planes[h] = (a,b,c,d,|n|,valid).
constexpr int kHypStride = 8;

__global__ __launch_bounds__(256) void ransac_hyp_kernel(
    const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
    const int64_t* __restrict__ cloud_pt_off, int n_clouds, int64_t total_pts,
    const double* __restrict__ mats, uint64_t seed, int n_hyp, double* __restrict__ planes)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_hyp) return;
    int64_t idx[3];
    for (int k = 0; k < 3; ++k) {
        int64_t v = (int64_t)(splitmix64(seed + 3ull * (uint64_t)h + (uint64_t)k) % (uint64_t)total_pts);
        for (int t = 0; t < k; ++t)
            if (v == idx[t]) { v = (v + 1) % total_pts; t = -1; }
        idx[k] = v;
    }
    double q[3][3];
    for (int k = 0; k < 3; ++k) {
        int j = 0;
        while (j + 1 < n_clouds && idx[k] >= cloud_pt_off[j + 1]) ++j;
        const double* M = mats + 12 * j;
        const double xd = (double)xs[idx[k]], yd = (double)ys[idx[k]], zd = (double)zs[idx[k]];
        for (int r = 0; r < 3; ++r) {
            double p = M[r * 4 + 3];
            p = xd * M[r * 4 + 0] + p;
            p = yd * M[r * 4 + 1] + p;
            p = zd * M[r * 4 + 2] + p;
            q[k][r] = (double)(float)p;
        }
    }
    const double u[3] = {q[1][0] - q[0][0], q[1][1] - q[0][1], q[1][2] - q[0][2]};
    const double v[3] = {q[2][0] - q[0][0], q[2][1] - q[0][1], q[2][2] - q[0][2]};
    double n[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
    const double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    double* out = planes + (int64_t)h * kHypStride;
    if (!(len > 1e-12)) {
        for (int e = 0; e < kHypStride; ++e) out[e] = 0.0;   // collinear sample: invalid
        out[4] = 1.0;
        return;
    }
    if (n[2] < 0.0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    n[0] = n[0] / len; n[1] = n[1] / len; n[2] = n[2] / len;
    out[0] = n[0]; out[1] = n[1]; out[2] = n[2];
    out[3] = -(n[0] * q[0][0] + n[1] * q[0][1] + n[2] * q[0][2]);
    out[4] = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    out[5] = 1.0;
    out[6] = 0.0; out[7] = 0.0;
}

// Folded 4-vectors of every (hypothesis, cloud) for the fixed pose matrices.
__global__ __launch_bounds__(256) void wgen_planes_kernel(const double* __restrict__ planes,
                                                          int n_hyp, int n_ctiles, int n_clouds,
                                                          const double* __restrict__ mats,
                                                          const double* __restrict__ centres,
                                                          float* __restrict__ W)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)n_ctiles * kTilePoints) return;
    const bool valid = c < n_hyp && planes[c * kHypStride + 5] != 0.0;
    double n[3] = {0, 0, 0}, d = 0, inv = 1;
    if (valid) {
        const double* pl = planes + c * kHypStride;
        n[0] = pl[0]; n[1] = pl[1]; n[2] = pl[2]; d = pl[3];
        inv = 1.0 / pl[4];
    }
    const int64_t tile = c >> 4;
    const int col = (int)(c & 15);
    for (int j = 0; j < n_clouds; ++j) {
        double w[4] = {0, 0, 0, 0};
        if (valid) fold_plane(mats + 12 * j, n, d, inv, centres + 3 * j, w);
        float* dst = W + ((int64_t)j * n_ctiles + tile) * 64 + col;
        dst[0] = (float)w[0];
        dst[16] = (float)w[1];
        dst[32] = (float)w[2];
        dst[48] = (float)w[3];
    }
}

// Exact inlier counts for PB hypotheses (reference-arithmetic distance < thr).
template <int PB>
__global__ __launch_bounds__(256) void exact_count_kernel(
    const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
    const BlockDesc* __restrict__ blocks, const double* __restrict__ mats,
    const double* __restrict__ planes /* PB x kPlaneStride */, double thr,
    unsigned long long* __restrict__ counts)
{
    __shared__ double sM[12];
    __shared__ double sP[PB * kPlaneStride];
    const BlockDesc bd = blocks[blockIdx.x];
    if (threadIdx.x < 12) sM[threadIdx.x] = mats[(int64_t)bd.cloud * 12 + threadIdx.x];
    for (int i = threadIdx.x; i < PB * kPlaneStride; i += blockDim.x) sP[i] = planes[i];
    __syncthreads();
    unsigned int cnt[PB];
#pragma unroll
    for (int k = 0; k < PB; ++k) cnt[k] = 0;
    for (int64_t i = bd.begin + threadIdx.x; i < bd.end; i += blockDim.x) {
        const float x = xs[i], y = ys[i], z = zs[i];
#pragma unroll
        for (int k = 0; k < PB; ++k)
            cnt[k] += exact_distance(sM, sP + k * kPlaneStride, x, y, z) < thr ? 1u : 0u;
    }
#pragma unroll
    for (int k = 0; k < PB; ++k) {
        unsigned int c = cnt[k];
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(&counts[k], (unsigned long long)c);
    }
}

} // namespace ccal
