// kernels_residual.hpp -- the dense sweep: candidate 4-vectors, the residual kernels on the matrix cores, split reduction, arg-min and re-score selection
//
// One of the kernel headers of libccal; each is included by exactly one translation unit
// (api_search.hip), so its __global__ functions have one definition in the library.
#pragma once

#include "kernels_common.hpp"
#include "bf16x3_split.hpp"

#pragma clang fp contract(off)

namespace ccal {

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
// (split_bf16x3, split_point_operand, split_candidate_operand: bf16x3_split.hpp)

// Promotes the FP32 lane accumulators of a wave into NA/4 FP64 totals per lane.
// acc[a][r] of lane l belongs to candidate (l>>4)*4 + r of tile a and point column
// l&15; the 16 lanes of a row are reduce-scattered (4 exchange steps: every step a
// lane keeps one half of its values and adds the partner's copy of that half), after
// which lane column c owns NA/4 consecutive row totals, value = a*4 + r.
template <int NA>
__device__ inline void promote_rows(float (&acc)[NA][4], double scale, double (&tot)[NA / 4])
{
    static_assert(NA == 8 || NA == 16, "the reduce-scatter needs 4 NA values per lane to be a multiple of 16");
    constexpr int V = 4 * NA;
    const int col = threadIdx.x & 15;
    float v[V];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[a * 4 + r] = acc[a][r]; acc[a][r] = 0.f; }
    float w2[V / 2];
    {
        const bool hi = (col & 8) != 0;
#pragma unroll
        for (int i = 0; i < V / 2; ++i) {
            const float keep = hi ? v[V / 2 + i] : v[i];
            const float give = hi ? v[i] : v[V / 2 + i];
            w2[i] = keep + __shfl_xor(give, 8, 64);
        }
    }
    float w4[V / 4];
    {
        const bool hi = (col & 4) != 0;
#pragma unroll
        for (int i = 0; i < V / 4; ++i) {
            const float keep = hi ? w2[V / 4 + i] : w2[i];
            const float give = hi ? w2[i] : w2[V / 4 + i];
            w4[i] = keep + __shfl_xor(give, 4, 64);
        }
    }
    float w8[V / 8];
    {
        const bool hi = (col & 2) != 0;
#pragma unroll
        for (int i = 0; i < V / 8; ++i) {
            const float keep = hi ? w4[V / 8 + i] : w4[i];
            const float give = hi ? w4[i] : w4[V / 8 + i];
            w8[i] = keep + __shfl_xor(give, 2, 64);
        }
    }
    {
        const bool hi = (col & 1) != 0;
#pragma unroll
        for (int i = 0; i < V / 16; ++i) {
            const float keep = hi ? w8[V / 16 + i] : w8[i];
            const float give = hi ? w8[i] : w8[V / 16 + i];
            tot[i] += (double)(keep + __shfl_xor(give, 1, 64)) * scale;
        }
    }
}

// Value index (a*4 + r) of total `which` of lane column c after promote_rows: bit 3 of c selects the
// upper half of the 4 NA values, bit 2 the upper half of that, and so on; the totals are consecutive.
template <int NA>
__device__ inline int promoted_value(int col, int which)
{
    constexpr int V = 4 * NA;
    return ((col & 8) ? V / 2 : 0) + ((col & 4) ? V / 4 : 0) + ((col & 2) ? V / 8 : 0) + ((col & 1) ? V / 16 : 0) + which;
}

// NA = candidate tiles per wave: 8 (three waves per SIMD) or 16 (two: the split of a point tile and its
// load are shared by twice as many matrix instructions).
template <int NA, int kBurst = 4>
__global__ __launch_bounds__(256, NA == 16 ? 2 : 3) void residual_bf16x3_kernel(
    const float* __restrict__ W, const f32x4* __restrict__ groups,
    const int* __restrict__ group_off, const double* __restrict__ scale,
    double* __restrict__ partial, int n_clouds, int n_ctiles, int n_splits)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ctile0 = (blockIdx.x * 4 + wave) * NA;
    const int split = blockIdx.y;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    double tot[NA / 4] = {};

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
                promote_rows<NA>(acc, sc, tot);
            }
        }
#pragma unroll
        for (int q = 0; q < kBurst; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[NA - kBurst + q][r] += __builtin_fabsf(carry[q][r]);
        promote_rows<NA>(acc, sc, tot);
    }

    // lane (row, col) owns pairs p0, p0+1 of its row: candidate = tile a, row*4 + r
    double* out = partial + (int64_t)split * n_ctiles * 16;
    const int row = lane >> 4, col = lane & 15;
#pragma unroll
    for (int which = 0; which < NA / 4; ++which) {
        const int value = promoted_value<NA>(col, which);
        const int a = value >> 2, r = value & 3;
        out[(int64_t)(ctile0 + a) * 16 + row * 4 + r] = tot[which];
    }
}

// ---------------------------------------------------------------------------
// Wide form of the bf16x3 kernel: v_mfma_f32_32x32x16_bf16 pairs
// ---------------------------------------------------------------------------
// Every MFMA keeps the VALU port closed for ~8-10 cycles whatever its size, and
// one |d| add per residual and lane is irreducible, so the 16x16x32 form pays one
// MFMA issue per 4 adds.  A 32x32x16 MFMA covers 32 candidates x 32 points with
// K = 16, two chained ones (x|y slots, then z|constant slots) give the same
// K = 32 contraction for 1024 residuals: one MFMA issue per 8 adds.  The point
// operands are split once, at upload, into their bf16x3 K-slot form
// (split_groups_kernel), so the kernel loads them ready-made (64 B per point
// instead of 16; the stream stays L2-resident per candidate block as before).
//
// Operand layout of v_mfma_f32_32x32x16_bf16: lane l supplies row/column l&31 and
// the eight K-slots 8*(l>>5)..+7.  First MFMA: lanes 0-31 carry coordinate x,
// lanes 32-63 coordinate y; second: z and the constant.  The result puts point
// column l&31 and candidate rows 8*(i/4) + 4*(l>>5) + i%4 (i = 0..15) in lane l.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kFlushTiles32 = 128;  // FP32 partial sums are promoted every 128 adds

// bq[((tile32 * 2 + h) * 64 + lane]: operand h of the 32-point tile, tile32 = 2*group + s.
__global__ __launch_bounds__(256) void split_groups_kernel(const f32x4* __restrict__ groups,
                                                            int64_t n_groups, u32x4* __restrict__ bq)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_groups * 256) return;
    const int lane = (int)(idx & 63), h = (int)((idx >> 6) & 1), s = (int)((idx >> 7) & 1);
    const int64_t g = idx >> 8;
    const int n = lane & 31, k = 2 * h + (lane >> 5);
    const int u = 2 * s + (n >> 4), col = n & 15;
    const f32x4 src = groups[g * 64 + k * 16 + col];
    bq[idx] = __builtin_bit_cast(u32x4, split_point_operand(src[u]));
}

// Promotes the FP32 lane accumulators (64 per lane: tile a, register i) into two
// FP64 totals per lane by a reduce-scatter over the 32 point columns of each lane
// half (5 exchange steps); afterwards lane column n owns values 2n and 2n+1,
// value = a*16 + i.
template <int NA>
__device__ inline void promote_wide(float (&acc)[NA][16], double scale, double& tot0, double& tot1)
{
    static_assert(NA == 4, "the reduce-scatter below is written for 64 values per lane");
    const int n = threadIdx.x & 31;
    float v[64];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) { v[a * 16 + i] = acc[a][i]; acc[a][i] = 0.f; }
    float w32[32];
    {
        const bool hi = (n & 16) != 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const float keep = hi ? v[32 + i] : v[i];
            const float give = hi ? v[i] : v[32 + i];
            w32[i] = keep + __shfl_xor(give, 16, 64);
        }
    }
    float w16[16];
    {
        const bool hi = (n & 8) != 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float keep = hi ? w32[16 + i] : w32[i];
            const float give = hi ? w32[i] : w32[16 + i];
            w16[i] = keep + __shfl_xor(give, 8, 64);
        }
    }
    float w8[8];
    {
        const bool hi = (n & 4) != 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float keep = hi ? w16[8 + i] : w16[i];
            const float give = hi ? w16[i] : w16[8 + i];
            w8[i] = keep + __shfl_xor(give, 4, 64);
        }
    }
    float w4[4];
    {
        const bool hi = (n & 2) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float keep = hi ? w8[4 + i] : w8[i];
            const float give = hi ? w8[i] : w8[4 + i];
            w4[i] = keep + __shfl_xor(give, 2, 64);
        }
    }
    {
        const bool hi = (n & 1) != 0;
        const float k0 = hi ? w4[2] : w4[0], g0 = hi ? w4[0] : w4[2];
        const float k1 = hi ? w4[3] : w4[1], g1 = hi ? w4[1] : w4[3];
        tot0 += (double)(k0 + __shfl_xor(g0, 1, 64)) * scale;
        tot1 += (double)(k1 + __shfl_xor(g1, 1, 64)) * scale;
    }
}

// grid = (candidate blocks of 512, point splits); block = 4 waves; wave w of block
// bx owns the NA 32-candidate tiles starting at candidate (bx*4 + w) * 32 * NA.
// W is the 16-wide tile layout of wgen_kernel (n_ctiles16 tiles per cloud).
template <int NA>
__global__ __launch_bounds__(256, 3) void residual_bf16x3_wide_kernel(
    const float* __restrict__ W, const u32x4* __restrict__ bq,
    const int* __restrict__ group_off, const double* __restrict__ scale,
    double* __restrict__ partial, int n_clouds, int n_ctiles16, int n_splits)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int m = lane & 31, half = lane >> 5;
    const int ct32_0 = (blockIdx.x * 4 + wave) * NA;
    const int split = blockIdx.y;
    __shared__ u32x4 stage[2][4 * 2 * 64];   // two stages of four tiles x two operands (16 KiB)
    f32x16 zero;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero[i] = 0.f;
    double tot0 = 0.0, tot1 = 0.0;

    for (int j = 0; j < n_clouds; ++j) {
        const int g0 = group_off[j];
        const int G = group_off[j + 1] - g0;
        const int gb = g0 + (int)(((int64_t)G * split) / n_splits);
        const int ge = g0 + (int)(((int64_t)G * (split + 1)) / n_splits);
        if (gb >= ge) continue;
        bf16x8 A1[NA], A2[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int ct16 = (ct32_0 + a) * 2 + (m >> 4);
            const float* w = W + ((int64_t)j * n_ctiles16 + ct16) * 64 + (m & 15);
            A1[a] = split_candidate_operand(w[half * 16], false);
            A2[a] = split_candidate_operand(w[(2 + half) * 16], half == 1);
        }
        const double sc = scale[j];

        float acc[NA][16];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;

        // Stages of four 32-point tiles go through LDS: wave w fetches tile w of the next
        // stage from global memory while all four waves contract the current stage, so the
        // L2 sees every operand once per workgroup instead of once per wave.
        const int t_begin = 2 * gb, n_tiles = 2 * (ge - gb);
        const int n_stages = (n_tiles + 3) / 4;
        {
            const int t = wave;
            if (t < n_tiles) {
                const u32x4* src = bq + ((int64_t)(t_begin + t) * 2) * 64 + lane;
                stage[0][(wave * 2 + 0) * 64 + lane] = src[0];
                stage[0][(wave * 2 + 1) * 64 + lane] = src[64];
            }
        }
        __syncthreads();
        f32x16 carry = zero;
        int since = 0;
        for (int st = 0; st < n_stages; ++st) {
            const int t_next = 4 * (st + 1) + wave;
            const bool have_next = t_next < n_tiles;
            u32x4 g1 = {0u, 0u, 0u, 0u}, g2 = {0u, 0u, 0u, 0u};
            if (have_next) {
                const u32x4* src = bq + ((int64_t)(t_begin + t_next) * 2) * 64 + lane;
                g1 = src[0];
                g2 = src[64];
            }
            const u32x4* buf = stage[st & 1] + lane;
            const int tiles_here = n_tiles - 4 * st < 4 ? n_tiles - 4 * st : 4;
            u32x4 b1 = buf[0], b2 = buf[64];
            for (int tt = 0; tt < tiles_here; ++tt) {
                u32x4 n1 = b1, n2 = b2;
                if (tt + 1 < tiles_here) { n1 = buf[(tt + 1) * 128]; n2 = buf[(tt + 1) * 128 + 64]; }
                const bf16x8 B1 = __builtin_bit_cast(bf16x8, b1);
                const bf16x8 B2 = __builtin_bit_cast(bf16x8, b2);
                f32x16 d[NA];
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const f32x16 done = a == 0 ? carry : d[a == 0 ? 0 : a - 1];
                    const int pa = (a + NA - 1) % NA;
                    d[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1[a], B1, zero, 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[pa][i] += __builtin_fabsf(done[i]);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                    d[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2[a], B2, d[a], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
                    for (int i = 8; i < 16; ++i) acc[pa][i] += __builtin_fabsf(done[i]);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                }
                carry = d[NA - 1];
                b1 = n1;
                b2 = n2;
                if (++since == kFlushTiles32) {
                    since = 0;
                    promote_wide<NA>(acc, sc, tot0, tot1);
                }
            }
            if (have_next) {
                stage[(st + 1) & 1][(wave * 2 + 0) * 64 + lane] = g1;
                stage[(st + 1) & 1][(wave * 2 + 1) * 64 + lane] = g2;
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[NA - 1][i] += __builtin_fabsf(carry[i]);
        promote_wide<NA>(acc, sc, tot0, tot1);
    }

    // lane (half, column n) owns values 2n and 2n+1: tile a = value >> 4, register i = value & 15
    double* out = partial + (int64_t)split * n_ctiles16 * 16;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const int value = 2 * m + which;
        const int a = value >> 4, i = value & 15;
        const int row = 8 * (i >> 2) + 4 * half + (i & 3);
        out[(int64_t)(ct32_0 + a) * 32 + row] = which ? tot1 : tot0;
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

// Appends k0 + k for every k in [0, n) with cost[k] (1 - rel) <= min (1 + rel) + abs to `list`
// (order arbitrary; the host sorts).  count may exceed cap: entries past cap are dropped and the
// host walks the range in windows instead.
__global__ __launch_bounds__(256) void select_band_kernel(const float* __restrict__ cost,
                                                          int64_t n, int64_t k0, const MinRec* __restrict__ mn,
                                                          double rel, double abs_band, int64_t* __restrict__ list,
                                                          unsigned int cap,
                                                          unsigned int* __restrict__ count)
{
    // rounded up: a candidate on the edge is kept
    const double limit = ((double)mn->cost * (1.0 + rel) + abs_band) * (1.0 + 0x1p-40);
    const double keep = 1.0 - rel;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
         k += (int64_t)gridDim.x * blockDim.x) {
        if (!((double)cost[k] * keep > limit)) {
            unsigned int slot = atomicAdd(count, 1u);
            if (slot < cap) list[slot] = k0 + k;
        }
    }
}


} // namespace ccal
