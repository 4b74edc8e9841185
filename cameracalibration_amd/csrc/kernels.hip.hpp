// kernels.hip.hpp -- gfx950 kernels of the calibration hot path.
//
// Arithmetic families:
//   * "fast": FP32, the point-to-plane residual folded into a 4-vector per
//     (candidate, cloud) and contracted with homogeneous points on the matrix
//     cores: v_mfma_f32_16x16x4_f32 is exactly a K=4 contraction of a
//     16-candidate tile with a 16-point tile, bit-identical to an fmaf chain.
//     |.| and the running sum are VALU adds on the MFMA result.
//   * "exact": the reference's own arithmetic (FP64 SE3 apply in the SSE2
//     summation order, point rounded to float, FP64 distance with a correctly
//     rounded division, distances of a cloud added one by one in point order),
//     used for single poses, finite-difference stencils, the error report and
//     for re-scoring near-minimum grid candidates.  Its results are
//     bit-identical to the CPU evaluation of the same lines.
//   * "separable": for product grids, all translations of a rotation from one
//     histogram pass over the points (fast-path accuracy, integer sums).
//
// Included once by ccal_api.hip; everything here is internal.
#pragma once

#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "pose_math.hpp"

#pragma clang fp contract(off)

namespace ccal {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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
constexpr int kFlushGroups = 64;  // FP32 partial sums are promoted every 256 adds

// ---------------------------------------------------------------------------
// Upload: device-side re-tiling
// ---------------------------------------------------------------------------
// The caller's clouds arrive as AoS xyz (one contiguous copy per cloud); this
// kernel writes both device layouts from them: the SoA coordinates of the exact
// path and the centred 64-point groups of the fast path (which must be zero
// beforehand: padding points stay all zero).  One thread per point.
// Centre of every cloud from the staged AoS copy: FP64 sums in a fixed order (kCentreParts strided
// partial sums per cloud, each a fixed wave tree, then added in index order), so the centre -- and with
// it every fast-path value -- does not depend on how the copy was scheduled.  A non-finite mean
// becomes 0 (the cloud then carries an infinite radius and no fast-path bound is promised for it).
constexpr int kCentreParts = 64;

__global__ __launch_bounds__(256) void centre_partial_kernel(const float* __restrict__ aos /* all clouds */,
                                                             const int64_t* __restrict__ cloud_pt_off, int cloud0,
                                                             double* __restrict__ part /* [cloud][kCentreParts][3] */)
{
    __shared__ double red[4][3];
    const int j = cloud0 + blockIdx.y;
    const int64_t b = cloud_pt_off[j], e = cloud_pt_off[j + 1];
    double s[3] = {0.0, 0.0, 0.0};
    for (int64_t i = b + (int64_t)blockIdx.x * 256 + threadIdx.x; i < e; i += (int64_t)kCentreParts * 256) {
        s[0] += (double)aos[3 * i + 0];
        s[1] += (double)aos[3 * i + 1];
        s[2] += (double)aos[3 * i + 2];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double v = s[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3)
        part[((int64_t)j * kCentreParts + blockIdx.x) * 3 + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void centre_final_kernel(const double* __restrict__ part,
                                                           const int64_t* __restrict__ cloud_pt_off, int cloud0,
                                                           int cloud1, double* __restrict__ centres /* [cloud][3] */)
{
    const int t = 3 * cloud0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * cloud1) return;
    const int j = t / 3, k = t % 3;
    double s = 0.0;
    for (int b = 0; b < kCentreParts; ++b) s += part[((int64_t)j * kCentreParts + b) * 3 + k];
    s /= (double)(cloud_pt_off[j + 1] - cloud_pt_off[j]);
    centres[t] = (s - s == 0.0) ? s : 0.0;
}

// The exact path's arrays keep the caller's point order.  The fast path's groups do not: slot i holds point
// (i * stride) mod n, stride coprime to n and close to n / golden ratio, so that the points a wave works
// on together come from all over the cloud.  Every fast kernel sums over all points of a cloud and is
// indifferent to their order, except for speed: the histogram pass sends points whose residuals fall
// into the same bin to the same LDS word, and neighbours in scan or voxel order do that all the time.
__host__ __device__ inline int64_t fast_slot_source(int64_t i, int64_t n, int64_t stride)
{
    return (int64_t)(((unsigned long long)i * (unsigned long long)stride) % (unsigned long long)n);
}

__global__ __launch_bounds__(256) void tile_cloud_kernel(const float* __restrict__ aos /* this cloud */,
                                                         int64_t n, int64_t stride, int64_t first_point, int64_t first_group,
                                                         const double* __restrict__ centre /* this cloud */,
                                                         float* __restrict__ xs, float* __restrict__ ys,
                                                         float* __restrict__ zs, float* __restrict__ groups,
                                                         unsigned int* __restrict__ radius2_bits /* this cloud */,
                                                         float* __restrict__ radius_sums /* this cloud: one per block */)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double cx = centre[0], cy = centre[1], cz = centre[2];
    float r2 = 0.f;
    if (i < n) {
        const float x = aos[3 * i + 0], y = aos[3 * i + 1], z = aos[3 * i + 2];
        xs[first_point + i] = x;
        ys[first_point + i] = y;
        zs[first_point + i] = z;
        const int64_t g = first_group + i / kGroupPoints;
        const int in_group = (int)(i % kGroupPoints);
        const int u = in_group / kTilePoints, col = in_group % kTilePoints;
        float* gp = groups + g * 256;
        // lane l = k*16 + col holds coordinate k; component u selects the tile
        const float dx = (float)((double)x - cx), dy = (float)((double)y - cy), dz = (float)((double)z - cz);
        r2 = dx * dx + dy * dy + dz * dz;
        const int64_t src = fast_slot_source(i, n, stride);
        const float sx = aos[3 * src + 0], sy = aos[3 * src + 1], sz = aos[3 * src + 2];
        gp[(0 * 16 + col) * 4 + u] = (float)((double)sx - cx);
        gp[(1 * 16 + col) * 4 + u] = (float)((double)sy - cy);
        gp[(2 * 16 + col) * 4 + u] = (float)((double)sz - cz);
        gp[(3 * 16 + col) * 4 + u] = 1.0f;
    }
    // largest squared distance from the centre: non-negative floats order like their bit patterns,
    // and a NaN's pattern lies above infinity's, so a non-finite point leaves a non-finite mark
    // sum of the distances from the centre, per block in a fixed order (the host adds the blocks)
    __shared__ float wave_sums[4];
    float rs = __builtin_sqrtf(r2);
    for (int off = 32; off > 0; off >>= 1) rs += __shfl_xor(rs, off, 64);
    if ((threadIdx.x & 63) == 0) wave_sums[threadIdx.x >> 6] = rs;
    __shared__ unsigned int wave_bits[4];
    unsigned int bits = __float_as_uint(r2);
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned int o = (unsigned int)__shfl_xor((int)bits, off, 64);
        bits = o > bits ? o : bits;
    }
    if ((threadIdx.x & 63) == 0) wave_bits[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) bits = wave_bits[w] > bits ? wave_bits[w] : bits;
        atomicMax(radius2_bits, bits);
        radius_sums[blockIdx.x] = ((wave_sums[0] + wave_sums[1]) + wave_sums[2]) + wave_sums[3];
    }
}

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
    l = __float_as_uint(r2);      // what is left has at most 8 significant bits: already a bf16 value
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

// Same contraction, counting instead of summing (RANSAC scoring).  The FP32 residual of a point
// lies within a known distance e of the reference-arithmetic one (the caller derives e per cloud),
// so two counts bracket the reference's inlier count of a hypothesis: points with |w.p| < thr - e
// are inliers for certain, points with |w.p| >= thr + e are not.  thr_tab[cloud] = (thr - e, thr + e),
// rounded outwards.  partial[0][split][candidate] = certain, partial[1][split][candidate] = possible.
// Padding points have residual 0 and are counted wherever 0 lies below the bound; the host
// subtracts them.  Lane counters are 32 bits wide: a lane sees 1/16 of a split's points.
template <int NA>
__global__ __launch_bounds__(256) void inlier_mfma_kernel(
    const float* __restrict__ W, const f32x4* __restrict__ groups,
    const int* __restrict__ group_off, unsigned long long* __restrict__ partial,
    const f32x2* __restrict__ thr_tab, int n_clouds, int n_ctiles, int n_splits)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ctile0 = (blockIdx.x * 4 + wave) * NA;
    const int split = blockIdx.y;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    unsigned int lo[NA][4], hi[NA][4];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) { lo[a][r] = 0; hi[a][r] = 0; }

    for (int j = 0; j < n_clouds; ++j) {
        const int g0 = group_off[j];
        const int G = group_off[j + 1] - g0;
        const int gb = g0 + (int)(((int64_t)G * split) / n_splits);
        const int ge = g0 + (int)(((int64_t)G * (split + 1)) / n_splits);
        if (gb >= ge) continue;
        const f32x2 t = thr_tab[j];
        const float tlo = t[0], thi = t[1];
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
                    for (int r = 0; r < 4; ++r) {
                        const float m = __builtin_fabsf(d[r]);
                        lo[a][r] += (m < tlo) ? 1u : 0u;
                        hi[a][r] += (m < thi) ? 1u : 0u;
                    }
                }
            }
            b = bn;
        }
    }
    const int64_t plane = (int64_t)n_splits * n_ctiles * 16;
    unsigned long long* out = partial + (int64_t)split * n_ctiles * 16;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            unsigned long long v = lo[a][r], w = hi[a][r];
            v += __shfl_xor(v, 1, 64); w += __shfl_xor(w, 1, 64);
            v += __shfl_xor(v, 2, 64); w += __shfl_xor(w, 2, 64);
            v += __shfl_xor(v, 4, 64); w += __shfl_xor(w, 4, 64);
            v += __shfl_xor(v, 8, 64); w += __shfl_xor(w, 8, 64);
            if ((lane & 15) == 0) {
                const int64_t at = (int64_t)(ctile0 + a) * 16 + (lane >> 4) * 4 + r;
                out[at] = v;
                out[plane + at] = w;
            }
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

// ---------------------------------------------------------------------------
// Separable grid search
// ---------------------------------------------------------------------------
// On a translation x rotation product grid the folded residual separates:
//   |w.(p,1)| = |u_i(r) + b_t|,   u_i(r) = w123(r,j).(p_i - c_j) + a(r,j),   b_t = b(t,j)
// (w123 depends on the rotation only, the constant is a rotation part plus a
// translation part; both are taken relative to a per-cloud reference so that
// they stay small).  For one rotation and cloud, f(b) = sum_i |u_i + b| is a
// piecewise-linear function of the scalar b, so all translations are answered
// from ONE pass over the points: with the b_t sorted, point i falls into bin
// k_i = #{t : b_t < -u_i}; per bin a count and the sum of u; then with prefix
// sums N(q), S(q) over bins <= q
//   f(b_q) = 2 S(q) - S_total + b_q (2 N(q) - N).
// Work per (rotation, point) is two LDS reads and one LDS atomic instead of one
// residual per translation.  Sums are integers (u in 32-bit fixed point,
// accumulated with 64-bit integer atomics), so the result does not depend on the
// order of the atomics.
//
// One workgroup = RB rotations x one cloud x one range of at most TC translations.

struct SepMeta {
    float inv_delta;   // buckets per metre
    float offset;      // -(lowest threshold) * inv_delta
};

// Bucket of a key among n_buckets equal slices of the thresholds' range; monotone in the key
// (one correctly rounded fma, then a clamp).  Host (table construction) and device (lookup)
// evaluate this same float expression.
__host__ __device__ inline int separable_bucket(float key, const SepMeta& mt, int n_buckets)
{
    float fb = __builtin_fmaf(key, mt.inv_delta, mt.offset);
    const float top = (float)(n_buckets - 1);
#if defined(__HIP_DEVICE_COMPILE__)
    fb = __builtin_amdgcn_fmed3f(fb, 0.f, top);
#else
    fb = fb < 0.f ? 0.f : (fb > top ? top : fb);
#endif
    return (int)fb;
}

// Lookup-table entry (16 bits): four times the rank of the bucket's first threshold -- the byte offset
// of that threshold in the sorted list, and half the byte offset of its histogram word -- with the top
// bit set when the bucket holds two thresholds or more.  At most 8191 thresholds per launch.
constexpr unsigned kSepCrowded = 0x8000u, kSepOffsetMask = 0x7ffcu;

// BPB = buckets of the lookup table per histogram bin: with several the typical bucket holds no
// threshold or one, so the rank of a key is the table entry plus one or two comparisons.
//
// Fixed point: u * fx_scale rounded to a 32-bit integer (fx_scale = a power of two chosen by the
// caller so that |u| * fx_scale < 2^31 over the whole grid; the conversion saturates, so a wrong
// bound can only saturate values).  At least 2^-30 of the largest |u|: finer than the float u itself
// for all but the smallest residuals.
//
// A bin is ONE 64-bit word in LDS: count in the top 16 bits, the sum of the biased values (fixed point
// + 2^31, below 2^32) in the low 48, so a point costs one LDS atomic.  The word holds 65535 points
// without a carry into the count; before a workgroup has put that many points into its words (every 31
// trips of the point loop at 2048 points per trip) it empties its words into wide (32-bit count, 64-bit sum) accumulators in global memory
// -- its own slice, touched a handful of times per cloud, L2-resident.  With 8 bytes of LDS per bin a
// workgroup can serve RB = 4 rotations from one pass over the cloud's points (the pass is bound by the
// rate at which the points come out of L2: 16 bytes per point and workgroup).
//
// The point loop works on all 4 points x RB rotations of a trip at once, phase by phase (values and
// table reads; threshold reads; the rare crowded buckets; atomics), so that the LDS reads of a trip
// are in flight together instead of one round trip per item.
//
// NT = threads per workgroup: 512 where two or three workgroups share a CU, 1024 for the widest form,
// whose tables fill most of a CU's LDS.
template <int TC, int RB, int BPB, int NT>
__global__ __launch_bounds__(NT) void separable_hist_kernel(
    const f32x4* __restrict__ groups, const int* __restrict__ group_off,
    const f32x4* __restrict__ wr,            // [rotation][cloud]: w1, w2, w3, a
    const float* __restrict__ sb,            // [cloud][nt]: sorted translation constants
    const int* __restrict__ perm,            // [cloud][nt]: rank -> local translation index
    const unsigned short* __restrict__ tab,  // [cloud][M + 1]: see kSepCrowded; entry [M] = 4 nt
    const SepMeta* __restrict__ meta,        // [cloud]
    const double* __restrict__ scale,        // [cloud]: 1 / (n_clouds * n_points)
    const int64_t* __restrict__ cloud_pt_off,
    int n_clouds, int nt, int nr, float fx_scale,
    unsigned int* __restrict__ wide_cnt, long long* __restrict__ wide_sum,   // [workgroup][RB][TC + 1]
    double* __restrict__ partial /* [cloud][nt][nr] */)
{
    constexpr int M = BPB * TC;
    constexpr int kBins = RB * (TC + 1);
    __shared__ unsigned long long pk[RB][TC + 1];
    __shared__ float sbs[TC + 1];          // [nt] = +infinity
    __shared__ unsigned short tabs[M + 2];
    __shared__ unsigned long long tot_c[RB];
    __shared__ long long tot_s[RB], tot_all_s[RB];
    const int j = blockIdx.y;
    const int r0 = blockIdx.x * RB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kWaves = NT / 64;
    constexpr int kSepFlushTrips = 65535 / (kWaves * 256);   // a trip = 256 points per wave
    static_assert(4 * TC <= (int)kSepOffsetMask, "ranks must fit the lookup-table entry");
    const int64_t wg = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    unsigned int* wcnt = wide_cnt + wg * kBins;
    long long* wsum = wide_sum + wg * kBins;

    // thresholds, rotation vectors and the bucket function in units of the fixed point (fx_scale is a power
    // of two: the scaled values are the same numbers, and a residual converts to its integer as it is)
    for (int i = threadIdx.x; i <= nt; i += NT) sbs[i] = i < nt ? sb[(int64_t)j * nt + i] * fx_scale : __builtin_inff();
    for (int i = threadIdx.x; i <= M; i += NT) tabs[i] = tab[(int64_t)j * (M + 1) + i];
    for (int i = threadIdx.x; i < kBins; i += NT) {
        (&pk[0][0])[i] = 0ull;
        wcnt[i] = 0u;          // thread t owns bins t, t + NT, ... in every pass over them
        wsum[i] = 0ll;
    }
    if (threadIdx.x < RB) tot_all_s[threadIdx.x] = 0ll;
    // rotation vectors in pairs: the residuals of two rotations of one point are one packed FP32 chain
    static_assert(RB % 2 == 0, "rotations are handled in pairs");
    f32x2 w0[RB / 2], w1[RB / 2], w2[RB / 2], w3[RB / 2];
#pragma unroll
    for (int rp = 0; rp < RB / 2; ++rp) {
        const int ra = r0 + 2 * rp < nr ? r0 + 2 * rp : nr - 1;       // a partial last group repeats its last rotation
        const int rc = r0 + 2 * rp + 1 < nr ? r0 + 2 * rp + 1 : nr - 1;
        const f32x4 a = wr[(int64_t)ra * n_clouds + j] * fx_scale, c = wr[(int64_t)rc * n_clouds + j] * fx_scale;
        w0[rp] = f32x2{a[0], c[0]}; w1[rp] = f32x2{a[1], c[1]}; w2[rp] = f32x2{a[2], c[2]}; w3[rp] = f32x2{a[3], c[3]};
    }
    SepMeta mt = meta[j];
    mt.inv_delta /= fx_scale;
    // The two outer bins (every point beyond the thresholds: most points, for rotations far from the
    // optimum) never touch LDS.  Their counts are wave-wide population counts (scalar registers); of the
    // sums a lane keeps the one of ALL its points and the one of the points below the first threshold --
    // the sum above the last threshold is what is left once the bins in between are known.
    unsigned c_lo[RB] = {};
    unsigned long long s_all[RB] = {}, s_lo[RB] = {};
    const unsigned top8 = 8u * (unsigned)nt;
    __syncthreads();

    auto flush = [&]() {
        // every bin by the thread that owns it: plain read-modify-write, no atomics
        for (int i = threadIdx.x; i < kBins; i += NT) {
            const unsigned long long p = (&pk[0][0])[i];
            if (p == 0ull) continue;
            const unsigned long long c = p >> 48;
            wcnt[i] += (unsigned int)c;
            wsum[i] += (long long)(p & 0xffffffffffffull) - (long long)(c << 31);
            (&pk[0][0])[i] = 0ull;
        }
    };

    // One trip of a lane: the four points of its column of group g, all RB rotations, one point (RB items)
    // at a time and phase by phase.  Every item reads the table and its bucket's first threshold (an empty
    // bucket's entry names the NEXT threshold, which lies in a higher bucket and so above the key: the
    // comparison adds nothing; one past the last threshold sits +infinity).  All offsets stay in bytes: the
    // table entry is the offset of the threshold, twice that the offset of the histogram word.
    // kCheck: the group may hold padding points.
    auto trip = [&](const f32x4* gp, auto check) {
        constexpr bool kCheck = decltype(check)::value;
        const f32x4 x4 = gp[0], y4 = gp[16], z4 = gp[32];
        f32x4 o4 = {1.f, 1.f, 1.f, 1.f};
        if (kCheck) o4 = gp[48];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v[RB], thr[RB];
            unsigned e[RB];
            // phase 1: residuals of rotation pairs (packed FP32, three roundings), table read
            const f32x2 xx = {x4[u], x4[u]}, yy = {y4[u], y4[u]}, zz = {z4[u], z4[u]};
#pragma unroll
            for (int rp = 0; rp < RB / 2; ++rp) {
                const f32x2 v2 = __builtin_elementwise_fma(w0[rp], xx, __builtin_elementwise_fma(w1[rp], yy,
                                 __builtin_elementwise_fma(w2[rp], zz, w3[rp])));
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int rb = 2 * rp + h;
                    v[rb] = v2[h];
                    // k = #{q : sbs[q] < -v}.  The bucket function is monotone and the table was built with
                    // this very expression: every threshold of a lower bucket is below the key, every threshold
                    // of a higher bucket above it, so k is the table entry (thresholds in lower buckets) plus
                    // the thresholds of the key's own bucket that lie below it.
                    e[rb] = tabs[separable_bucket(-v[rb], mt, M)];
                }
            }
            // phase 2: the first threshold that could still lie below the key
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                thr[rb] = *(const float*)((const char*)sbs + (e[rb] & kSepOffsetMask));
            unsigned k8[RB];
            unsigned long long crowded = 0ull;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const bool below = thr[rb] < -v[rb];
                k8[rb] = 2u * (e[rb] & kSepOffsetMask) + (below ? 8u : 0u);
                crowded |= __builtin_amdgcn_ballot_w64(below) & __builtin_amdgcn_ballot_w64(e[rb] >= kSepCrowded);
            }
            // phase 3 (rare): a bucket with several thresholds, the first of them below the key
            if (crowded != 0ull) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    if (!(thr[rb] < -v[rb] && e[rb] >= kSepCrowded)) continue;
                    unsigned hi = (tabs[separable_bucket(-v[rb], mt, M) + 1] & kSepOffsetMask) >> 2;
                    unsigned kk = k8[rb] >> 3;
                    while (kk < hi) {
                        const unsigned mid = (kk + hi) >> 1;
                        if (sbs[mid] < -v[rb]) kk = mid + 1; else hi = mid;
                    }
                    k8[rb] = kk << 3;
                }
            }
            // phase 4: outer bins in registers, the others in the histogram
            const bool valid = !kCheck || o4[u] != 0.f;                      // not a padding point
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const unsigned biased = (unsigned)(int)v[rb] ^ 0x80000000u;   // towards zero: below one unit
                const bool lo = valid && k8[rb] == 0u, hi = k8[rb] == top8;
                c_lo[rb] += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(lo));
                s_all[rb] += valid ? biased : 0u;
                if (lo) s_lo[rb] += biased;
                if (valid && !lo && !hi)
                    atomicAdd((unsigned long long*)((char*)&pk[rb][0] + k8[rb]), (1ull << 48) + (unsigned long long)biased);
            }
        }
    };

    // points: a wave takes four 64-point groups per trip, a lane the four points of one column; padding
    // points exist in a cloud's last group only
    const int g0 = group_off[j], g1 = group_off[j + 1];
    const int64_t n_points = cloud_pt_off[j + 1] - cloud_pt_off[j];
    const int col = lane & 15, gsel = lane >> 4;
    int trips = 0;
    for (int gb = g0; gb < g1; gb += kWaves * 4) {
        const int g = gb + wave * 4 + gsel;
        if (g < g1) {
            const f32x4* gp = groups + (int64_t)g * 64 + col;
            if (__builtin_amdgcn_ballot_w64(g == g1 - 1) != 0ull) trip(gp, std::true_type{});
            else trip(gp, std::false_type{});
        }
        if (++trips == kSepFlushTrips) {          // uniform over the workgroup
            trips = 0;
            __syncthreads();
            flush();
            __syncthreads();
        }
    }
    __syncthreads();
    flush();
    __syncthreads();
    // the outer bins: the count below the first threshold is already a wave total; the sums are reduced
    // over the wave, and one lane per wave adds (c_lo, s_lo) to bin 0 and s_all to the rotation's total
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        long long sl = (long long)s_lo[rb], sa = (long long)s_all[rb];
        for (int off = 32; off > 0; off >>= 1) {
            sl += __shfl_xor(sl, off, 64);
            sa += __shfl_xor(sa, off, 64);
        }
        if (lane == 0) {
            atomicAdd(&wcnt[rb * (TC + 1)], c_lo[rb]);
            atomicAdd((unsigned long long*)&wsum[rb * (TC + 1)], (unsigned long long)(sl - ((long long)c_lo[rb] << 31)));
            atomicAdd((unsigned long long*)&tot_all_s[rb], (unsigned long long)sa);
        }
    }
    __threadfence_block();
    __syncthreads();

    // inclusive prefix sums over the bins in place, one wave per rotation; the totals go to LDS
    for (int rb = wave; rb < RB; rb += kWaves) {
        unsigned long long cn = 0;
        long long cs = 0;
        for (int base = 0; base <= nt; base += 64) {
            const int kk = base + lane;
            unsigned long long c = 0;
            long long sv = 0;
            if (kk <= nt) {
                c = wcnt[rb * (TC + 1) + kk];
                sv = wsum[rb * (TC + 1) + kk];
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned long long c2 = __shfl_up(c, off, 64);
                const long long s2 = __shfl_up(sv, off, 64);
                if (lane >= off) { c += c2; sv += s2; }
            }
            c += cn;
            sv += cs;
            if (kk <= nt) {
                wcnt[rb * (TC + 1) + kk] = (unsigned int)c;
                wsum[rb * (TC + 1) + kk] = sv;
            }
            cn = __shfl(c, 63, 64);
            cs = __shfl(sv, 63, 64);
        }
        // the top bin is never stored: the count of all points is the cloud's size, and the sum of
        // everything is s_all less the bias of all points
        if (lane == 0) {
            const unsigned long long n_all = (unsigned long long)n_points;
            tot_c[rb] = n_all;
            tot_s[rb] = tot_all_s[rb] - (long long)(n_all << 31);
        }
    }
    __threadfence_block();
    __syncthreads();

    const double sc = scale[j];
    const double inv_fx = 1.0 / (double)fx_scale;
    for (int q = threadIdx.x; q < nt; q += NT) {
        const double bq = (double)sbs[q] * inv_fx;
        const int tl = perm[(int64_t)j * nt + q];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if (r0 + rb >= nr) break;
            const double n_le = (double)wcnt[rb * (TC + 1) + q], n_all = (double)tot_c[rb];
            const double s_le = (double)wsum[rb * (TC + 1) + q], s_all = (double)tot_s[rb];
            const double f = (2.0 * s_le - s_all) * inv_fx + bq * (2.0 * n_le - n_all);
            partial[((int64_t)j * nt + tl) * nr + (r0 + rb)] = f * sc;
        }
    }
}

// cost[(t_off + tl) * row + r_off + rl] = sum over clouds (in cloud order) of partial[cloud][tl][rl]
__global__ __launch_bounds__(256) void separable_reduce_kernel(const double* __restrict__ partial,
                                                                int n_clouds, int nt, int nr,
                                                                float* __restrict__ cost, int64_t row,
                                                                int64_t t_off, int64_t r_off)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nt * nr) return;
    double s = 0.0;
    for (int j = 0; j < n_clouds; ++j) s += partial[(int64_t)j * nt * nr + i];
    const int64_t tl = i / nr, rl = i % nr;
    cost[(t_off + tl) * row + r_off + rl] = (float)s;
}

// ---------------------------------------------------------------------------
// Screening pass: the folded FP32 objective of a few poses, one pass over the points
// ---------------------------------------------------------------------------
// sum_i |w.(p_i - c, 1)| per (pose, cloud) for up to a few dozen poses at once: every lane keeps
// 4 points in registers and walks the poses in tiles of kScreenTile (the four-vectors are uniform:
// scalar loads).  Roundings per residual: four (three fused multiply-adds and a product); per
// partial sum: 4 FP32 adds in the lane and 6 in the wave reduction; everything after that is
// FP64 in a fixed order.  The caller turns these counts into an error bound.
struct BlockDesc;
constexpr int kScreenTile = 16;
constexpr int kScreenQuarters = 4;      // workgroups per block of the table, see screen_cost_kernel

__global__ __launch_bounds__(256) void screen_cost_kernel(
    const f32x4* __restrict__ groups, const int* __restrict__ group_off, const int64_t* __restrict__ cloud_pt_off,
    const BlockDesc* __restrict__ blocks, const f32x4* __restrict__ w /* [pose][cloud] */, int n_poses,
    int n_clouds, double* __restrict__ partial /* [block * kScreenQuarters + quarter][pose] */);

// out[pose * n_clouds + cloud] = sum of the cloud's block partials; one wave per (pose, cloud): lane l adds
// the partials l, l + 64, ... in order, then the lanes combine in a fixed tree (the same bits on every run).
__global__ __launch_bounds__(256) void screen_reduce_kernel(const double* __restrict__ partial,
                                                             const int* __restrict__ cloud_block_off, int n_poses,
                                                             int n_clouds, double* __restrict__ out)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n_poses * n_clouds) return;
    const int p = i / n_clouds, j = i % n_clouds;
    double s = 0.0;
    for (int b = kScreenQuarters * cloud_block_off[j] + lane; b < kScreenQuarters * cloud_block_off[j + 1]; b += 64)
        s += partial[(int64_t)b * n_poses + p];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) out[i] = s;
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

// Totals over the 64 lanes of a wave of 16 values per lane at once: the 16 lanes of a row reduce-scatter
// (four exchange steps: a lane keeps one half of its values and adds the partner's copy of that half), then
// the four rows are added; lane l returns the total of value l & 15.  Six FP32 adds per total, as a
// butterfly per value would take, for a third of the instructions.
__device__ inline float wave_sum16(const float (&v)[16], int lane)
{
    const int col = lane & 15;
    float w8[8], w4[4], w2[2];
    {
        const bool hi = (col & 8) != 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float keep = hi ? v[8 + i] : v[i], give = hi ? v[i] : v[8 + i];
            w8[i] = keep + __shfl_xor(give, 8, 64);
        }
    }
    {
        const bool hi = (col & 4) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float keep = hi ? w8[4 + i] : w8[i], give = hi ? w8[i] : w8[4 + i];
            w4[i] = keep + __shfl_xor(give, 4, 64);
        }
    }
    {
        const bool hi = (col & 2) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float keep = hi ? w4[2 + i] : w4[i], give = hi ? w4[i] : w4[2 + i];
            w2[i] = keep + __shfl_xor(give, 2, 64);
        }
    }
    const bool hi = (col & 1) != 0;
    float r = (hi ? w2[1] : w2[0]) + __shfl_xor(hi ? w2[0] : w2[1], 1, 64);
    r += __shfl_xor(r, 16, 64);
    r += __shfl_xor(r, 32, 64);
    return r;
}

// A workgroup takes a quarter of a block of the table (kScreenQuarters per block: 1024 points, four per
// lane): with whole blocks the benchmark's 1465 workgroups were one and a half rounds of what the chip holds
// at once, and every round began with all of them waiting for their loads.

__global__ __launch_bounds__(256) void screen_cost_kernel(
    const f32x4* __restrict__ groups, const int* __restrict__ group_off, const int64_t* __restrict__ cloud_pt_off,
    const BlockDesc* __restrict__ blocks, const f32x4* __restrict__ w, int n_poses, int n_clouds,
    double* __restrict__ partial /* [block * kScreenQuarters + quarter][pose] */)
{
    __shared__ float red[4][kScreenTile];
    const BlockDesc bd = blocks[blockIdx.x / kScreenQuarters];
    const int quarter = blockIdx.x % kScreenQuarters;
    const int j = bd.cloud;
    // a block of the table is 4096 consecutive points of one cloud = 64 groups
    const int64_t lb = bd.begin - cloud_pt_off[j], le = bd.end - cloud_pt_off[j];
    const int g0 = group_off[j] + (int)(lb / kGroupPoints) + quarter * 16;
    const int g1 = group_off[j] + (int)((le + kGroupPoints - 1) / kGroupPoints);
    double* out = partial + (int64_t)blockIdx.x * n_poses;
    if (g0 >= g1) {                                   // a short last block: nothing in this quarter
        for (int p = threadIdx.x; p < n_poses; p += 256) out[p] = 0.0;
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, gsel = lane >> 4;
    // The fourth row of a group (1 for a point, 0 for padding) is not loaded: it follows from the point's index.
    const int64_t n_j = cloud_pt_off[j + 1] - cloud_pt_off[j];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 X = zero, Y = zero, Z = zero, O = zero;      // absent groups add |0|
    const int g = g0 + wave * 4 + gsel;
    if (g < g1) {
        const f32x4* gp = groups + (int64_t)g * 64 + col;
        X = gp[0]; Y = gp[16]; Z = gp[32];
        const int64_t first = (int64_t)(g - group_off[j]) * kGroupPoints + col;
#pragma unroll
        for (int u = 0; u < 4; ++u) O[u] = first + u * kTilePoints < n_j ? 1.f : 0.f;
    }
    for (int p0 = 0; p0 < n_poses; p0 += kScreenTile) {
        float acc[kScreenTile];
#pragma unroll
        for (int t = 0; t < kScreenTile; ++t) {
            const int p = p0 + t < n_poses ? p0 + t : n_poses - 1;   // a short last tile repeats its last pose
            const f32x4 wv = w[(int64_t)p * n_clouds + j];
            // two residuals per instruction (packed FP32); each still is the chain product, fma, fma, fma,
            // and the lane adds them in point order
            const f32x2 w0 = {wv[0], wv[0]}, w1 = {wv[1], wv[1]}, w2 = {wv[2], wv[2]}, w3 = {wv[3], wv[3]};
            float a = 0.f;
#pragma unroll
            for (int h = 0; h < 4; h += 2) {
                const f32x2 x = {X[h], X[h + 1]}, y = {Y[h], Y[h + 1]};
                const f32x2 z = {Z[h], Z[h + 1]}, o = {O[h], O[h + 1]};
                const f32x2 r = __builtin_elementwise_fma(w0, x, __builtin_elementwise_fma(w1, y,
                                __builtin_elementwise_fma(w2, z, w3 * o)));
                a += __builtin_fabsf(r[0]);
                a += __builtin_fabsf(r[1]);
            }
            acc[t] = a;
        }
        const float tot = wave_sum16(acc, lane);
        if (lane < kScreenTile) red[wave][lane] = tot;
        __syncthreads();
        if (threadIdx.x < kScreenTile && p0 + (int)threadIdx.x < n_poses) {
            const int t = threadIdx.x;
            out[p0 + t] = (((double)red[0][t] + (double)red[1][t]) + (double)red[2][t]) + (double)red[3][t];
        }
        __syncthreads();
    }
}

// Per (pose, cloud): rows 0..2 of M = base_T_flange_j * F(pose), 12 doubles.
// Per pose: plane a,b,c,d, sqrt(a^2+b^2+c^2), its rounded reciprocal and a flag
// (non-zero when the norm is in the range the reciprocal path covers), padded to 8.
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

// a / b, correctly rounded, for b with a precomputed y = RN(1/b): two
// Newton corrections with exact FMA residuals.  After the first, q1 is a
// faithful rounding of a/b; with y within half an ulp of 1/b the second then
// lands on RN(a/b) (Markstein, "Computation of elementary functions on the IBM
// RISC System/6000 processor", 1990, theorem on q' = RN(q + r*y)).  The residuals
// are exact only away from underflow and overflow, so callers take this path for
// b in [2^-300, 2^300] and a zero or in [2^-600, 2^600] and divide otherwise.
__device__ inline double divide_by_known(double a, double b, double y)
{
    const double q0 = a * y;
    const double r0 = __builtin_fma(-b, q0, a);
    const double q1 = __builtin_fma(r0, y, q0);
    const double r1 = __builtin_fma(-b, q1, a);
    return __builtin_fma(r1, y, q1);
}

// |a*qx + b*qy + c*qz + d| of exact_distance, before the division by the norm.
__device__ inline double exact_numerator(const double* M, const double* pl, float x, float y, float z)
{
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
    const double num = pl[0] * (double)q[0] + pl[1] * (double)q[1] + pl[2] * (double)q[2] + pl[3];
    return __builtin_fabs(num);
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

// Ordered sum of one chain of non-negative values: ((0 + d_0) + d_1) + ... in
// element order with a rounding after every add, exactly the reference's
// accumulation of a cloud's distances
// (src/camera_calibration_optimization.cpp:305; floating-point addition is not
// associative, so a tree reduction would give other bits) -- computed by one
// workgroup per chain as an exact parallel emulation of the sequential sum:
//
//   While the running sum s stays inside one binade [2^e, 2^(e+1)) every add
//   rounds to a multiple of u = 2^(e-52).  With S = s/u (an integer in
//   [2^52, 2^53)) and t = d/u = q + r (q integer, 0 <= r < 1), round-to-nearest-
//   even gives  S' = S + q + [r > 1/2]  and, only for an exact tie r == 1/2,
//   S' = S + q + ((S + q) odd).  So an element is a map  parity(S) -> increment,
//   and maps compose associatively:  (A then B)[p] = A[p] + B[(p + A[p]) & 1].
//
//   A pass covers W*512 consecutive elements: each lane folds its run of 8
//   consecutive elements into a pair (inc[0], inc[1]), a wave forms the inclusive
//   prefix composition of its 64 lane maps (one 512-element segment), and the
//   first lanes of wave 0 compose the W segment maps and apply them to the
//   actual S.  A prefix is valid iff S + increment < 2^53, which (all increments
//   being >= 0) holds exactly when no add inside it left the binade.  At the
//   first invalid prefix the owning wave finds the first lane whose run leaves
//   the binade from its lane prefixes, that lane adds its 8 elements one by one
//   in plain FP64 from the exact sum in front of it, and the rest of the pass is
//   folded again with the new binade's unit (about once per binade, ~20 times
//   per chain, most of them in the first pass where s doubles every few adds).
//
// The elements come from a source functor: the point-to-plane distances of one
// (pose, cloud) pair computed on the fly from the SoA coordinates, or a plain
// array.  Passes start at the chain's begin rounded down to a multiple of 8 so
// that every lane's run is 32-byte aligned; elements outside [begin, end) count
// as +0.0, which adds nothing.
constexpr int kChainRun = 8;                       // consecutive elements per lane
constexpr int kChainSeg = 64 * kChainRun;          // elements per wave segment

// IEEE maxNum of two doubles as the single hardware instruction (a NaN operand
// yields the other one).
__device__ inline double max_f64(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

struct DistanceSource {
    const float* xs; const float* ys; const float* zs;
    int64_t limit;                                 // elements readable in xs/ys/zs
    double M[12];
    double pl[7];                                  // a, b, c, d, |n|, RN(1/|n|), range flag
    struct Raw { float x[kChainRun], y[kChainRun], z[kChainRun]; };

    __device__ inline Raw fetch(int64_t g0) const
    {
        Raw r;
        if (g0 + kChainRun <= limit) {
#pragma unroll
            for (int h = 0; h < kChainRun / 4; ++h) {
                const float4 a = *reinterpret_cast<const float4*>(xs + g0 + 4 * h);
                const float4 b = *reinterpret_cast<const float4*>(ys + g0 + 4 * h);
                const float4 c = *reinterpret_cast<const float4*>(zs + g0 + 4 * h);
                r.x[4 * h] = a.x; r.x[4 * h + 1] = a.y; r.x[4 * h + 2] = a.z; r.x[4 * h + 3] = a.w;
                r.y[4 * h] = b.x; r.y[4 * h + 1] = b.y; r.y[4 * h + 2] = b.z; r.y[4 * h + 3] = b.w;
                r.z[4 * h] = c.x; r.z[4 * h + 1] = c.y; r.z[4 * h + 2] = c.z; r.z[4 * h + 3] = c.w;
            }
        } else {
#pragma unroll
            for (int t = 0; t < kChainRun; ++t) {
                const bool in = g0 + t < limit;
                r.x[t] = in ? xs[g0 + t] : 0.f; r.y[t] = in ? ys[g0 + t] : 0.f; r.z[t] = in ? zs[g0 + t] : 0.f;
            }
        }
        return r;
    }
    // The run's distances; returns false if one of them is not an ordinary finite
    // non-negative number.  pl[5] = RN(1/pl[4]) and pl[6] != 0 when pl[4] is in the
    // range divide_by_known covers; numerators outside its range (zero included) send
    // the whole run through the plain division.
    __device__ inline bool values(const Raw& r, double d[kChainRun]) const
    {
        double a[kChainRun];
        numerators(r, a);
        return quotients(a, d);
    }
    // The two halves of values(): |a qx + b qy + c qz + d| of the run, and its division by the norm.
    __device__ inline void numerators(const Raw& r, double a[kChainRun]) const
    {
#pragma unroll
        for (int t = 0; t < kChainRun; ++t) a[t] = exact_numerator(M, pl, r.x[t], r.y[t], r.z[t]);
    }
    __device__ inline bool quotients(const double a[kChainRun], double d[kChainRun]) const
    {
        constexpr unsigned kLo = 0x1a700000u, kHi = 0x65700000u;   // high words of 2^-600, 2^600
        unsigned hmin = 0xffffffffu, hmax = 0u;
#pragma unroll
        for (int t = 0; t < kChainRun; ++t) {
            const unsigned hi = (unsigned)((unsigned long long)__double_as_longlong(a[t]) >> 32);
            hmin = hi < hmin ? hi : hmin;
            hmax = hi > hmax ? hi : hmax;
            d[t] = divide_by_known(a[t], pl[4], pl[5]);
        }
        bool finite = true;
        if (__builtin_expect(!(pl[6] != 0.0 && hmin >= kLo && hmax < kHi), 0)) {
#pragma unroll
            for (int t = 0; t < kChainRun; ++t) {
                d[t] = a[t] / pl[4];
                finite = finite && d[t] >= 0.0 && d[t] < __builtin_inf();
            }
        }
        return finite;
    }
};

struct ArraySource {
    const double* v;
    int64_t limit;
    struct Raw { double d[kChainRun]; };
    __device__ inline Raw fetch(int64_t g0) const
    {
        Raw r;
#pragma unroll
        for (int t = 0; t < kChainRun; ++t) r.d[t] = g0 + t < limit ? v[g0 + t] : 0.0;
        return r;
    }
    __device__ inline bool values(const Raw& r, double d[kChainRun]) const
    {
        bool finite = true;
#pragma unroll
        for (int t = 0; t < kChainRun; ++t) {
            d[t] = r.d[t];
            finite = finite && d[t] >= 0.0 && d[t] < __builtin_inf();
        }
        return finite;
    }
};

// a = (a then b) for parity maps
__device__ inline void chain_compose(unsigned long long& a0, unsigned long long& a1,
                                     unsigned long long b0, unsigned long long b1)
{
    const unsigned long long n0 = a0 + ((a0 & 1ull) ? b1 : b0);
    const unsigned long long n1 = a1 + (((a1 + 1ull) & 1ull) ? b1 : b0);
    a0 = n0;
    a1 = n1;
}

// A 64-bit value moved between lanes by the data-parallel-primitive path of the vector unit (no LDS round
// trip, which is what a ds_bpermute shuffle costs a lone wave: ~100 cycles each).  Lanes without a source
// (the first lanes of a row for a row shift, the rows the mask leaves out) receive 0.  Every lane of the wave
// must be active.
template <int kCtrl, int kRowMask>
__device__ inline unsigned long long dpp_move_u64(unsigned long long v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, kCtrl, kRowMask, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), kCtrl, kRowMask, 0xf, true);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
}
constexpr int kDppRowShr = 0x110;        // + n: lane l of a row of 16 reads lane l - n of the same row
constexpr int kDppRowBcast15 = 0x142;    // every lane of a row reads lane 15 of the row before
constexpr int kDppRowBcast31 = 0x143;    // rows 2 and 3 read lane 31
constexpr int kDppWaveShr1 = 0x138;      // lane l reads lane l - 1 of the wave

// v of lane `from`, the same lane for the whole wave (a scalar register index: v_readlane, no LDS).
__device__ inline unsigned long long read_lane_u64(unsigned long long v, int from)
{
    const int f = __builtin_amdgcn_readfirstlane(from);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, f);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), f);
    return ((unsigned long long)hi << 32) | lo;
}

// Inclusive prefix composition over the first N lanes of a wave (lower lane first): shifts by 1, 2, 4, 8
// inside the rows of 16, then the row totals are carried into the rows behind them.  The map (0, 0) a lane
// without a source receives is the identity of the composition, so no lane needs to be excluded.
template <int kCtrl, int kRowMask>
__device__ inline void chain_scan_step(unsigned long long& i0, unsigned long long& i1)
{
    unsigned long long l0 = dpp_move_u64<kCtrl, kRowMask>(i0), l1 = dpp_move_u64<kCtrl, kRowMask>(i1);
    chain_compose(l0, l1, i0, i1);
    i0 = l0;
    i1 = l1;
}

// Sum of v over the 64 lanes of a wave (every lane active), the same value returned to all of them: prefix
// sums by the same six moves, the total read back from lane 63.
__device__ inline unsigned long long wave_sum_u64(unsigned long long v)
{
    v += dpp_move_u64<kDppRowShr + 1, 0xf>(v);
    v += dpp_move_u64<kDppRowShr + 2, 0xf>(v);
    v += dpp_move_u64<kDppRowShr + 4, 0xf>(v);
    v += dpp_move_u64<kDppRowShr + 8, 0xf>(v);
    v += dpp_move_u64<kDppRowBcast15, 0xa>(v);
    v += dpp_move_u64<kDppRowBcast31, 0xc>(v);
    return read_lane_u64(v, 63);
}

template <int N>
__device__ inline void chain_scan(unsigned long long& i0, unsigned long long& i1, int /*lane*/)
{
    if (N > 1) chain_scan_step<kDppRowShr + 1, 0xf>(i0, i1);
    if (N > 2) chain_scan_step<kDppRowShr + 2, 0xf>(i0, i1);
    if (N > 4) chain_scan_step<kDppRowShr + 4, 0xf>(i0, i1);
    if (N > 8) chain_scan_step<kDppRowShr + 8, 0xf>(i0, i1);
    if (N > 16) chain_scan_step<kDppRowBcast15, 0xa>(i0, i1);
    if (N > 32) chain_scan_step<kDppRowBcast31, 0xc>(i0, i1);
}

// Composition of the 64 lane maps of a wave in lane order; valid in lane 0.
__device__ inline void chain_compose_wave(unsigned long long& i0, unsigned long long& i1)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long r0 = __shfl_down(i0, off, 64);
        const unsigned long long r1 = __shfl_down(i1, off, 64);
        chain_compose(i0, i1, r0, r1);
    }
}

// Map of one lane's run for the unit 1/scale (every element below 2^52 units).
// n = RN-even(t) comes from adding 2^52 and reading the significand.  An element
// that is not a tie adds n whatever the parity, so a run without ties is just the
// sum of its n: one integer add per element, and `tie` stays false.  A tie
// t = q + 1/2 has n = the even one of {q, q+1} and adds q + ((S + q) odd)
// = n + parity(S) (t > n) or n - parity(S) (t < n); such runs are folded again
// with both parities carried.
__device__ inline void chain_fold_run(const double d[kChainRun], double scale, unsigned long long& inc0,
                                      unsigned long long& inc1, bool& tie)
{
    const double two52 = 4503599627370496.0;
    constexpr unsigned long long kBits52 = 0x4330000000000000ull;   // bit pattern of 2^52
    unsigned long long sum = 0;
    tie = false;
#pragma unroll
    for (int t = 0; t < kChainRun; ++t) {
        const double tt = d[t] * scale;             // exact (power of two)
        const double tm = tt + two52;               // in [2^52, 2^53]: the integer nearest to tt, ties to even
        sum += (unsigned long long)__double_as_longlong(tm);
        const double diff = tt - (tm - two52);      // both exact; in [-1/2, 1/2]
        tie = tie || __builtin_fabs(diff) == 0.5;
    }
    sum -= (unsigned long long)kChainRun * kBits52;
    inc0 = sum;
    inc1 = sum;
    if (__builtin_expect(tie, 0)) {
        // keep this rare path's arithmetic out of the common one (no shared subexpressions)
        asm volatile("" : "+v"(scale));
        inc0 = 0; inc1 = 0;
#pragma unroll
        for (int t = 0; t < kChainRun; ++t) {
            const double tt = d[t] * scale;
            const double tm = tt + two52;
            const unsigned long long n = (unsigned long long)__double_as_longlong(tm) - kBits52;
            const double diff = tt - (tm - two52);
            const bool is_tie = __builtin_fabs(diff) == 0.5;
            const unsigned long long sgn = diff > 0.0 ? 1ull : ~0ull;
            inc0 += n + ((is_tie && (inc0 & 1ull)) ? sgn : 0ull);
            inc1 += n + ((is_tie && ((inc1 + 1ull) & 1ull)) ? sgn : 0ull);
        }
    }
}

// The same map from the numerators, without dividing, for the runs in which that is certain to agree.
// The reference's element is d = RN(a / b) (src/camera_calibration_optimization.cpp:121-129) and the fold
// wants n = RN-even(d * scale).  With y = RN(1/b) and ys = y * scale (exact: scale is a power of two and the
// caller has checked that ys is an ordinary number), t = RN(a * ys) lies within 3 * 2^-53 * t of d * scale
// (one rounding in d, one in y, one in t), so whenever t is farther than thr = 2^-51 * t (+ 2^-60 for
// results in the subnormal range) from every half-integer, d * scale and t round to the same integer and
// d * scale is not a tie.  The distance of t from the half-integers is 1/2 - |t - n|, which the fold
// computes anyway.  half_lo = 1/2 - thr for the run's largest element (the caller has checked that it is
// below 2^51); `unsure` comes back true if an element is closer than that, is not a number or is infinite
// -- such runs are divided and folded by chain_fold_run.  Six instructions per element fewer (the
// division's five, the range test and the scaling give way to one product).
__device__ inline void chain_fold_run_quick(const double a[kChainRun], double ys, double half_lo,
                                            unsigned long long& inc, bool& unsure)
{
    const double two52 = 4503599627370496.0;
    constexpr unsigned long long kBits52 = 0x4330000000000000ull;
    unsigned long long sum = 0;
    unsure = false;
#pragma unroll
    for (int t = 0; t < kChainRun; ++t) {
        const double tt = a[t] * ys;
        const double tm = tt + two52;
        sum += (unsigned long long)__double_as_longlong(tm);
        const double diff = tt - (tm - two52);
        unsure = unsure || !(__builtin_fabs(diff) < half_lo);     // written so that a NaN is unsure
    }
    inc = sum - (unsigned long long)kChainRun * kBits52;
}

// Sum (ordered, continuing from s0) and, if kWantMax, maximum (:306-307, order-free) of elements
// [b, e) of src.  Every thread of the W-wave workgroup returns the same sum; the
// maximum is valid in thread 0.
template <int W, bool kWantMax, typename Source>
__device__ inline void chain_reduce(const Source& src, int64_t b, int64_t e, double s0, double& out_sum,
                                    double& out_max, double* seg_out = nullptr)
{
    static_assert(W >= 1 && W <= 64 && (W & (W - 1)) == 0, "W must be a power of two");
    constexpr int kPass = kChainSeg * W;
    constexpr unsigned long long kTwo53 = 1ull << 53;
    constexpr unsigned long long kMantissa = (1ull << 52) - 1;
    const double two52 = 4503599627370496.0;
    __shared__ unsigned long long sh_inc[W][2];
    __shared__ int sh_ok[W];
    __shared__ int sh_tie[W];
    __shared__ unsigned long long sh_S;
    __shared__ int sh_adv;
    __shared__ double sh_s;
    __shared__ int sh_consumed;
    __shared__ double sh_red[W];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int my = wave * kChainSeg + lane * kChainRun;   // this lane's offset inside a pass
    const int64_t b0 = b & ~(int64_t)(kChainRun - 1);

    double s = s0, mx = -1e9;                             // :288 error_mx = -1e9
    typename Source::Raw raw = src.fetch(b0 + my);

    for (int64_t i = b0; i < e; i += kPass) {
        const int64_t left = e - i;
        const int nelem = left >= kPass ? kPass : (int)left;
        const int nseg = (nelem + kChainSeg - 1) / kChainSeg;
        // the sum in front of this pass, left as a rough value of the sum at each of its segment
        // starts (the split evaluation guesses its units from these; the end of the chain follows)
        if (seg_out && (int)threadIdx.x < nseg) seg_out[(i - b0) / kChainSeg + threadIdx.x] = s;
        double d[kChainRun];
        bool finite = src.values(raw, d);
        if (!(i >= b && i + kPass <= e)) {
            // first / last pass: elements outside [b, e) count as +0.0
            finite = true;
#pragma unroll
            for (int t = 0; t < kChainRun; ++t) {
                const int64_t g = i + my + t;
                const bool in = g >= b && g < e;
                d[t] = in ? d[t] : 0.0;
                finite = finite && d[t] >= 0.0 && d[t] < __builtin_inf();
                if (kWantMax && in) mx = d[t] > mx ? d[t] : mx;
            }
        }
        double run_max = d[0];
#pragma unroll
        for (int t = 1; t < kChainRun; ++t) run_max = max_f64(run_max, d[t]);
        if (kWantMax && i >= b && i + kPass <= e) mx = max_f64(mx, run_max);
        // next pass in flight while this one is folded
        if (i + kPass < e) raw = src.fetch(i + kPass + my);

        int consumed = 0;                                 // elements of this pass already in s
        for (;;) {
            const unsigned long long sbits = (unsigned long long)__double_as_longlong(s);
            const int ebiased = (int)(sbits >> 52);       // sign included: negative sums are not usable
            const bool usable = ebiased >= 64 && ebiased <= 2000;
            if (!usable) {
                // s is zero (or out of the range the units cover): the next run one by one
                if (my == consumed) {
                    double sv = s;
#pragma unroll
                    for (int t = 0; t < kChainRun; ++t) sv += d[t];
                    sh_s = sv;
                }
                __syncthreads();
                s = sh_s;
                consumed += kChainRun;
                __syncthreads();
                if (consumed >= nelem) break;
                continue;
            }
            // scale = 1/u = 2^(52 - e): exponent field 1023 + 52 - (ebiased - 1023)
            const double scale = __longlong_as_double((long long)((unsigned long long)(2098 - ebiased) << 52));
            // an integer count of units in [2^52, 2^53) back to a double of this binade: exact
            auto from_units = [ebiased](unsigned long long n) {
                return __longlong_as_double((long long)((n & kMantissa) | ((unsigned long long)ebiased << 52)));
            };
            // this lane's map; consumed lanes are the identity
            unsigned long long inc0 = 0, inc1 = 0;
            bool ok = true, tie = false;
            const bool folding = (wave + 1) * kChainSeg > consumed && wave < nseg;
            if (folding) {
                if (my >= consumed) {
                    ok = finite && run_max * scale < two52;
                    if (ok) chain_fold_run(d, scale, inc0, inc1, tie);
                }
                const bool wave_tie = __any(tie ? 1 : 0);
                if (!wave_tie) {
                    // no ties in the segment: both parities add the plain sum of the lane totals,
                    // formed by the LDS adder (a wave's LDS operations execute in order)
                    if (W == 1) {
                        // a lone wave (the finishing pass) waits for every LDS round trip: sum in registers
                        const unsigned long long tot = wave_sum_u64(inc0);
                        if (lane == 0) sh_inc[wave][0] = tot;
                    } else {
                        // several waves per SIMD hide the round trip, and the LDS adder is one instruction
                        // (measured with the register sum here too: whole chains 1.7 % slower)
                        if (lane == 0) sh_inc[wave][0] = 0ull;
                        atomicAdd(&sh_inc[wave][0], inc0);
                    }
                } else {
                    unsigned long long t0 = inc0, t1 = inc1;
                    chain_compose_wave(t0, t1);
                    if (lane == 0) { sh_inc[wave][0] = t0; sh_inc[wave][1] = t1; }
                }
                const bool all_ok = __all(ok ? 1 : 0);
                if (lane == 0) {
                    sh_ok[wave] = all_ok ? 1 : 0;
                    sh_tie[wave] = wave_tie ? 1 : 0;
                }
            }
            __syncthreads();
            if (wave == 0) {
                const int first_seg = consumed / kChainSeg;
                // S = s / unit, an integer in [2^52, 2^53): the significand of s with its hidden bit
                const unsigned long long S = (sbits & kMantissa) | (1ull << 52);
                unsigned long long a0 = 0, a1 = 0;
                bool seg_ok = true;
                if (lane >= first_seg && lane < nseg) {
                    a0 = sh_inc[lane][0];
                    a1 = sh_tie[lane] ? sh_inc[lane][1] : a0;
                    // totals of 2^53 or more are rejected whatever their value: clamp so that
                    // the composition over segments cannot wrap
                    a0 = a0 < (kTwo53 << 1) ? a0 : (kTwo53 << 1);
                    a1 = a1 < (kTwo53 << 1) ? a1 : (kTwo53 << 1);
                    seg_ok = sh_ok[lane] != 0;
                }
                chain_scan<W>(a0, a1, lane);
                const unsigned long long tot = (S & 1ull) ? a1 : a0;
                const bool bad = lane < nseg && (!seg_ok || S + tot >= kTwo53);
                const unsigned long long mask = __ballot(bad ? 1 : 0);
                const int adv = mask ? __builtin_ctzll(mask) : nseg;
                const unsigned long long before = read_lane_u64(tot, adv > 0 ? adv - 1 : 0);
                if (lane == 0) {
                    sh_S = adv > 0 ? S + before : S;
                    sh_adv = adv;
                }
            }
            __syncthreads();
            const int adv = sh_adv;
            const unsigned long long Sx = sh_S;
            if (adv >= nseg) {
                s = from_units(Sx);
                break;
            }
            if (wave == adv) {
                // the run that leaves the binade (or cannot be folded): from the lane prefixes
                chain_scan<64>(inc0, inc1, lane);
                const unsigned long long tot = (Sx & 1ull) ? inc1 : inc0;
                const bool bad = !ok || Sx + tot >= kTwo53;
                const unsigned long long mask = __ballot(bad ? 1 : 0);
                const int L = mask ? __builtin_ctzll(mask) : 63;             // mask is never empty here
                const unsigned long long before = read_lane_u64(tot, L > 0 ? L - 1 : 0);
                if (lane == L) {
                    double sv = from_units(L > 0 ? Sx + before : Sx);
#pragma unroll
                    for (int t = 0; t < kChainRun; ++t) sv += d[t];
                    sh_s = sv;
                    sh_consumed = adv * kChainSeg + (L + 1) * kChainRun;
                }
            }
            __syncthreads();
            s = sh_s;
            consumed = sh_consumed;
            if (consumed >= nelem) break;
        }
    }
    if (kWantMax) {
        for (int off = 32; off > 0; off >>= 1) {
            const double om = __shfl_xor(mx, off, 64);
            mx = om > mx ? om : mx;
        }
        if (lane == 0) sh_red[wave] = mx;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int w = 1; w < W; ++w) mx = sh_red[w] > mx ? sh_red[w] : mx;
    }
    if (seg_out && threadIdx.x == 0) seg_out[(e - b0 + kChainSeg - 1) / kChainSeg] = s;
    out_sum = s;
    out_max = mx;
}

// Exact objective: one workgroup per (cloud j, pose k) chain; blocks of one cloud
// are adjacent so that the poses of a batch stream the same coordinates through
// the L2 caches together.  Results are stored pose-major: out[k * n_clouds + j];
// out_max may be null when kWantMax is false.  W waves per workgroup: few chains want
// many waves each (latency), many chains want small workgroups that overlap each
// other's barriers and segment walks (throughput).  Five waves per SIMD (96 VGPRs, no
// spills) measured 4 % faster than four; six spill.
template <int W, bool kWantMax>
__global__ __launch_bounds__(64 * W, 5) void exact_chain_kernel(
    const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
    const int64_t* __restrict__ cloud_pt_off, const double* __restrict__ mats,
    const double* __restrict__ planes, int n_poses, int n_clouds, int64_t total_pts,
    double* __restrict__ out_sum, double* __restrict__ out_max,
    const int64_t* __restrict__ cloud_seg_off, double* __restrict__ seg_sums /* null: not wanted */)
{
    const int j = blockIdx.x / n_poses, k = blockIdx.x % n_poses;
    DistanceSource src;
    src.xs = xs; src.ys = ys; src.zs = zs;
    src.limit = total_pts;
    const double* M = mats + ((int64_t)k * n_clouds + j) * 12;
    const double* pl = planes + (int64_t)k * kPlaneStride;
#pragma unroll
    for (int i = 0; i < 12; ++i) src.M[i] = M[i];
#pragma unroll
    for (int i = 0; i < 7; ++i) src.pl[i] = pl[i];
    double sum, mx;
    // the first pose's chains leave their running sums behind for a later split evaluation
    chain_reduce<W, kWantMax>(src, cloud_pt_off[j], cloud_pt_off[j + 1], 0.0, sum, mx,
                              seg_sums && k == 0 ? seg_sums + cloud_seg_off[j] : nullptr);
    if (threadIdx.x == 0) {
        out_sum[(int64_t)k * n_clouds + j] = sum;
        if (kWantMax) out_max[(int64_t)k * n_clouds + j] = mx;
    }
}

// ---------------------------------------------------------------------------
// One chain over several workgroups
// ---------------------------------------------------------------------------
// A request of a few dozen poses is a few hundred chains: too few, and too coarse, to fill 256 compute
// units evenly.  The sequential sum only looks indivisible: an element's map depends on the running sum
// through its binade (the unit) and its parity alone, and the binade at any place of a chain hardly moves
// from one evaluation to the next (a descent's poses are close).  So a chain is cut into K parts:
//   * part 0 starts from zero and is summed as before (most binade crossings are there);
//   * every other part folds its 512-element segments into maps WITHOUT knowing the sum in front of it,
//     in the unit that the previous evaluation's sum at that place suggests -- and in the next unit too
//     where that sum is close to a power of two -- one (unit, map) record per segment;
//   * a last, short kernel (one wave per chain) walks the records with the true sum: a record is used only
//     if its unit is the sum's actual binade and the sum stays inside it (sum + increment < 2^53, the very
//     test of the single-workgroup kernel); any other segment -- the one where the sum leaves its binade, a
//     wrong guess -- is summed again from the true sum by the sequential routine.  It also leaves the sums
//     at the segment starts behind for the next evaluation.
// The result is the sequential sum whatever the guesses were; they only decide how much is summed twice.
struct SegMap {
    unsigned long long inc[2][2];   // [unit: the guessed binade, the one above][parity of the sum]
    int ebiased;                    // biased exponent of the guessed binade
    unsigned int flags;             // 1: first map usable, 2: second map present, 4: second map usable
    unsigned long long pad;
};

constexpr int kSplitWaves = 4;      // waves per workgroup of the part kernel

// segments of a chain's part: the chain's segment count divided by K, rounded up to whole passes
__device__ __host__ inline int split_part_segments(int n_seg, int K)
{
    const int per = (n_seg + K - 1) / K;
    return (per + kSplitWaves - 1) / kSplitWaves * kSplitWaves;
}

// Five waves per SIMD (96 registers, one spilled): since the parts fold without dividing the loop fits, and the
// fifth wave is worth 1.2 % (with the division in the loop it spilled 67 registers at five waves and ran 8 %
// slower than at four); six waves spill 127.
__global__ __launch_bounds__(64 * kSplitWaves, 5) void split_chain_kernel(
    const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
    const int64_t* __restrict__ cloud_pt_off, const int64_t* __restrict__ cloud_seg_off,
    const double* __restrict__ mats, const double* __restrict__ planes, int n_poses, int n_clouds,
    int64_t total_pts, int K, const double* __restrict__ sref /* sums at segment starts, previous evaluation */,
    int64_t total_segs, SegMap* __restrict__ maps /* [pose][segment] */, double* __restrict__ first_sum)
{
    // Part-major order: consecutive workgroups go to the 8 XCDs in turn, and the workgroups of part 0
    // (the sequential routine, slower than the others) must not all land on the same ones; they also
    // start first.
    const int n_chains = n_poses * n_clouds;
    const int chain = blockIdx.x % n_chains, c = blockIdx.x / n_chains;
    const int j = chain / n_poses, k = chain % n_poses;
    const int64_t b = cloud_pt_off[j], e = cloud_pt_off[j + 1];
    const int64_t b0 = b & ~(int64_t)(kChainRun - 1);
    const int n_seg = (int)((e - b0 + kChainSeg - 1) / kChainSeg);
    const int per = split_part_segments(n_seg, K);
    const int s_begin = c * per;
    if (s_begin >= n_seg) return;
    const int s_end = s_begin + per < n_seg ? s_begin + per : n_seg;

    DistanceSource src;
    src.xs = xs; src.ys = ys; src.zs = zs;
    src.limit = total_pts;
    const double* M = mats + ((int64_t)k * n_clouds + j) * 12;
    const double* pl = planes + (int64_t)k * kPlaneStride;
#pragma unroll
    for (int i = 0; i < 12; ++i) src.M[i] = M[i];
#pragma unroll
    for (int i = 0; i < 7; ++i) src.pl[i] = pl[i];

    if (c == 0) {
        const int64_t e0 = b0 + (int64_t)s_end * kChainSeg;
        double sum, mx;
        chain_reduce<kSplitWaves, false>(src, b, e0 < e ? e0 : e, 0.0, sum, mx);
        if (threadIdx.x == 0) first_sum[(int64_t)k * n_clouds + j] = sum;
        return;
    }

    const double two52 = 4503599627370496.0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* ref = sref + cloud_seg_off[j];
    SegMap* out = maps + ((int64_t)k * total_segs + cloud_seg_off[j]);
    DistanceSource::Raw raw = src.fetch(b0 + (int64_t)(s_begin + wave) * kChainSeg + lane * kChainRun);
    for (int seg = s_begin + wave; seg < s_end; seg += kSplitWaves) {
        const int64_t g0 = b0 + (int64_t)seg * kChainSeg + lane * kChainRun;
        double a[kChainRun];
        src.numerators(raw, a);
        // The next segment's coordinates are in flight while this one is folded.  The numerators must be
        // complete before the loads are issued (the empty statement pins them here): computed after the
        // loads they would read a copy of the coordinates, a second set of 24 registers filled and emptied
        // by 48 moves per trip -- a tenth of the loop's instructions (if not of its time: 0.2 %).
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        static_assert(kChainRun == 8, "the statement above names every numerator of a run");
        if (seg + kSplitWaves < s_end) raw = src.fetch(g0 + (int64_t)kSplitWaves * kChainSeg);
        const bool past_end = g0 + kChainRun > e;     // the chain's last run(s): elements past the end count as +0.0

        // the previous evaluation's sums at this segment's ends, widened by 1/64 either way
        const double lo = ref[seg] * (1.0 - 0.015625), hi = ref[seg + 1] * (1.0 + 0.015625);
        const int eb_lo = (int)((unsigned long long)__double_as_longlong(lo) >> 52);
        const int eb_hi = (int)((unsigned long long)__double_as_longlong(hi) >> 52);
        const bool guess = eb_lo >= 64 && eb_lo <= 1999;      // sign bit set or out of range: no map
        const bool dual = guess && eb_hi != eb_lo;
        const double scale_a = __longlong_as_double((long long)((unsigned long long)(2098 - (guess ? eb_lo : 1023)) << 52));
        const double scale_b = scale_a * 0.5;
        unsigned long long a0 = 0, a1 = 0, c0 = 0, c1 = 0;
        bool ok_a = false, ok_b = false;

        // First without dividing (chain_fold_run_quick): needs the reciprocal of the norm (pl[6]), the
        // run's largest element below 2^51 units and every element clear of the rounding boundaries.
        bool quick = guess;
        if (guess) {
            // reciprocal of the norm in units of the finer guess; its scaling must not leave the normal range
            const double ys = src.pl[5] * scale_a;
            bool unsure = !(src.pl[6] != 0.0 && ys > 0x1p-1000 && ys < 0x1p1000);
            if (!unsure) {
                if (past_end) {
#pragma unroll
                    for (int t = 0; t < kChainRun; ++t) a[t] = g0 + t < e ? a[t] : 0.0;
                }
                double amax = a[0];
#pragma unroll
                for (int t = 1; t < kChainRun; ++t) amax = max_f64(amax, a[t]);
                const double tmax = amax * ys;
                unsure = !(tmax < 2251799813685248.0);                     // 2^51; a NaN is unsure
                const double thr = tmax * 0x1p-51 + 0x1p-60;
                bool u1 = false, u2 = false;
                chain_fold_run_quick(a, ys, 0.5 - thr, a0, u1);
                // the coarser unit: half the values, half the distance (the subnormal allowance stays)
                if (dual) chain_fold_run_quick(a, ys * 0.5, 0.5 - (tmax * 0x1p-52 + 0x1p-60), c0, u2);
                unsure = unsure || u1 || u2;
            }
            quick = !__any(unsure ? 1 : 0);
        }
        if (quick) {
            a0 = wave_sum_u64(a0);
            a1 = a0;
            ok_a = true;
            if (dual) {
                c0 = wave_sum_u64(c0);
                c1 = c0;
                ok_b = true;
            }
        } else if (guess) {
            // the reference's own division, then the general fold
            double d[kChainRun];
            bool finite = src.quotients(a, d);
            if (past_end) {
                finite = true;
#pragma unroll
                for (int t = 0; t < kChainRun; ++t) {
                    d[t] = g0 + t < e ? d[t] : 0.0;
                    finite = finite && d[t] >= 0.0 && d[t] < __builtin_inf();
                }
            }
            double run_max = d[0];
#pragma unroll
            for (int t = 1; t < kChainRun; ++t) run_max = max_f64(run_max, d[t]);
            {
                bool tie = false;
                bool ok = finite && run_max * scale_a < two52;
                a0 = 0; a1 = 0;
                if (ok) chain_fold_run(d, scale_a, a0, a1, tie);
                if (__any(tie ? 1 : 0)) {
                    chain_compose_wave(a0, a1);
                } else {
                    a0 = wave_sum_u64(a0);
                    a1 = a0;
                }
                ok_a = __all(ok ? 1 : 0);
            }
            c0 = 0; c1 = 0;
            if (dual) {
                bool tie = false;
                bool ok = finite && run_max * scale_b < two52;
                if (ok) chain_fold_run(d, scale_b, c0, c1, tie);
                if (__any(tie ? 1 : 0)) {
                    chain_compose_wave(c0, c1);
                } else {
                    c0 = wave_sum_u64(c0);
                    c1 = c0;
                }
                ok_b = __all(ok ? 1 : 0);
            }
        }
        if (lane == 0) {
            SegMap m;
            m.inc[0][0] = a0; m.inc[0][1] = a1; m.inc[1][0] = c0; m.inc[1][1] = c1;
            m.ebiased = guess ? eb_lo : 0;
            m.flags = (ok_a ? 1u : 0u) | (dual ? 2u : 0u) | (ok_b ? 4u : 0u);
            m.pad = 0;
            out[seg] = m;
        }
    }
}

// One wave per chain: the records of split_chain_kernel applied to the true sum (see above).
__global__ __launch_bounds__(64) void split_finish_kernel(
    const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
    const int64_t* __restrict__ cloud_pt_off, const int64_t* __restrict__ cloud_seg_off,
    const double* __restrict__ mats, const double* __restrict__ planes, int n_poses, int n_clouds,
    int64_t total_pts, int K, const SegMap* __restrict__ maps, int64_t total_segs,
    const double* __restrict__ first_sum, double* __restrict__ sref /* rewritten from pose 0 */,
    double* __restrict__ out_sum, double* __restrict__ out_resummed)
{
    constexpr unsigned long long kTwo53 = 1ull << 53;
    constexpr unsigned long long kMantissa = (1ull << 52) - 1;
    const int chain = blockIdx.x;
    const int j = chain / n_poses, k = chain % n_poses;
    const int64_t b = cloud_pt_off[j], e = cloud_pt_off[j + 1];
    const int64_t b0 = b & ~(int64_t)(kChainRun - 1);
    const int n_seg = (int)((e - b0 + kChainSeg - 1) / kChainSeg);
    const int per = split_part_segments(n_seg, K);
    const int s1 = per < n_seg ? per : n_seg;      // part 0 covered the segments below
    const int lane = threadIdx.x;

    DistanceSource src;
    src.xs = xs; src.ys = ys; src.zs = zs;
    src.limit = total_pts;
    const double* M = mats + ((int64_t)k * n_clouds + j) * 12;
    const double* pl = planes + (int64_t)k * kPlaneStride;
#pragma unroll
    for (int i = 0; i < 12; ++i) src.M[i] = M[i];
#pragma unroll
    for (int i = 0; i < 7; ++i) src.pl[i] = pl[i];

    double s = first_sum[(int64_t)k * n_clouds + j];
    const SegMap* mp = maps + ((int64_t)k * total_segs + cloud_seg_off[j]);
    double* ref = k == 0 ? sref + cloud_seg_off[j] : nullptr;
    int resummed = 0;
    // 64 records per trip, one per lane; the next trip's records are in flight while this one is applied
    // (a single wave has nothing else to hide the latency behind)
    SegMap m_next;
    m_next.inc[0][0] = m_next.inc[0][1] = m_next.inc[1][0] = m_next.inc[1][1] = 0;
    m_next.ebiased = 0; m_next.flags = 0; m_next.pad = 0;
    if (s1 + lane < n_seg) m_next = mp[s1 + lane];
    for (int base = s1; base < n_seg; base += 64) {
        const int nl = n_seg - base < 64 ? n_seg - base : 64;
        const SegMap m = m_next;
        if (base + 64 + lane < n_seg) m_next = mp[base + 64 + lane];
        int cur = 0;                                // segments base .. base+cur-1 are in s
        while (cur < nl) {
            const unsigned long long sbits = (unsigned long long)__double_as_longlong(s);
            const int eb = (int)(sbits >> 52);      // sign included
            const bool usable = eb >= 64 && eb <= 2000;
            const unsigned long long S = (sbits & kMantissa) | (1ull << 52);
            unsigned long long a0 = 0, a1 = 0;
            bool valid = false;
            if (lane >= cur && lane < nl) {
                if ((m.flags & 1u) && m.ebiased == eb) { a0 = m.inc[0][0]; a1 = m.inc[0][1]; valid = true; }
                else if ((m.flags & 6u) == 6u && m.ebiased + 1 == eb) { a0 = m.inc[1][0]; a1 = m.inc[1][1]; valid = true; }
                a0 = a0 < (kTwo53 << 1) ? a0 : (kTwo53 << 1);   // rejected anyway; keeps the scan from wrapping
                a1 = a1 < (kTwo53 << 1) ? a1 : (kTwo53 << 1);
            }
            chain_scan<64>(a0, a1, lane);
            const unsigned long long tot = (S & 1ull) ? a1 : a0;
            const bool bad = lane >= cur && lane < nl && (!valid || !usable || S + tot >= kTwo53);
            const unsigned long long mask = __ballot(bad ? 1 : 0);
            const int adv = mask ? __builtin_ctzll(mask) : nl;          // first segment that cannot be applied
            const unsigned long long prev = dpp_move_u64<kDppWaveShr1, 0xf>(tot);
            if (ref && lane >= cur && lane < adv) {
                const unsigned long long at = lane > cur ? S + prev : S;
                ref[base + lane] = __longlong_as_double((long long)((at & kMantissa) | ((unsigned long long)eb << 52)));
            }
            if (adv > cur) {
                const unsigned long long before = read_lane_u64(tot, adv - 1);
                s = __longlong_as_double((long long)(((S + before) & kMantissa) | ((unsigned long long)eb << 52)));
            }
            if (adv < nl) {
                // this segment again, from the true sum, by the sequential routine
                if (ref && lane == 0) ref[base + adv] = s;
                const int64_t sb = b0 + (int64_t)(base + adv) * kChainSeg;
                const int64_t se = sb + kChainSeg < e ? sb + kChainSeg : e;
                double s_new, mx;
                chain_reduce<1, false>(src, sb > b ? sb : b, se, s, s_new, mx);
                s = s_new;
                ++resummed;
                cur = adv + 1;
            } else {
                cur = nl;
            }
        }
    }
    if (lane == 0) {
        if (ref) ref[n_seg] = s;
        out_sum[(int64_t)k * n_clouds + j] = s;
        out_resummed[(int64_t)k * n_clouds + j] = (double)resummed;
    }
}

// Ordered sum of a plain array of non-negative values (the accumulation primitive on
// its own); a negative, infinite or NaN element is added one by one with its run.
constexpr int kChainWaves = 16;   // of the single-chain kernel

__global__ __launch_bounds__(64 * kChainWaves) void ordered_sum_kernel(
    const double* __restrict__ values, int64_t n, double* __restrict__ out_sum, double* __restrict__ out_max)
{
    ArraySource src;
    src.v = values;
    src.limit = n;
    double sum, mx;
    chain_reduce<kChainWaves, true>(src, 0, n, 0.0, sum, mx);
    if (threadIdx.x == 0) {
        out_sum[0] = sum;
        out_max[0] = mx;
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

// Workgroups of the moment pass: each takes a contiguous share of the 4096-point block table (a fixed
// assignment: the partial sums, and so the result, are the same bits on every run).
constexpr int kMomentGrid = 2048;

__host__ __device__ inline int moment_rows(int n_table) { return n_table < kMomentGrid ? n_table : kMomentGrid; }

__global__ __launch_bounds__(256) void moments_kernel(
    const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
    const BlockDesc* __restrict__ blocks, int n_table, const double* __restrict__ mats, double ox, double oy,
    double oz, InlierMask mask, double* __restrict__ part /* [moment_rows(n_table)][kMoments] */)
{
    __shared__ double sM[12];
    __shared__ double red[4][kMoments];
    const int rows = moment_rows(n_table);
    const int t0 = (int)((int64_t)n_table * blockIdx.x / rows), t1 = (int)((int64_t)n_table * (blockIdx.x + 1) / rows);
    double m[kMoments];
#pragma unroll
    for (int k = 0; k < kMoments; ++k) m[k] = 0.0;
    int cloud = -1;
    for (int t = t0; t < t1; ++t) {
        const BlockDesc bd = blocks[t];
        if (bd.cloud != cloud) {             // uniform over the workgroup
            __syncthreads();
            if (threadIdx.x < 12) sM[threadIdx.x] = mats[(int64_t)bd.cloud * 12 + threadIdx.x];
            __syncthreads();
            cloud = bd.cloud;
        }
        // four points per thread and trip: their twelve loads are in flight together (the pass is bound by
        // HBM and FP64 issue about equally).  The transform keeps the reference's operation order (the point
        // must be the float the optimiser sees); the moment sums are fused multiply-adds -- any order and
        // rounding of these sums is as good as another, only the fixed order matters.
        constexpr int kPer = 4;
        for (int64_t i0 = bd.begin + threadIdx.x; i0 < bd.end; i0 += (int64_t)kPer * blockDim.x) {
            float X[kPer], Y[kPer], Z[kPer];
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int64_t i = i0 + (int64_t)u * blockDim.x;
                const int64_t ic = i < bd.end ? i : bd.begin;
                X[u] = xs[ic]; Y[u] = ys[ic]; Z[u] = zs[ic];
            }
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                if (i0 + (int64_t)u * blockDim.x >= bd.end) continue;
                const double xd = (double)X[u], yd = (double)Y[u], zd = (double)Z[u];
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
                m[4] = __builtin_fma(x, x, m[4]); m[5] = __builtin_fma(x, y, m[5]); m[6] = __builtin_fma(y, y, m[6]);
                m[7] = __builtin_fma(x, z, m[7]); m[8] = __builtin_fma(y, z, m[8]); m[9] = __builtin_fma(z, z, m[9]);
            }
        }
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

// Sums the per-block partial moments in a fixed order (one workgroup: thread t takes rows
// t, t + 1024, ...; then a wave tree and the 16 wave totals in wave order).
__global__ __launch_bounds__(1024) void moments_finalize_kernel(const double* __restrict__ part,
                                                                int n_blocks,
                                                                double* __restrict__ out)
{
    __shared__ double red[16][kMoments];
    double m[kMoments];
#pragma unroll
    for (int k = 0; k < kMoments; ++k) m[k] = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += 1024) {
        const double* row = part + (int64_t)b * kMoments;
#pragma unroll
        for (int k = 0; k < kMoments; ++k) m[k] += row[k];
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
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += red[w][threadIdx.x];
        out[threadIdx.x] = s;
    }
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
