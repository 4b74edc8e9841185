// kernels_exact.hpp -- the reference arithmetic: ordered sums of point-to-plane distances (whole chains and chains cut into parts), the per-point report, and the FP32 screening pass of the line search
//
// One of the kernel headers of libccal; each is included by exactly one translation unit
// (api_exact.hip), so its __global__ functions have one definition in the library.
#pragma once

#include "kernels_common.hpp"

#pragma clang fp contract(off)

namespace ccal {

// ---------------------------------------------------------------------------
// Screening pass: the folded FP32 objective of a few poses, one pass over the points
// ---------------------------------------------------------------------------
// sum_i |w.(p_i - c, 1)| per (pose, cloud) for up to a few dozen poses at once: every lane keeps
// 8 points in registers and walks the poses in tiles of kScreenTile (the four-vectors are uniform:
// scalar loads).  Roundings per residual: three fused multiply-adds (the constant enters as the
// innermost addend); per partial sum: 8 FP32 adds in the lane and 6 in the wave reduction;
// everything after that is FP64 in a fixed order.  The caller turns these counts into an error bound.
constexpr int kScreenTile = 16;
constexpr int kScreenParts = 2;         // workgroups per block of the table, see screen_cost_kernel
constexpr int kScreenLaneAdds = 8 + 6;  // FP32 adds behind a value that leaves a wave

// The four-vectors of a screening request and their share of the error budget, one thread per
// (pose, cloud), from one record per pose: t[3], sin[3], cos[3] of the pose (the host takes the sines
// and cosines), plane a, b, c, d and 1 / |n|.  Same expressions as the host's pose algebra.
//   budget = 2^-24 sum_r |n_r|/|n| (|(M (c,1))_r| + sum_a |R_ra| rho_a)  +  5 * 2^-24 (sum_a |w_a| rho_a + |w4|)
// (reference side: every transformed coordinate rounded to float; fast side: w and p - c rounded to
// float and three fused multiply-adds; R = the 3x3 part of M, rho_a = mean |p_a - c_a| over the cloud's
// points along the cloud's own axis a.  Both sums are means of per-point bounds that are linear in the
// |p_a - c_a|.  Depth-camera clouds of a plane are thin along one axis, and w points along it: on the
// benchmark's scene this is 0.45 of the same budget written with the mean distance |p - c| in both places,
// and about half as many trial centres stay undecided.)
constexpr int kScreenRec = 16;          // doubles per pose record (14 used)

__global__ __launch_bounds__(64) void screen_prep_kernel(
    const double* __restrict__ recs, const double* __restrict__ base_T_flange, const double* __restrict__ centres,
    const double* __restrict__ cloud_bound /* [cloud][kCloudBound] */, int n_poses, int n_clouds,
    f32x4* __restrict__ W /* [pose][cloud] */, double* __restrict__ budget /* [pose][cloud] */)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_poses * n_clouds) return;
    const int k = i / n_clouds, j = i % n_clouds;
    const double* r = recs + (int64_t)k * kScreenRec;
    const double t[3] = {r[0], r[1], r[2]}, sn[3] = {r[3], r[4], r[5]}, cs[3] = {r[6], r[7], r[8]};
    const double pl[4] = {r[9], r[10], r[11], r[12]};
    const double inv_norm = r[13];
    double R[9], M[12], wv[4];
    rotation_zyx(sn, cs, R);
    compose_rows(base_T_flange + 16 * (int64_t)j, R, t, M);
    const double* c = centres + 3 * (int64_t)j;
    fold_plane(M, pl, pl[3], inv_norm, c, wv);
    const f32x4 w4 = {(float)wv[0], (float)wv[1], (float)wv[2], (float)wv[3]};
    W[i] = w4;
    const double* rho = cloud_bound + kCloudBound * (int64_t)j + 2;     // along the cloud's x, y, z
    double ex = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double q = M[a * 4] * c[0] + M[a * 4 + 1] * c[1] + M[a * 4 + 2] * c[2] + M[a * 4 + 3];
        const double spread = __builtin_fabs(M[a * 4]) * rho[0] + __builtin_fabs(M[a * 4 + 1]) * rho[1] + __builtin_fabs(M[a * 4 + 2]) * rho[2];
        ex += __builtin_fabs(pl[a]) * inv_norm * (__builtin_fabs(q) + spread);
    }
    const double wspread = __builtin_fabs(wv[0]) * rho[0] + __builtin_fabs(wv[1]) * rho[1] + __builtin_fabs(wv[2]) * rho[2];
    budget[i] = 0x1p-24 * ex + 5.0 * 0x1p-24 * (wspread + __builtin_fabs(wv[3]));
}

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
    for (int b = kScreenParts * cloud_block_off[j] + lane; b < kScreenParts * cloud_block_off[j + 1]; b += 64)
        s += partial[(int64_t)b * n_poses + p];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) out[i] = s;
}

// Totals over the 64 lanes of a wave of T (16 or 8) values per lane at once: the T lanes of a group
// reduce-scatter (a lane keeps one half of its values and adds the partner's copy of that half, log2 T
// exchange steps), then the 64 / T groups are added; lane l returns the total of value l & (T - 1).
// Six FP32 adds per total either way, as a butterfly per value would take, for a third of the instructions.
template <int T>
__device__ inline float wave_sum_tile(const float (&v)[T], int lane)
{
    static_assert(T == 16 || T == 8, "tiles of 16 or 8 values");
    const int col = lane & (T - 1);
    float w8[8], w4[4], w2[2];
    if constexpr (T == 16) {
        const bool hi = (col & 8) != 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float keep = hi ? v[8 + i] : v[i], give = hi ? v[i] : v[8 + i];
            w8[i] = keep + __shfl_xor(give, 8, 64);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) w8[i] = v[i];
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
    if constexpr (T == 8) r += __shfl_xor(r, 8, 64);
    r += __shfl_xor(r, 16, 64);
    r += __shfl_xor(r, 32, 64);
    return r;
}

// One tile of T poses p0 .. p0 + T - 1 over the eight points of every lane; a short last tile repeats
// its last pose.  kRagged: some of the workgroup's point slots are padding or lie beyond the cloud (their
// coordinates are zero and their row of ones, O, is zero too); otherwise every slot is a point and the
// constant enters the chain as it is.  Either way a residual is fma(w0, x, fma(w1, y, fma(w2, z, w3))).
template <int T, bool kRagged>
__device__ inline void screen_tile(const f32x4 (&X)[2], const f32x4 (&Y)[2], const f32x4 (&Z)[2], const f32x4 (&O)[2],
                                   const f32x4* __restrict__ w, int n_clouds, int j, int p0, int n_poses,
                                   float (*red)[kScreenTile], double* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int p = p0 + t < n_poses ? p0 + t : n_poses - 1;
        const f32x4 wv = w[(int64_t)p * n_clouds + j];
        // two residuals per instruction (packed FP32); the lane adds them in point order
        const f32x2 w0 = {wv[0], wv[0]}, w1 = {wv[1], wv[1]}, w2 = {wv[2], wv[2]}, w3 = {wv[3], wv[3]};
        float a = 0.f;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int h = 0; h < 4; h += 2) {
                const f32x2 x = {X[g][h], X[g][h + 1]}, y = {Y[g][h], Y[g][h + 1]}, z = {Z[g][h], Z[g][h + 1]};
                f32x2 c = w3;
                if constexpr (kRagged) {
                    const f32x2 o = {O[g][h], O[g][h + 1]};
                    c = w3 * o;            // exact: o is 0 or 1
                }
                const f32x2 r = __builtin_elementwise_fma(w0, x, __builtin_elementwise_fma(w1, y,
                                __builtin_elementwise_fma(w2, z, c)));
                a += __builtin_fabsf(r[0]);
                a += __builtin_fabsf(r[1]);
            }
        acc[t] = a;
    }
    const float tot = wave_sum_tile<T>(acc, lane);
    if (lane < T) red[wave][lane] = tot;
    __syncthreads();
    if (threadIdx.x < T && p0 + (int)threadIdx.x < n_poses) {
        const int t = threadIdx.x;
        out[p0 + t] = (((double)red[0][t] + (double)red[1][t]) + (double)red[2][t]) + (double)red[3][t];
    }
    __syncthreads();
}

// A workgroup takes half a block of the table (kScreenParts per block: 2048 points, eight per lane --
// two groups, so that the wave reduction of a tile is shared by twice the points; with whole blocks the
// benchmark's 1465 workgroups were one and a half rounds of what the chip holds at once, and every round
// began with all of them waiting for their loads).  Tiles of 16 poses, and one of 8 when no more are left.
__global__ __launch_bounds__(256) void screen_cost_kernel(
    const f32x4* __restrict__ groups, const int* __restrict__ group_off, const int64_t* __restrict__ cloud_pt_off,
    const BlockDesc* __restrict__ blocks, const f32x4* __restrict__ w /* [pose][cloud] */, int n_poses, int n_clouds,
    double* __restrict__ partial /* [block * kScreenParts + part][pose] */)
{
    __shared__ float red[4][kScreenTile];
    constexpr int kPartGroups = 64 / kScreenParts;      // a block of the table is 4096 consecutive points = 64 groups
    const BlockDesc bd = blocks[blockIdx.x / kScreenParts];
    const int part = blockIdx.x % kScreenParts;
    const int j = bd.cloud;
    const int64_t lb = bd.begin - cloud_pt_off[j], le = bd.end - cloud_pt_off[j];
    const int g0 = group_off[j] + (int)(lb / kGroupPoints) + part * kPartGroups;
    const int g1 = group_off[j] + (int)((le + kGroupPoints - 1) / kGroupPoints);
    double* out = partial + (int64_t)blockIdx.x * n_poses;
    if (g0 >= g1) {                                   // a short last block: nothing in this part
        for (int p = threadIdx.x; p < n_poses; p += 256) out[p] = 0.0;
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, gsel = lane >> 4;
    // The fourth row of a group (1 for a point, 0 for padding) is not loaded: it follows from the point's index.
    const int64_t n_j = cloud_pt_off[j + 1] - cloud_pt_off[j];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 X[2] = {zero, zero}, Y[2] = {zero, zero}, Z[2] = {zero, zero}, O[2] = {zero, zero};   // absent groups add |0|
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int g = g0 + wave * (kPartGroups / 4) + h * 4 + gsel;
        if (g < g1) {
            const f32x4* gp = groups + (int64_t)g * 64 + col;
            X[h] = gp[0]; Y[h] = gp[16]; Z[h] = gp[32];
            const int64_t first = (int64_t)(g - group_off[j]) * kGroupPoints + col;
#pragma unroll
            for (int u = 0; u < 4; ++u) O[h][u] = first + u * kTilePoints < n_j ? 1.f : 0.f;
        }
    }
    // every slot of this workgroup is a point of the cloud (true for all but a cloud's last workgroup)
    const bool full = (int64_t)(g0 - group_off[j] + kPartGroups) * kGroupPoints <= n_j;
    int p0 = 0;
    if (full) {
        for (; n_poses - p0 > 8; p0 += 16) screen_tile<16, false>(X, Y, Z, O, w, n_clouds, j, p0, n_poses, red, out);
        if (p0 < n_poses) screen_tile<8, false>(X, Y, Z, O, w, n_clouds, j, p0, n_poses, red, out);
    } else {
        for (; n_poses - p0 > 8; p0 += 16) screen_tile<16, true>(X, Y, Z, O, w, n_clouds, j, p0, n_poses, red, out);
        if (p0 < n_poses) screen_tile<8, true>(X, Y, Z, O, w, n_clouds, j, p0, n_poses, red, out);
    }
}

// ---------------------------------------------------------------------------
// Exact path
// ---------------------------------------------------------------------------
// Rows 0..2 of M = base_T_flange_j * F(pose) for every (pose, cloud) of a request, one thread each, from
// one record per pose: t[3], sin[3], cos[3] (sines and cosines taken by the host's libm).  The same
// contraction-free expressions as the host's pose_matrices, so the same bits.
constexpr int kPoseRec = 9;

__global__ __launch_bounds__(64) void pose_table_kernel(const double* __restrict__ recs,
                                                        const double* __restrict__ base_T_flange, int n_poses,
                                                        int n_clouds, double* __restrict__ mats /* [pose][cloud][12] */,
                                                        const double* __restrict__ planes_in,
                                                        double* __restrict__ planes_out /* [pose][kPlaneStride] */,
                                                        const double* __restrict__ expected_sum /* [pose][cloud] or null */,
                                                        const double* __restrict__ sref /* running sums of an earlier evaluation */,
                                                        const int64_t* __restrict__ cloud_seg_off,
                                                        double* __restrict__ guess_scale /* [pose][cloud][2]: factors for the lower and the upper end */)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_poses * n_clouds) return;
    const int k = i / n_clouds, j = i % n_clouds;
    // What the caller expects this chain to sum to, over what the chain that left the running sums behind
    // summed to: the part kernel scales those sums by it before it guesses its units (1 when unknown), and
    // widens them by 1/64 either way -- by more when the totals have moved a lot, since the sums along the
    // chain then have not all moved alike (up to 0.3: the widened ends stay within a factor of two of each
    // other, which is as far as a record's two units reach).
    double gs = 1.0, w = 0.015625;
    if (expected_sum) {
        const double q = expected_sum[i] / sref[cloud_seg_off[j + 1] - 1];    // a cloud's last entry: the chain's total
        if (q > 0.0625 && q < 16.0) {             // a NaN fails both
            gs = q;
            const double moved = 0.5 * __builtin_fabs(q - 1.0);
            w = moved > w ? (moved < 0.3 ? moved : 0.3) : w;
        }
    }
    guess_scale[2 * (int64_t)i] = gs * (1.0 - w);
    guess_scale[2 * (int64_t)i + 1] = gs * (1.0 + w);
    // the planes move into device memory with the table: every workgroup of the chain kernels reads its
    // pose's plane, and the request's copy sits in host memory
    if (j == 0) {
#pragma unroll
        for (int e = 0; e < kPlaneStride; ++e) planes_out[(int64_t)k * kPlaneStride + e] = planes_in[(int64_t)k * kPlaneStride + e];
    }
    const double* r = recs + (int64_t)k * kPoseRec;
    const double t[3] = {r[0], r[1], r[2]}, sn[3] = {r[3], r[4], r[5]}, cs[3] = {r[6], r[7], r[8]};
    double R[9], M[12];
    rotation_zyx(sn, cs, R);
    compose_rows(base_T_flange + 16 * (int64_t)j, R, t, M);
#pragma unroll
    for (int e = 0; e < 12; ++e) mats[(int64_t)i * 12 + e] = M[e];
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
    // the last pose's chains leave their running sums behind for a later split evaluation
    chain_reduce<W, kWantMax>(src, cloud_pt_off[j], cloud_pt_off[j + 1], 0.0, sum, mx,
                              seg_sums && k == n_poses - 1 ? seg_sums + cloud_seg_off[j] : nullptr);
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
    const double* __restrict__ guess_scale /* [pose][cloud][2]: what to multiply the ends of a segment's range of sref by */,
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
    // the previous evaluation's sums scaled to this chain's expected total and widened (pose_table_kernel)
    const double widen_lo = guess_scale[2 * ((int64_t)k * n_clouds + j)], widen_hi = guess_scale[2 * ((int64_t)k * n_clouds + j) + 1];
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

        const double lo = ref[seg] * widen_lo, hi = ref[seg + 1] * widen_hi;
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

// The sequential sum continued from s over one segment held eight elements per lane (d, in chain order;
// `finite`: the lane's elements are ordinary non-negative numbers), by one wave and in registers only: the
// lane-level half of chain_reduce (same units, same tests, same one-by-one additions where the sum leaves
// its binade or is not usable), without its workgroup barriers and LDS round trips, which a lone wave
// can only wait out.  Returns the same value in every lane.
__device__ inline double wave_resum_segment(const double (&d)[kChainRun], bool finite, double s)
{
    constexpr unsigned long long kTwo53 = 1ull << 53;
    constexpr unsigned long long kMantissa = (1ull << 52) - 1;
    const double two52 = 4503599627370496.0;
    const int lane = threadIdx.x & 63;
    double run_max = d[0];
#pragma unroll
    for (int t = 1; t < kChainRun; ++t) run_max = max_f64(run_max, d[t]);
    int consumed = 0;                                   // lanes whose runs are in s
    while (consumed < 64) {
        const unsigned long long sbits = (unsigned long long)__double_as_longlong(s);
        const int ebiased = (int)(sbits >> 52);         // sign included
        const bool usable = ebiased >= 64 && ebiased <= 2000;
        if (!usable) {
            // s is zero (or out of the range the units cover): the next run one by one
            double sv = s;
#pragma unroll
            for (int t = 0; t < kChainRun; ++t) sv += d[t];
            s = __longlong_as_double((long long)read_lane_u64((unsigned long long)__double_as_longlong(sv), consumed));
            ++consumed;
            continue;
        }
        const double scale = __longlong_as_double((long long)((unsigned long long)(2098 - ebiased) << 52));
        const unsigned long long S = (sbits & kMantissa) | (1ull << 52);
        unsigned long long i0 = 0, i1 = 0;              // consumed lanes: the identity
        bool ok = true, tie = false;
        if (lane >= consumed) {
            ok = finite && run_max * scale < two52;
            if (ok) chain_fold_run(d, scale, i0, i1, tie);
        }
        chain_scan<64>(i0, i1, lane);
        const unsigned long long tot = (S & 1ull) ? i1 : i0;
        const bool bad = !ok || S + tot >= kTwo53;
        const unsigned long long mask = __ballot(bad ? 1 : 0);
        if (!mask) {
            const unsigned long long all = read_lane_u64(tot, 63);
            return __longlong_as_double((long long)(((S + all) & kMantissa) | ((unsigned long long)ebiased << 52)));
        }
        // lane L's run leaves the binade (or cannot be folded): everything before it by the prefix, the run
        // itself one by one
        const int L = __builtin_ctzll(mask);
        const unsigned long long before = L > 0 ? read_lane_u64(tot, L - 1) : 0ull;
        double sv = __longlong_as_double((long long)(((S + before) & kMantissa) | ((unsigned long long)ebiased << 52)));
#pragma unroll
        for (int t = 0; t < kChainRun; ++t) sv += d[t];
        s = __longlong_as_double((long long)read_lane_u64((unsigned long long)__double_as_longlong(sv), L));
        consumed = L + 1;
    }
    return s;
}

// One wave per chain: the records of split_chain_kernel applied to the true sum (see above).
__global__ __launch_bounds__(64) void split_finish_kernel(
    const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
    const int64_t* __restrict__ cloud_pt_off, const int64_t* __restrict__ cloud_seg_off,
    const double* __restrict__ mats, const double* __restrict__ planes, int n_poses, int n_clouds,
    int64_t total_pts, int K, const SegMap* __restrict__ maps, int64_t total_segs,
    const double* __restrict__ first_sum, double* __restrict__ sref /* rewritten from pose ref_pose */, int ref_pose,
    double* __restrict__ out_sum, double* __restrict__ out_resummed)
{
    constexpr unsigned long long kTwo53 = 1ull << 53;
    constexpr unsigned long long kMantissa = (1ull << 52) - 1;
    // A few hundred lone waves, each bound by its own latency: when they share the chip with another
    // launch's chains (a two-part request) they must not queue for issue slots behind five busy waves
    // per SIMD (measured: 123 us beside a chain launch without this, 62 us with it, 25 us alone).
    __builtin_amdgcn_s_setprio(3);
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
    // One pose of a request (the caller says which; -1: none of this launch) leaves its sums behind for
    // the next request's guesses.
    double* ref = k == ref_pose ? sref + cloud_seg_off[j] : nullptr;
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
                const int64_t sb = b0 + (int64_t)(base + adv) * kChainSeg;     // part 0 holds the chain's first segment: sb > b
                const int64_t g0 = sb + lane * kChainRun;
                double d[kChainRun];
                bool finite = src.values(src.fetch(g0), d);
                if (g0 + kChainRun > e) {
                    // the chain's last segment: elements past the end count as +0.0
                    finite = true;
#pragma unroll
                    for (int t = 0; t < kChainRun; ++t) {
                        d[t] = g0 + t < e ? d[t] : 0.0;
                        finite = finite && d[t] >= 0.0 && d[t] < __builtin_inf();
                    }
                }
                s = wave_resum_segment(d, finite, s);
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


} // namespace ccal
