// kernels_fit.hpp -- plane fits: the moment pass of the least-squares fit and of the RANSAC refit; RANSAC hypotheses, their FP32 inlier brackets and exact recounts
//
// One of the kernel headers of libccal; each is included by exactly one translation unit
// (api_fit.hip), so its __global__ functions have one definition in the library.
#pragma once

#include "kernels_common.hpp"
#include "bf16x3_split.hpp"

#pragma clang fp contract(off)

namespace ccal {

// Moments of the base-frame points q = float(M p): for the z-regression plane fit
// (src/camera_calibration_optimization.cpp:100-119 on the cloud of :240-272) and,
// with an inlier mask, for the RANSAC refit.  Accumulated about a fixed origin
// `org` (an approximate centroid) so the normal equations / covariance stay well
// conditioned in FP64.  Order of the 10 sums: 1, x, y, z, xx, xy, yy, xz, yz, zz.
// mask: plane (a,b,c,d,|n|) and threshold; thr < 0 keeps every point.

struct InlierMask {
    double pl[5];
    double thr;
};

// Workgroups of the moment pass: each takes a contiguous share of the 4096-point block table (a fixed
// assignment: the partial sums, and so the result, are the same bits on every run).

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

// The residual contraction of the search, counting instead of summing (RANSAC scoring).  The FP32 residual
// of a point lies within a known distance e of the reference-arithmetic one (the caller derives e per
// cloud), so a point with |w.p| >= thr + e is no inlier of the reference's count: the points below that
// mark bound a hypothesis's inlier count from above.  thr_tab[cloud][1] = thr + e, rounded upwards
// ([0] = thr - e is not used here).  partial[split][candidate] = the possible inliers.  The caller re-counts
// exactly what these bounds cannot exclude; a second counter for the points below thr - e (a lower
// bound, which an earlier form of this kernel kept) only chose which hypotheses to re-count first and
// doubled the kernel's vector instructions.
// Padding points have residual 0 and are counted; the host subtracts them.  Lane counters are 32 bits
// wide: a lane sees 1/16 of a split's points.
//
// The contraction runs on the bf16 matrix pipe with both FP32 operands as exact three-way bf16 splits
// (bf16x3_split.hpp), as in residual_bf16x3_kernel: the matrix instruction then issues in the shadow of
// the eight compare / add-carry instructions that each result costs, which bound the kernel either way
// (with the FP32-input instruction, which shares the vector unit's issue port, the two added up).
template <int NA>
__device__ inline void count_possible(const f32x4& d, float thi, unsigned int (&hi)[4])
{
#pragma unroll
    for (int r = 0; r < 4; ++r) hi[r] += (__builtin_fabsf(d[r]) < thi) ? 1u : 0u;
}

template <int NA>
__global__ __launch_bounds__(256, 2) void inlier_mfma_kernel(
    const float* __restrict__ W, const f32x4* __restrict__ groups,
    const int* __restrict__ group_off, unsigned long long* __restrict__ partial,
    const f32x2* __restrict__ thr_tab, int n_clouds, int n_ctiles, int n_splits)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ctile0 = (blockIdx.x * 4 + wave) * NA;
    const int split = blockIdx.y;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    constexpr int kBurst = 4;
    static_assert(NA % kBurst == 0, "candidate tiles per wave must be a multiple of the burst");

    unsigned int hi[NA][4];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) hi[a][r] = 0;

    for (int j = 0; j < n_clouds; ++j) {
        const int g0 = group_off[j];
        const int G = group_off[j + 1] - g0;
        const int gb = g0 + (int)(((int64_t)G * split) / n_splits);
        const int ge = g0 + (int)(((int64_t)G * (split + 1)) / n_splits);
        if (gb >= ge) continue;
        const float thi = thr_tab[j][1];
        bf16x8 A[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a)
            A[a] = split_candidate_operand(W[((int64_t)j * n_ctiles + ctile0 + a) * 64 + lane], lane >= 48);
        const f32x4* gp = groups + (int64_t)gb * 64 + lane;
        f32x4 b = gp[0];
        bf16x8 Bc = split_point_operand(b[0]);
        // Bursts: kBurst matrix instructions back to back, then the counting of the previous burst's results
        // (long landed by then); the bf16 split of the next point tile rides along.  The last burst of a
        // group is carried into the next trip; the carry starts at +infinity, which no bound lies above.
        const float inf = __builtin_inff();
        f32x4 carry[kBurst];
#pragma unroll
        for (int q = 0; q < kBurst; ++q) carry[q] = f32x4{inf, inf, inf, inf};
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
                        count_possible<NA>(done, thi, hi[a]);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x002, 8 * kBurst + 5, 0);
                }
                Bc = Bn;
            }
#pragma unroll
            for (int q = 0; q < kBurst; ++q) carry[q] = d[4 * NA - kBurst + q];
            b = bn;
        }
#pragma unroll
        for (int q = 0; q < kBurst; ++q) count_possible<NA>(carry[q], thi, hi[NA - kBurst + q]);
    }
    unsigned long long* out = partial + (int64_t)split * n_ctiles * 16;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            unsigned long long w = hi[a][r];
            w += __shfl_xor(w, 1, 64);
            w += __shfl_xor(w, 2, 64);
            w += __shfl_xor(w, 4, 64);
            w += __shfl_xor(w, 8, 64);
            if ((lane & 15) == 0) out[(int64_t)(ctile0 + a) * 16 + (lane >> 4) * 4 + r] = w;
        }
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

// Exact inlier counts for PB hypotheses (reference-arithmetic distance < thr, the strict compare of a
// sequential scan).  The point is carried to the base frame once (src/helpers.hpp:139-151, rounded to
// float) and measured against every plane; the division by the norm is the correctly rounded one either
// way -- by the precomputed reciprocal where its corrections are exact (divide_by_known: plane rows carry
// the norm, its rounded reciprocal and a flag for the norm's range), by the divider elsewhere.
template <int PB>
__global__ __launch_bounds__(256) void exact_count_kernel(
    const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
    const BlockDesc* __restrict__ blocks, int n_blocks, const double* __restrict__ mats,
    const double* __restrict__ planes /* PB x kPlaneStride */, int n_planes /* <= PB in use */, double thr,
    unsigned long long* __restrict__ counts)
{
    __shared__ double sM[12];
    __shared__ double sP[PB * kPlaneStride];
    __shared__ unsigned int sC[4][PB];
    for (int i = threadIdx.x; i < PB * kPlaneStride; i += blockDim.x) sP[i] = planes[i];
    constexpr unsigned kLo = 0x1a700000u, kHi = 0x65700000u;   // high words of 2^-600, 2^600
    // a workgroup walks a contiguous share of the block table and adds its counts to the totals once:
    // with one workgroup per block, every wave's sixteen atomic adds landed on the same two cache lines
    // and the pass took 10 ms for 10^8 points, ten times what its arithmetic needs
    const int b0 = (int)(((int64_t)n_blocks * blockIdx.x) / gridDim.x), b1 = (int)(((int64_t)n_blocks * (blockIdx.x + 1)) / gridDim.x);
    unsigned int cnt[PB];
#pragma unroll
    for (int k = 0; k < PB; ++k) cnt[k] = 0;
    int cloud = -1;
    for (int b = b0; b < b1; ++b) {
        const BlockDesc bd = blocks[b];
        if (bd.cloud != cloud) {                      // uniform: the next cloud's matrix
            __syncthreads();
            if (threadIdx.x < 12) sM[threadIdx.x] = mats[(int64_t)bd.cloud * 12 + threadIdx.x];
            __syncthreads();
            cloud = bd.cloud;
        }
        // four points of a thread in flight: a trip of one load and its arithmetic was all latency
        constexpr int kInFlight = 4;
        for (int64_t base = bd.begin + threadIdx.x; base < bd.end; base += (int64_t)kInFlight * blockDim.x) {
          float px[kInFlight], py[kInFlight], pz[kInFlight];
#pragma unroll
          for (int u = 0; u < kInFlight; ++u) {
              const int64_t i = base + (int64_t)u * blockDim.x;
              const bool in = i < bd.end;
              px[u] = in ? xs[i] : 0.f; py[u] = in ? ys[i] : 0.f; pz[u] = in ? zs[i] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < kInFlight; ++u) {
            if (base + (int64_t)u * blockDim.x >= bd.end) break;
            const double xd = (double)px[u], yd = (double)py[u], zd = (double)pz[u];
            double q[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                double p = sM[r * 4 + 3];
                p = xd * sM[r * 4 + 0] + p;
                p = yd * sM[r * 4 + 1] + p;
                p = zd * sM[r * 4 + 2] + p;
                q[r] = (double)(float)p;
            }
#pragma unroll
            for (int k = 0; k < PB; ++k) {
                if (k < n_planes) {                   // uniform
                    const double* pl = sP + k * kPlaneStride;
                    const double num = __builtin_fabs(pl[0] * q[0] + pl[1] * q[1] + pl[2] * q[2] + pl[3]);
                    const unsigned hi = (unsigned)((unsigned long long)__double_as_longlong(num) >> 32);
                    double d;
                    if (__builtin_expect(pl[6] != 0.0 && hi >= kLo && hi < kHi, 1)) d = divide_by_known(num, pl[4], pl[5]);
                    else d = num / pl[4];
                    cnt[k] += d < thr ? 1u : 0u;
                }
            }
          }
        }
    }
#pragma unroll
    for (int k = 0; k < PB; ++k) {
        unsigned int c = cnt[k];
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
        if ((threadIdx.x & 63) == 0) sC[threadIdx.x >> 6][k] = c;
    }
    __syncthreads();
    if (threadIdx.x < PB) {
        const unsigned int c = sC[0][threadIdx.x] + sC[1][threadIdx.x] + sC[2][threadIdx.x] + sC[3][threadIdx.x];
        if (c) atomicAdd(&counts[threadIdx.x], (unsigned long long)c);
    }
}

} // namespace ccal
