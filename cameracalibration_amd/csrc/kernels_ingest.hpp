// kernels_ingest.hpp -- upload: cloud centres, SoA arrays and centred 64-point groups from the staged AoS copy
//
// One of the kernel headers of libccal; each is included by exactly one translation unit
// (api_core.hip), so its __global__ functions have one definition in the library.
#pragma once

#include "kernels_common.hpp"

#pragma clang fp contract(off)

namespace ccal {

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
                                                         float* __restrict__ radius_sums /* this cloud: four per block */)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double cx = centre[0], cy = centre[1], cz = centre[2];
    float r2 = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
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
        ax = __builtin_fabsf(dx); ay = __builtin_fabsf(dy); az = __builtin_fabsf(dz);
        const int64_t src = fast_slot_source(i, n, stride);
        const float sx = aos[3 * src + 0], sy = aos[3 * src + 1], sz = aos[3 * src + 2];
        gp[(0 * 16 + col) * 4 + u] = (float)((double)sx - cx);
        gp[(1 * 16 + col) * 4 + u] = (float)((double)sy - cy);
        gp[(2 * 16 + col) * 4 + u] = (float)((double)sz - cz);
        gp[(3 * 16 + col) * 4 + u] = 1.0f;
    }
    // largest squared distance from the centre: non-negative floats order like their bit patterns,
    // and a NaN's pattern lies above infinity's, so a non-finite point leaves a non-finite mark
    // sums of the distances from the centre and of their components along the cloud's axes, per block in a
    // fixed order (the host adds the blocks)
    __shared__ float wave_sums[4][4];
    float rs[4] = {__builtin_sqrtf(r2), ax, ay, az};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        for (int off = 32; off > 0; off >>= 1) rs[k] += __shfl_xor(rs[k], off, 64);
        if ((threadIdx.x & 63) == 0) wave_sums[threadIdx.x >> 6][k] = rs[k];
    }
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
#pragma unroll
        for (int k = 0; k < 4; ++k)
            radius_sums[4 * (int64_t)blockIdx.x + k] = ((wave_sums[0][k] + wave_sums[1][k]) + wave_sums[2][k]) + wave_sums[3][k];
    }
}


} // namespace ccal
