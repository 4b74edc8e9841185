// voxel_impl.hpp -- device voxel-grid downsample (included at the end of ccal_api.hip).
//
// Stands in for pcl::VoxelGrid<PointXYZ> at src/camera_calibration_optimization.cpp:196-205
// (leaf 0.02f) with the same definition as the host filter (csrc/host/cloud_io.cpp): PCL's
// index arithmetic in float, voxels emitted in ascending linear index, the points of a voxel
// accumulated in input order in float.  Per cloud: bounds reduction -> keys -> stable radix
// sort of (key, point index) -> run-length encode -> coordinates gathered in sorted order; then ONE
// launch for the voxels of all clouds, one thread per voxel summing its run in order (a cloud has
// only a few hundred voxels of thousands of points each: per cloud that launch would be a few
// hundred threads adding for a long time).
#pragma once

#include <rocprim/rocprim.hpp>

namespace ccal {

__global__ __launch_bounds__(256) void voxel_bounds_kernel(const float* __restrict__ xs,
                                                           const float* __restrict__ ys,
                                                           const float* __restrict__ zs, int64_t begin,
                                                           int64_t end, float* __restrict__ out /* 6*blocks */)
{
    __shared__ float red[4][6];
    float mn[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f};
    float mx[3] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
    for (int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < end;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float v[3] = {xs[i], ys[i], zs[i]};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (!__builtin_isfinite(v[k])) continue;
            mn[k] = v[k] < mn[k] ? v[k] : mn[k];
            mx[k] = v[k] > mx[k] ? v[k] : mx[k];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        for (int off = 32; off > 0; off >>= 1) {
            const float a = __shfl_xor(mn[k], off, 64), b = __shfl_xor(mx[k], off, 64);
            mn[k] = a < mn[k] ? a : mn[k];
            mx[k] = b > mx[k] ? b : mx[k];
        }
        if (lane == 0) { red[wave][k] = mn[k]; red[wave][3 + k] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float v = red[0][k];
        for (int w = 1; w < 4; ++w) v = k < 3 ? (red[w][k] < v ? red[w][k] : v) : (red[w][k] > v ? red[w][k] : v);
        out[blockIdx.x * 6 + k] = v;
    }
}

struct VoxelDims {
    float inv;
    float minb[3];
    long long div0, div01;
    unsigned long long nonfinite_key;   // one past the last cell: sorts last with the fewest key bits
};

__global__ __launch_bounds__(256) void voxel_keys_kernel(const float* __restrict__ xs,
                                                         const float* __restrict__ ys,
                                                         const float* __restrict__ zs, int64_t begin,
                                                         int64_t n, VoxelDims d,
                                                         unsigned long long* __restrict__ keys,
                                                         unsigned int* __restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = xs[begin + i], y = ys[begin + i], z = zs[begin + i];
    unsigned long long key = d.nonfinite_key;   // non-finite points sort last and are dropped
    if (__builtin_isfinite(x) && __builtin_isfinite(y) && __builtin_isfinite(z)) {
        const long long i0 = (long long)(floorf(x * d.inv) - d.minb[0]);
        const long long i1 = (long long)(floorf(y * d.inv) - d.minb[1]);
        const long long i2 = (long long)(floorf(z * d.inv) - d.minb[2]);
        key = (unsigned long long)(i0 + i1 * d.div0 + i2 * d.div01);
    }
    keys[i] = key;
    idx[i] = (unsigned int)i;
}

// Coordinates in sorted order (voxel by voxel, input order inside a voxel: the sort is stable),
// so that the per-voxel sums below read contiguous runs.
__global__ __launch_bounds__(256) void voxel_gather_kernel(const float* __restrict__ xs,
                                                           const float* __restrict__ ys,
                                                           const float* __restrict__ zs, int64_t begin,
                                                           const unsigned int* __restrict__ sorted_idx, int64_t n,
                                                           float* __restrict__ gx, float* __restrict__ gy,
                                                           float* __restrict__ gz)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t i = begin + sorted_idx[t];
    gx[t] = xs[i];
    gy[t] = ys[i];
    gz[t] = zs[i];
}

// The run table of one cloud appended to the table of all clouds (offsets become global).
__global__ __launch_bounds__(256) void voxel_append_runs_kernel(const unsigned int* __restrict__ counts,
                                                                const unsigned int* __restrict__ offsets,
                                                                unsigned int n_runs, unsigned long long base,
                                                                unsigned int* __restrict__ all_counts,
                                                                unsigned long long* __restrict__ all_offsets)
{
    const unsigned int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_runs) return;
    all_counts[v] = counts[v];
    all_offsets[v] = base + offsets[v];
}

// One thread per voxel: float sums of its run in order (one add at a time, as a serial filter
// does); eight elements are in flight per trip so the loads do not wait for the adds.
__global__ __launch_bounds__(64) void voxel_centroid_kernel(
    const float* __restrict__ gx, const float* __restrict__ gy, const float* __restrict__ gz,
    const unsigned int* __restrict__ counts, const unsigned long long* __restrict__ offsets, unsigned int n_runs,
    float* __restrict__ out /* AoS xyz */)
{
    const unsigned int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_runs) return;                     // the host passes only the runs of real cells
    const unsigned long long o = offsets[v];
    const unsigned int c = counts[v];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    unsigned int t = 0;
    for (; t + 8 <= c; t += 8) {
        float x[8], y[8], z[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { x[u] = gx[o + t + u]; y[u] = gy[o + t + u]; z[u] = gz[o + t + u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { sx += x[u]; sy += y[u]; sz += z[u]; }
    }
    for (; t < c; ++t) { sx += gx[o + t]; sy += gy[o + t]; sz += gz[o + t]; }
    const float cnt = (float)c;
    out[3 * (size_t)v + 0] = sx / cnt;
    out[3 * (size_t)v + 1] = sy / cnt;
    out[3 * (size_t)v + 2] = sz / cnt;
}

} // namespace ccal

namespace {

// Grow a device buffer to `need` bytes keeping its first `keep` bytes.
hipError_t grow_keeping(ccal::DevBuf& b, size_t need, size_t keep, hipStream_t st)
{
    if (need <= b.bytes) return hipSuccess;
    ccal::DevBuf bigger;
    hipError_t e = bigger.ensure(need + need / 2);
    if (e != hipSuccess) return e;
    if (keep) {
        e = hipMemcpyAsync(bigger.p, b.p, keep, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { bigger.release(); return e; }
    }
    b.release();
    b = bigger;
    return hipSuccess;
}

// Voxels of points [begin, begin+n) of a resident set: their runs (count, offset into the gathered
// coordinates of the whole set) are appended to the run table of the call at position `used`;
// *n_vox_out = how many.  The centroids are formed afterwards, for all clouds at once.
int voxel_runs_of_range(ccal_ctx* ctx, const CloudSet& cs, int64_t begin, int64_t n, float leaf, size_t used,
                        unsigned int* n_vox_out)
{
    *n_vox_out = 0;
    hipStream_t st = ctx->stream;
    const float* xs = cs.xs.as<float>();
    const float* ys = cs.ys.as<float>();
    const float* zs = cs.zs.as<float>();
    const int nb = (int)std::min<int64_t>(256, (n + 255) / 256);
    CCAL_HIP(ctx, ctx->psum_buf.ensure((size_t)nb * 6 * sizeof(float)));
    hipLaunchKernelGGL(voxel_bounds_kernel, dim3(nb), dim3(256), 0, st, xs, ys, zs, begin, begin + n,
                       ctx->psum_buf.as<float>());
    std::vector<float> hb((size_t)nb * 6);
    CCAL_HIP(ctx, hipMemcpyAsync(hb.data(), ctx->psum_buf.p, hb.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    CCAL_HIP(ctx, hipStreamSynchronize(st));
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int b = 0; b < nb; ++b)
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::min(mn[k], hb[(size_t)b * 6 + k]);
            mx[k] = std::max(mx[k], hb[(size_t)b * 6 + 3 + k]);
        }
    if (!(mn[0] <= mx[0])) return CCAL_OK;   // no finite point
    VoxelDims d;
    d.inv = 1.0f / leaf;
    long long minb[3], div[3];
    for (int k = 0; k < 3; ++k) {
        minb[k] = (long long)std::floor(mn[k] * d.inv);
        div[k] = (long long)std::floor(mx[k] * d.inv) - minb[k] + 1;
        d.minb[k] = (float)minb[k];
    }
    d.div0 = div[0];
    d.div01 = div[0] * div[1];
    const double cells = (double)div[0] * (double)div[1] * (double)div[2];
    if (!(cells < 9.0e18)) return fail(ctx, CCAL_ERR_ARG, "voxel grid: leaf too small for the cloud extent");
    d.nonfinite_key = (unsigned long long)div[0] * (unsigned long long)div[1] * (unsigned long long)div[2];
    unsigned key_bits = 1;
    while (key_bits < 64 && (1ull << key_bits) <= d.nonfinite_key) ++key_bits;

    // keys, sort, run-length encode (scratch lives in the context and only grows)
    const size_t N = (size_t)n;
    DevBuf& keys = ctx->vox[0];
    DevBuf& keys2 = ctx->vox[1];
    DevBuf& idx = ctx->vox[2];
    DevBuf& idx2 = ctx->vox[3];
    DevBuf& uniq = ctx->vox[4];
    DevBuf& counts = ctx->vox[5];
    DevBuf& offsets = ctx->vox[6];
    DevBuf& nruns = ctx->vox[7];
    DevBuf& tmp = ctx->vox[8];
    DevBuf& gathered = ctx->vox[10];      // x | y | z of the whole set in sorted order, sized by the caller
    DevBuf& all_counts = ctx->vox[11];
    DevBuf& all_offsets = ctx->vox[12];
    CCAL_HIP(ctx, keys.ensure(N * 8)); CCAL_HIP(ctx, keys2.ensure(N * 8));
    CCAL_HIP(ctx, idx.ensure(N * 4)); CCAL_HIP(ctx, idx2.ensure(N * 4));
    CCAL_HIP(ctx, uniq.ensure(N * 8)); CCAL_HIP(ctx, counts.ensure(N * 4));
    CCAL_HIP(ctx, offsets.ensure(N * 4)); CCAL_HIP(ctx, nruns.ensure(4));
    hipLaunchKernelGGL(voxel_keys_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, xs, ys, zs, begin, n, d,
                       keys.as<unsigned long long>(), idx.as<unsigned int>());
    // keys are below 2^key_bits (the non-finite key included): only those bits are sorted
    size_t tmp_bytes = 0;
    CCAL_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys.as<unsigned long long>(),
                                            keys2.as<unsigned long long>(), idx.as<unsigned int>(),
                                            idx2.as<unsigned int>(), N, 0, key_bits, st));
    CCAL_HIP(ctx, tmp.ensure(tmp_bytes));
    CCAL_HIP(ctx, rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys.as<unsigned long long>(),
                                            keys2.as<unsigned long long>(), idx.as<unsigned int>(),
                                            idx2.as<unsigned int>(), N, 0, key_bits, st));
    size_t tmp2 = 0;
    CCAL_HIP(ctx, rocprim::run_length_encode(nullptr, tmp2, keys2.as<unsigned long long>(), (unsigned int)N,
                                             uniq.as<unsigned long long>(), counts.as<unsigned int>(),
                                             nruns.as<unsigned int>(), st));
    CCAL_HIP(ctx, tmp.ensure(tmp2));
    CCAL_HIP(ctx, rocprim::run_length_encode(tmp.p, tmp2, keys2.as<unsigned long long>(), (unsigned int)N,
                                             uniq.as<unsigned long long>(), counts.as<unsigned int>(),
                                             nruns.as<unsigned int>(), st));
    unsigned int n_runs = 0;
    CCAL_HIP(ctx, hipMemcpyAsync(&n_runs, nruns.p, 4, hipMemcpyDeviceToHost, st));
    CCAL_HIP(ctx, hipStreamSynchronize(st));
    size_t tmp3 = 0;
    CCAL_HIP(ctx, rocprim::exclusive_scan(nullptr, tmp3, counts.as<unsigned int>(), offsets.as<unsigned int>(), 0u,
                                          n_runs, rocprim::plus<unsigned int>(), st));
    CCAL_HIP(ctx, tmp.ensure(tmp3));
    CCAL_HIP(ctx, rocprim::exclusive_scan(tmp.p, tmp3, counts.as<unsigned int>(), offsets.as<unsigned int>(), 0u,
                                          n_runs, rocprim::plus<unsigned int>(), st));
    // the run of non-finite points, if any, is the last one
    unsigned long long last_key = 0;
    CCAL_HIP(ctx, hipMemcpyAsync(&last_key, uniq.as<unsigned long long>() + (n_runs - 1), 8, hipMemcpyDeviceToHost, st));
    CCAL_HIP(ctx, hipStreamSynchronize(st));
    const unsigned int n_vox = last_key == d.nonfinite_key ? n_runs - 1 : n_runs;
    if (n_vox) {
        const size_t total = cs.total_pts;
        float* gx = gathered.as<float>() + begin;
        hipLaunchKernelGGL(voxel_gather_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, xs, ys, zs, begin,
                           idx2.as<unsigned int>(), n, gx, gx + total, gx + 2 * total);
        CCAL_HIP(ctx, grow_keeping(all_counts, (used + n_vox) * sizeof(unsigned int), used * sizeof(unsigned int), st));
        CCAL_HIP(ctx, grow_keeping(all_offsets, (used + n_vox) * sizeof(unsigned long long),
                                   used * sizeof(unsigned long long), st));
        hipLaunchKernelGGL(voxel_append_runs_kernel, dim3((n_vox + 255) / 256), dim3(256), 0, st, counts.as<unsigned int>(),
                           offsets.as<unsigned int>(), n_vox, (unsigned long long)begin, all_counts.as<unsigned int>() + used,
                           all_offsets.as<unsigned long long>() + used);
        CCAL_HIP(ctx, hipGetLastError());
    }
    *n_vox_out = n_vox;
    return CCAL_OK;
}

} // namespace

extern "C" int ccal_downsample_raw(ccal_ctx* ctx, float leaf, size_t* n_pts_out)
{
    if (int rc = check_set(ctx, CCAL_RAW)) return rc;
    if (!(leaf > 0.f)) return fail(ctx, CCAL_ERR_ARG, "downsample_raw: leaf must be positive");
    if (int rc = bind(ctx)) return rc;
    const CloudSet& raw = ctx->sets[CCAL_RAW];
    const int C = raw.n_clouds;
    CCAL_HIP(ctx, ctx->vox[10].ensure(3 * raw.total_pts * sizeof(float)));
    std::vector<unsigned int> n_vox(C, 0);
    size_t used = 0;
    int64_t begin = 0;
    for (int j = 0; j < C; ++j) {
        if (int rc = voxel_runs_of_range(ctx, raw, begin, (int64_t)raw.n_pts[j], leaf, used, &n_vox[j])) return rc;
        begin += (int64_t)raw.n_pts[j];
        if (n_vox[j] == 0) return fail(ctx, CCAL_ERR_ARG, "downsample_raw: a cloud has no finite point");
        used += n_vox[j];
    }
    if (used >= ((size_t)1 << 32)) return fail(ctx, CCAL_ERR_CAPACITY, "downsample_raw: too many voxels");
    // the centroids of all clouds in one launch
    DevBuf& cent = ctx->vox[9];
    CCAL_HIP(ctx, cent.ensure(used * 3 * sizeof(float)));
    const float* gx = ctx->vox[10].as<float>();
    hipLaunchKernelGGL(voxel_centroid_kernel, dim3((unsigned)((used + 63) / 64)), dim3(64), 0, ctx->stream, gx,
                       gx + raw.total_pts, gx + 2 * raw.total_pts, ctx->vox[11].as<unsigned int>(),
                       ctx->vox[12].as<unsigned long long>(), (unsigned int)used, cent.as<float>());
    CCAL_HIP(ctx, hipGetLastError());
    std::vector<float> down(used * 3);
    CCAL_HIP(ctx, hipMemcpyAsync(down.data(), cent.p, down.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<const float*> ptrs(C);
    std::vector<size_t> sizes(C);
    size_t at = 0;
    for (int j = 0; j < C; ++j) {
        ptrs[j] = down.data() + 3 * at;
        sizes[j] = n_vox[j];
        at += n_vox[j];
        if (n_pts_out) n_pts_out[j] = sizes[j];
    }
    const std::vector<double> B = raw.base_T_flange;   // upload replaces sets[which] only, copy is defensive
    return ccal_upload_clouds(ctx, CCAL_DOWNSAMPLED, C, ptrs.data(), sizes.data(), B.data());
}

extern "C" int ccal_cloud_size(const ccal_ctx* ctx, int which, int cloud, size_t* n_out)
{
    if (!ctx || !n_out || (which != CCAL_DOWNSAMPLED && which != CCAL_RAW)) return CCAL_ERR_ARG;
    const CloudSet& cs = ctx->sets[which];
    if (!cs.ready) return CCAL_ERR_STATE;
    if (cloud < 0 || cloud >= cs.n_clouds) return CCAL_ERR_ARG;
    *n_out = cs.n_pts[cloud];
    return CCAL_OK;
}

extern "C" int ccal_download_cloud(ccal_ctx* ctx, int which, int cloud, float* xyz_out)
{
    if (int rc = check_set(ctx, which)) return rc;
    const CloudSet& cs = ctx->sets[which];
    if (cloud < 0 || cloud >= cs.n_clouds || !xyz_out) return fail(ctx, CCAL_ERR_ARG, "download_cloud: bad argument");
    if (int rc = bind(ctx)) return rc;
    size_t off = 0;
    for (int j = 0; j < cloud; ++j) off += cs.n_pts[j];
    const size_t n = cs.n_pts[cloud];
    std::vector<float> x(n), y(n), z(n);
    CCAL_HIP(ctx, hipMemcpy(x.data(), cs.xs.as<float>() + off, n * sizeof(float), hipMemcpyDeviceToHost));
    CCAL_HIP(ctx, hipMemcpy(y.data(), cs.ys.as<float>() + off, n * sizeof(float), hipMemcpyDeviceToHost));
    CCAL_HIP(ctx, hipMemcpy(z.data(), cs.zs.as<float>() + off, n * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) { xyz_out[3 * i] = x[i]; xyz_out[3 * i + 1] = y[i]; xyz_out[3 * i + 2] = z[i]; }
    return CCAL_OK;
}
